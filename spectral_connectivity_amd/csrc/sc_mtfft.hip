// sc_mtfft.hip -- fused multitaper transform for power-of-two FFT lengths:
//   sliding-window extraction + detrend + DPSS taper multiply + real FFT + transposed store,
// one kernel, the time series is read once and the one-sided spectra are written once,
// straight into the layout stage B consumes (X[f][w][r][k][c], channels fastest).
//
// Why not rocFFT here: its batched real transform is fast only for unit-stride batches
// ([batch][n] -> [batch][f]); the layouts the path needs (lanes <-> channels on both sides)
// cost it 8-10x (measured on MI355X, profiles/r01_rocfft_layouts.txt) and add a 6.4 GB
// tapered-window round trip through HBM.  rocFFT stays the transform for every other length.
//
// Workgroup = one (window w, trial r, tile of CT channels).  The L x CT window tile is loaded
// once into LDS with coalesced rows (channels are the fastest axis of x), the trend sums are taken in fp64,
// and then for every taper k the CT real sequences are packed two-by-two (channels c, c+1) into CT/2 complex
// sequences ("two-for-one" real FFT), transformed by register-resident radix-16 butterflies with LDS
// exchanges between the passes (fp32, twiddles from an LDS table rounded from fp64), separated by
// conjugate symmetry
//   A[f] = (Z[f] + conj Z[N-f]) / 2,   B[f] = (Z[f] - conj Z[N-f]) / (2i)
// (DC and Nyquist come out exactly real, like the reference's real input), and stored as
// float4 (A, B) so a wave writes contiguous CT*8-byte segments per frequency.
#include <cstdlib>
#include <type_traits>
#include "sc_common.h"
#include "sc_mtfft_bfly.h"

struct MtArgs {
    const float* x;
    const float* tapers;   // [K][L], already divided by fs
    const float2* tw;      // [N] exp(-2 pi i m / N)
    float2* X;             // [F][W][R][K][C]
    int T, R, C, L, step, W, K, detrend;
    float2* Z;             // long windows (N >= 2048): row-major output [w][r - r_off][k][c][F], transposed into X afterwards
    int r_off, Rc;         //   trials [r_off, r_off + Rc) of this launch
    int kh;                // tapers resident in LDS (K, or 1 = reload per taper)
    int dbg;               // profiling aid (env SC_MTFFT_DEBUG bit mask, results WRONG when set):
                           // 1 = no HBM stores, 2 = skip both radix-16 passes, 4 = skip the split/store loop,
                           // 8 = every wave takes the store loop with the silent / non-finite channel overrides (A/B of the fast loop),
                           // 16 = plain instead of non-temporal stores (A/B)
    // planes-format output (sc_fused2.hip: two f16 pieces per real, x * scale[c] = h + m, rows [f][w][r][k] of row_bytes):
    // when P is set the spectra go there INSTEAD of X
    unsigned char* P;
    const float* scale;    // [C] powers of two
    int64_t row_bytes;
};

// Optional phase timers (tools/mtfft_trace.py builds a copy of the library with -DMT_TRACE): shader-clock
// cycles of wave 0 of workgroup 0, summed over the tapers: [load+detrend, wait at the top barrier, passes, split+store issue].
#ifdef MT_TRACE
__device__ unsigned long long mt_trace_buf[8];
#define MT_T0() unsigned long long mt_t = __builtin_readcyclecounter()
#define MT_TICK(slot)                                                                   \
    do {                                                                                \
        const unsigned long long mt_n = __builtin_readcyclecounter();                   \
        if (blockIdx.x == 0 && blockIdx.y == 3 && blockIdx.z == 3 && threadIdx.x == 0) mt_trace_buf[slot] += mt_n - mt_t; \
        mt_t = mt_n;                                                                    \
    } while (0)
extern "C" int sc_debug_mtfft_trace(unsigned long long* out, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mt_trace_buf), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mt_trace_buf), z, sizeof z); }
    return 0;
}
#else
#define MT_T0() do {} while (0)
#define MT_TICK(slot) do {} while (0)
#endif

typedef _Float16 mt_h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned mt_pack_f16(float lo, float hi) {        // two f16 (round to nearest) in one dword
    const mt_h16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, v);
}
// two scaled reals -> the dwords of their leading and trailing f16 pieces (x = h + m to 22 significant bits)
__device__ __forceinline__ void mt_split2(float x0, float x1, unsigned& h, unsigned& m) {
    // h = f16(x), m = f16(x - h), two values per register: one packed conversion and two mixed-precision fmas that read
    // their f16 operand straight from the halves of h (left to the compiler: four conversions and a subtraction per value)
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(x0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(x1), "v"(h));
}

// One step of a 4 x 4 transpose inside a quad of lanes (planes-format store loop): lanes with `hi` clear give b and take the
// partner's a into b, lanes with `hi` set give a and take the partner's b into a; the partner is lane ^ 1 (CTRL = quad_perm
// [1, 0, 3, 2]) or lane ^ 2 ([2, 3, 0, 1]).
template <int CTRL>
__device__ __forceinline__ void mt_quad_xchg(u32x4_t& a, u32x4_t& b, bool hi) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned give = hi ? a[d] : b[d];
        const unsigned take = (unsigned)__builtin_amdgcn_mov_dpp((int)give, CTRL, 0xf, 0xf, true);
        if (hi) a[d] = take; else b[d] = take;
    }
}

// THREADS: 256 by default.  Long windows (N >= 1024) take 512-thread workgroups -- twice the transforms, so twice the
// contiguous piece of a frequency row per store group (128 bytes at N = 1024: a full line) -- at the same number of
// waves per CU; 1024 threads would need 128 registers per lane and spill (measured slower).
template <int LOG2N, int THREADS = 256, bool PL = false>      // PL: planes-format output (its own instantiation: registers)
__global__ void __launch_bounds__(THREADS, THREADS == 512 ? 2 : (LOG2N <= 8 ? 4 : (LOG2N == 9 ? 3 : (LOG2N >= 11 ? 3 : 1))))
mtfft16_kernel(MtArgs p) {
    constexpr int N = 1 << LOG2N;
    constexpr int TPF = N / 16;          // threads per FFT: 16 points each
    constexpr int NF = THREADS / TPF;    // complex FFTs (channel pairs) per workgroup
    constexpr int CT = 2 * NF;           // channels per workgroup
    constexpr int XS = CT + 2;           // padded window-row stride (floats)
    constexpr int ZS = N + N / 16 + 1;   // skewed exchange buffer per FFT (float2), odd stride
    extern __shared__ __align__(16) unsigned char smem[];
    // The window tile lives in LDS only until every thread has pulled its 16 x 2 samples into
    // registers (they are the same for all tapers); the exchange buffer z then reuses that space,
    // and the detrend scratch is later reused for the twiddle / taper tables: 39.6 KB per workgroup at N = 256,
    // four workgroups (16 waves) per CU.
    // Long windows (N >= 2048, 2-4 channels per workgroup) keep NO window tile, twiddle table or taper buffer in LDS --
    // samples and taper values come straight from HBM / L2 into registers, twiddles from two 64-entry tables -- so
    // that two workgroups share a CU (45 KB each instead of 131 KB at N = 4096).  N = 4096 stores every channel's spectrum
    // as one contiguous row that a tiled transpose turns into the frequency-major X (16-byte pieces per frequency row
    // otherwise); N = 2048 stores its 32-byte pieces directly, the tiles of a line on one XCD.
    constexpr bool LONG = LOG2N >= 11;
    // (Tried at N = 1024 and dropped: the window tile through LDS in two halves + the small twiddle tables of the long
    //  windows -- 43 KB, three workgroups per CU instead of two: 2.0 TB/s against 2.5.  The kernel is bound by VALU / LDS
    //  instruction issue there, not by latency: the twiddle products cost more than the third workgroup returns.)
    constexpr size_t XT_BYTES = LONG ? 0 : (size_t)N * XS * 4, Z_BYTES = (size_t)NF * ZS * 8;
    constexpr size_t UNION_BYTES = XT_BYTES > Z_BYTES ? XT_BYTES : Z_BYTES;
    float* xt = reinterpret_cast<float*>(smem);                                   // [N][XS]
    float2* z = reinterpret_cast<float2*>(smem);                                  // [NF][ZS] (aliases xt)
    // The detrend scratch is dead once the tile is detrended; twiddles and tapers then take its place.
    // (long windows: the trend sums borrow the exchange buffer, which nobody has written yet -- four arrays of THREADS
    //  doubles would not fit next to it)
    double* red = reinterpret_cast<double*>(smem + (LONG ? 0 : UNION_BYTES));     // [2][THREADS] + trend [2][CT]; LONG: [4][THREADS]
    float2* tw = reinterpret_cast<float2*>(smem + UNION_BYTES);                   // [N]   (aliases red)
    float* hk = reinterpret_cast<float*>(tw + N);                                 // [kh][L] tapers
    float2* tlo = reinterpret_cast<float2*>(smem + UNION_BYTES);                  // LONG: W_N^j, j < 64
    float2* thi = tlo + 64;                                                        // LONG: W_N^(64 q), q < N / 64
    // An FFT's 16 x TPF points are exchanged between lanes of ONE wavefront when TPF <= 64, and
    // LDS executes a wave's instructions in order: those exchanges need no workgroup barrier.
    constexpr bool WAVE_LOCAL = TPF <= 64;
    // A channel whose (detrended) window is identically zero must come out as exact zeros, like the reference's
    // per-channel transform gives (its measures turn NaN on zero power): the conjugate-symmetry split of a packed pair
    // would leave the rounding noise of its partner there.  One flag per channel of the tile.
    __shared__ int nzf[CT], nbf[CT];       // (plain stores of a constant: many threads may set the same flag)
    // complex64 output: largest finite |sample| of every channel of THIS window (bit pattern; non-negative floats order like
    // unsigned).  The two channels of a pair enter their shared transform scaled to [1, 2) by powers of two and leave it scaled
    // back -- exact -- so that a weak channel does not carry the float32 rounding of a strong pair partner (6e-5 of its own
    // largest coefficient for a x500 pair before).  (Planes output: the global channel scales already did this at the load.)
    __shared__ unsigned mxc[CT];

    const int tid = threadIdx.x;
    if (tid < CT) { nzf[tid] = 0; nbf[tid] = 0; mxc[tid] = 0u; }
    if constexpr (LONG) __syncthreads();
    MT_T0();
    int c0, r, w;
    constexpr bool ROWSTORE = LOG2N >= 12 && THREADS == 256;     // every channel's spectrum as one row of Z, transposed afterwards
    constexpr bool XCDMAP = LOG2N >= 10;         // N = 1024 stores 64-byte pieces of a frequency row: the tiles that complete a
                                                 // 128-byte line must meet in ONE XCD's L2 as well
    if constexpr (XCDMAP) {
        // A thread reads 8 bytes of every window row, so the channel tiles of one (window, trial) share every line they
        // fetch: keep them on ONE XCD (block b runs on XCD b % 8) and that XCD's L2 fetches each line from HBM once.
        const int n_ct = (p.C + CT - 1) / CT;
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int g = (j / n_ct) * 8 + xcd;                 // (window, trial of this launch)
        if (g >= p.W * p.Rc) return;
        c0 = (j % n_ct) * CT; w = g / p.Rc; r = p.r_off + (g - w * p.Rc);
    } else {
        c0 = blockIdx.x * CT; r = blockIdx.y; w = blockIdx.z;
    }
    const int L = p.L, C = p.C;
    const int64_t RC = (int64_t)p.R * C;
    const bool resident = p.kh == p.K;
    const float* xw = p.x + ((int64_t)w * p.step * p.R + r) * C + c0;
    const int pf = tid / TPF, i = tid - pf * TPF;     // FFT (channel pair) and butterfly index
    float2* zf = z + pf * ZS;
    const int F = N / 2 + 1;
    const int64_t sF = (int64_t)p.W * p.R * p.K * C;
    const bool vec_ok = (C % 2) == 0;
    float2 xs[16];                                    // this thread's pass-1 inputs, all tapers
    if constexpr (LONG) {
        const int c = c0 + 2 * pf;
        const float* src0 = xw + 2 * pf;
        const bool pair = vec_ok && c + 1 < C;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int n = i + t * TPF;
            float2 v = make_float2(0.f, 0.f);
            if (n < L && c < C) {
                const float* src = src0 + (int64_t)n * RC;
                if (pair) v = *reinterpret_cast<const float2*>(src);
                else { v.x = src[0]; if (c + 1 < C) v.y = src[1]; }
            }
            xs[t] = v;
        }
        if (p.detrend != SC_DETREND_NONE) {
            // trend sums in fp64: a thread's 16 samples, then a tree over the TPF threads of the transform
            double s0 = 0.0, t0 = 0.0, s1 = 0.0, t1 = 0.0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double l1 = (double)(i + t * TPF + 1);
                s0 += (double)xs[t].x; t0 += (double)xs[t].x * l1;
                s1 += (double)xs[t].y; t1 += (double)xs[t].y * l1;
            }
            red[tid] = s0; red[THREADS + tid] = t0; red[2 * THREADS + tid] = s1; red[3 * THREADS + tid] = t1;
            __syncthreads();
            for (int h = TPF / 2; h > 0; h >>= 1) {
                if (i < h) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[q * THREADS + tid] += red[q * THREADS + tid + h];
                }
                __syncthreads();
            }
            const double n = (double)L, invL = 1.0 / n;
            const double St = (n + 1.0) * 0.5, Stt = (n + 1.0) * (2.0 * n + 1.0) / (6.0 * n), den = n * Stt - St * St;
            double ab[2][2];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const double sum = red[(2 * ch) * THREADS + pf * TPF], sumt = red[(2 * ch + 1) * THREADS + pf * TPF] / n;
                double a = 0.0, b;
                if (p.detrend == SC_DETREND_CONSTANT) {
                    b = sum / n;
                } else {
                    a = (den != 0.0) ? (n * sumt - St * sum) / den : 0.0;
                    b = (sum - a * St) / n;
                }
                ab[ch][0] = a; ab[ch][1] = b;
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int nn = i + t * TPF;
                if (nn < L) {
                    const double tt = (double)(nn + 1) * invL;
                    xs[t].x = (float)((double)xs[t].x - (ab[0][0] * tt + ab[0][1]));
                    xs[t].y = (float)((double)xs[t].y - (ab[1][0] * tt + ab[1][1]));
                }
            }
        }
        if (tid < 64) tlo[tid] = p.tw[tid];
        else if (tid < 64 + N / 64) thi[tid - 64] = p.tw[(tid - 64) * 64];
        // (tlo / thi sit behind the exchange buffer; the trend sums in front of it are consumed before the barrier below)
    } else {
        if constexpr (CT % 4 == 0) {
            // 16-byte loads, all of a thread's rows in flight at once (the tile is N x CT floats = 16 KB x 2)
            constexpr int V = CT / 4, ROUNDS = N * V / THREADS;
            const bool vec = (C % 4) == 0;
            float4 v[ROUNDS];
            // Planes format: the channel scales (powers of two) go onto the SAMPLES -- exact, every later step is linear --
            // so the store loop has no multiply left, and the two channels that share a complex transform enter it
            // at the same magnitude (a weak channel no longer carries the rounding of a strong pair partner).
            float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (PL) {
                const int cs = c0 + 4 * (tid % V);          // (idx % V does not change over the rounds: THREADS % V == 0)
                if (cs + 3 < C) {
                    sc4 = *reinterpret_cast<const float4*>(p.scale + cs);       // (cs is a multiple of four)
                } else {
                    if (cs < C) sc4.x = p.scale[cs];
                    if (cs + 1 < C) sc4.y = p.scale[cs + 1];
                    if (cs + 2 < C) sc4.z = p.scale[cs + 2];
                }
            }
    #pragma unroll
            for (int it = 0; it < ROUNDS; ++it) {
                const int idx = tid + it * THREADS, l = idx / V, cc = 4 * (idx - l * V);
                const float* src = xw + (int64_t)l * RC + cc;
                v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (l < L) {
                    if (vec && c0 + cc + 3 < C) {
                        v[it] = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (c0 + cc < C) v[it].x = src[0];
                        if (c0 + cc + 1 < C) v[it].y = src[1];
                        if (c0 + cc + 2 < C) v[it].z = src[2];
                        if (c0 + cc + 3 < C) v[it].w = src[3];
                    }
                }
            }
    #pragma unroll
            for (int it = 0; it < ROUNDS; ++it) {
                const int idx = tid + it * THREADS, l = idx / V, cc = 4 * (idx - l * V);
                if (l < L) {
                    float2* d = reinterpret_cast<float2*>(xt + l * XS + cc);
                    if constexpr (PL) {
                        d[0] = make_float2(v[it].x * sc4.x, v[it].y * sc4.y);
                        d[1] = make_float2(v[it].z * sc4.z, v[it].w * sc4.w);
                    } else {
                        d[0] = make_float2(v[it].x, v[it].y);
                        d[1] = make_float2(v[it].z, v[it].w);
                    }
                }
            }
        } else {
            for (int idx = tid; idx < L * CT; idx += THREADS) {
                const int l = idx / CT, cc = idx - l * CT;
                xt[l * XS + cc] = (c0 + cc < C) ? xw[(int64_t)l * RC + cc] : 0.f;
            }
        }
        __syncthreads();
        if (p.detrend != SC_DETREND_NONE) {
            constexpr int SL = THREADS / CT;
            const int cc = tid % CT, sl = tid / CT;
            double s = 0.0, st = 0.0;
            for (int l = sl; l < L; l += SL) {
                const double v = (double)xt[l * XS + cc];
                s += v;
                st += v * (double)(l + 1);
            }
            red[tid] = s;
            red[THREADS + tid] = st;
            __syncthreads();
            if (tid < CT) {
                double sum = 0.0, sumt = 0.0;
                for (int q = 0; q < SL; ++q) { sum += red[q * CT + tid]; sumt += red[THREADS + q * CT + tid]; }
                sumt /= (double)L;
                const double n = (double)L;
                double a = 0.0, b;
                if (p.detrend == SC_DETREND_CONSTANT) {
                    b = sum / n;
                } else {
                    const double St = (n + 1.0) * 0.5, Stt = (n + 1.0) * (2.0 * n + 1.0) / (6.0 * n);
                    const double den = n * Stt - St * St;
                    a = (den != 0.0) ? (n * sumt - St * sum) / den : 0.0;
                    b = (sum - a * St) / n;
                }
                red[2 * THREADS + tid] = a;
                red[2 * THREADS + CT + tid] = b;
            }
            // (the trend a t + b is subtracted below, while the samples are pulled into registers)
        }
        __syncthreads();                                  // tile and trend coefficients visible

        {
            const bool detr = p.detrend != SC_DETREND_NONE;
            const double invL = 1.0 / (double)L;
            const double a0 = detr ? red[2 * THREADS + 2 * pf] : 0.0, a1 = detr ? red[2 * THREADS + 2 * pf + 1] : 0.0;
            const double b0 = detr ? red[2 * THREADS + CT + 2 * pf] : 0.0, b1 = detr ? red[2 * THREADS + CT + 2 * pf + 1] : 0.0;
    #pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int n = i + t * TPF;
                float2 v = make_float2(0.f, 0.f);
                if (n < L) {
                    v = *reinterpret_cast<const float2*>(xt + n * XS + 2 * pf);
                    if (detr) {           // same fp64 expression as a separate detrend pass would use
                        const double tt = (double)(n + 1) * invL;
                        v.x = (float)((double)v.x - (a0 * tt + b0));
                        v.y = (float)((double)v.y - (a1 * tt + b1));
                    }
                }
                xs[t] = v;
            }
        }
    }
    {
        // flag 1: the channel is not identically zero; flag 2: it holds a NaN / infinity.  The reference transforms every
        // channel on its own, so a non-finite sample spoils that channel's spectrum only: such a channel leaves the packed
        // transform (zeros in its place, its partner stays clean) and its bins are written as NaN.
        // (integer tests on the bit patterns -- OR of the magnitudes != 0: some sample is not zero; largest magnitude with
        //  the exponent field all ones: NaN or infinity -- measured 10 % cheaper on the mixed-radix kernels than the
        //  floating-point comparisons)
        unsigned or0 = 0u, or1 = 0u, mx0 = 0u, mx1 = 0u;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned u0 = __float_as_uint(xs[t].x) & 0x7fffffffu, u1 = __float_as_uint(xs[t].y) & 0x7fffffffu;
            or0 |= u0; or1 |= u1;
            mx0 = mx0 > u0 ? mx0 : u0; mx1 = mx1 > u1 ? mx1 : u1;
        }
        const bool n0 = or0 != 0u, n1 = or1 != 0u, b0 = mx0 >= 0x7f800000u, b1 = mx1 >= 0x7f800000u;
        if (n0) nzf[2 * pf] = 1;
        if (n1) nzf[2 * pf + 1] = 1;
        if (b0) nbf[2 * pf] = 1;
        if (b1) nbf[2 * pf + 1] = 1;
        if constexpr (!PL) {
            if (!b0 && n0) atomicMax(&mxc[2 * pf], mx0);
            if (!b1 && n1) atomicMax(&mxc[2 * pf + 1], mx1);
        }
    }
    __syncthreads();                                  // tile and detrend scratch consumed: their space is free
    if (nbf[2 * pf]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) xs[t].x = 0.f;
    }
    if (nbf[2 * pf + 1]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) xs[t].y = 0.f;
    }
    // The samples enter halved -- once per workgroup, they serve every taper: the 1/2 of the conjugate-symmetry split, an exact
    // scaling, leaves the store loop.  (Halving the TAPERS instead put a multiply, i.e. a wait for the taper loads, in front
    // of the barrier of the long-window kernels: 2048 samples 2.0 -> 2.6 ms, bisected with variant libraries.)
    // 2^(127 - E) for a largest magnitude m 2^(E - 127), m in [1, 2): exponent field 254 - E; 1 for zero / denormal / huge maxima
    auto pair_scale = [](unsigned mx, bool inverse) -> float {
        const unsigned E = mx >> 23;
        return (E >= 1u && E <= 253u) ? __uint_as_float((inverse ? E : 254u - E) << 23) : 1.f;
    };
    {
        const float h0 = 0.5f * (PL ? 1.f : pair_scale(mxc[2 * pf], false)), h1 = 0.5f * (PL ? 1.f : pair_scale(mxc[2 * pf + 1], false));
#pragma unroll
        for (int t = 0; t < 16; ++t) { xs[t].x *= h0; xs[t].y *= h1; }
    }
    const int spr = 2 * (tid & (NF - 1));                                   // the pair this thread stores
    const bool na = nbf[spr] != 0, nb = nbf[spr + 1] != 0, za = !na && nzf[spr] == 0, zb = !nb && nzf[spr + 1] == 0;
    const bool any_flag = __builtin_amdgcn_ballot_w64(na || nb || za || zb) != 0ull || (p.dbg & 8);      // wave-uniform
    if constexpr (!LONG) {
        for (int i2 = tid; i2 < N; i2 += THREADS) tw[i2] = p.tw[i2];
        if (resident) {
            for (int i2 = tid; i2 < p.K * p.L; i2 += THREADS) hk[i2] = p.tapers[i2];
        } else {
            for (int i2 = tid; i2 < L; i2 += THREADS) hk[i2] = p.tapers[i2];          // taper 0 into buffer 0
        }
    }
    // W_N^m: the LDS table, or (long windows) the product of the two 64-entry tables
    auto TW = [&](int m) -> float2 {
        if constexpr (LONG) return cmul(tlo[m & 63], thi[m >> 6]);
        else return tw[m];
    };
#define PHYS(idx) ((idx) + ((idx) >> 4))
#define XBAR()                                                      \
    do {                                                            \
        if constexpr (WAVE_LOCAL) {                                 \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
            __builtin_amdgcn_wave_barrier();                        \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
        } else {                                                    \
            __syncthreads();                                        \
        }                                                           \
    } while (0)
    MT_TICK(0);
    for (int k = 0; k < p.K; ++k) {
        // Two taper buffers when the exchanges are wave-local (a fast wave parks taper k + 1 while a slow one still
        // reads taper k in pass 1); from N = 2048 on every pass ends in a workgroup barrier, so by the time anyone parks
        // the next taper all of pass 1 is done and ONE buffer is enough -- which puts N = 2048 under 80 KB of LDS,
        // two workgroups per CU.
        constexpr int TWO = WAVE_LOCAL ? 1 : 0;
        const float* hkk = resident ? hk + k * L : hk + (k & TWO) * L;
        // the next taper travels HBM/L2 -> registers under this taper's passes and is parked in the other
        // LDS buffer before the stores go out (a load issued AFTER the stores would wait for them: vmcnt
        // retires in order)
        constexpr int HN = (N + THREADS - 1) / THREADS;
        float hn[HN];
        const bool fetch_next = !LONG && !resident && k + 1 < p.K;
        float hl[LONG ? 16 : 1];                      // long windows: this taper's 16 values, straight from L2
        if constexpr (LONG) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int n = i + t * TPF;
                hl[t] = (n < L) ? p.tapers[(int64_t)k * L + n] : 0.f;
            }
        }
        if (fetch_next) {
#pragma unroll
            for (int j = 0; j < HN; ++j) {
                const int n = tid + THREADS * j;
                hn[j] = (n < L) ? p.tapers[(int64_t)(k + 1) * L + n] : 0.f;
            }
        }
        __syncthreads();     // taper k visible; post of k-1 (and, first time, the tile reads) done
        MT_TICK(1);
        float2 a[16], o[16];
        if (!(p.dbg & 2)) {
        // pass 1: radix 16, P = 1, inputs straight from the window tile
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int n = i + t * TPF;
            float h;
            if constexpr (LONG) h = hl[t]; else h = (n < L) ? hkk[n] : 0.f;
            a[t] = make_float2(xs[t].x * h, xs[t].y * h);
        }
        dft16(a, o);
#pragma unroll
        for (int u = 0; u < 16; ++u) zf[PHYS(16 * i + u)] = o[u];
        XBAR();
        if constexpr (LOG2N < 8) {
            // N = 16 R2 (R2 = 4, 8): pass 2 is radix R2 with P = 16 -- thread i takes the outputs u = i + R2 b:
            // X[u + 16 v] = sum_j W_N^(j u) W_R2^(j v) Y[j][u], in place (it writes back exactly the slots it read)
            constexpr int R2 = N / 16;
#pragma unroll
            for (int b = 0; b < 16 / R2; ++b) {
                const int u = i + R2 * b;
                float2 q[R2];
#pragma unroll
                for (int j = 0; j < R2; ++j) {
                    const float2 v = zf[PHYS(16 * j + u)];
                    q[j] = (j == 0) ? v : cmul(v, TW(j * u));
                }
                if constexpr (R2 == 4) dft4r(q[0], q[1], q[2], q[3]); else dft8r(q);
#pragma unroll
                for (int v = 0; v < R2; ++v) a[b * R2 + v] = q[v];
            }
#pragma unroll
            for (int b = 0; b < 16 / R2; ++b) {
                const int u = i + R2 * b;
#pragma unroll
                for (int v = 0; v < R2; ++v) zf[PHYS(u + 16 * v)] = a[b * R2 + v];
            }
            __syncthreads();
        } else
        // pass 2: radix 16, P = 16
        {
            const int kk = i & 15;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float2 v = zf[PHYS(i + t * TPF)];
                a[t] = (t == 0) ? v : cmul(v, TW(t * kk * (N / 256)));
            }
            dft16(a, o);
            if constexpr (LOG2N != 8) XBAR();      // N = 256 writes back exactly the slots it read
            const int j = ((i - kk) << 4) + kk;
#pragma unroll
            for (int u = 0; u < 16; ++u) zf[PHYS(j + 16 * u)] = o[u];
            if constexpr (LOG2N == 8) __syncthreads(); else XBAR();
        }
        if constexpr (LOG2N == 9) {         // pass 3: radix 2, P = 256, eight butterflies per thread
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int ib = i + b * TPF;            // 0..255, k = ib
                const float2 u0 = zf[PHYS(ib)], u1 = cmul(zf[PHYS(ib + 256)], TW(ib));
                a[2 * b] = make_float2(u0.x + u1.x, u0.y + u1.y);
                a[2 * b + 1] = make_float2(u0.x - u1.x, u0.y - u1.y);
            }
            // in place: every thread writes back exactly the slots it read
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int ib = i + b * TPF;
                zf[PHYS(ib)] = a[2 * b];
                zf[PHYS(ib + 256)] = a[2 * b + 1];
            }
            __syncthreads();
        }
        if constexpr (LOG2N == 11) {        // pass 3: radix 8, P = 256, two butterflies per thread
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ib = i + b * TPF;            // 0..255, k = ib
                float2 q[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float2 v = zf[PHYS(ib + t * 256)];
                    q[t] = (t == 0) ? v : cmul(v, TW(t * ib));
                }
                dft8r(q);
#pragma unroll
                for (int u = 0; u < 8; ++u) a[8 * b + u] = q[u];
            }
            // in place: every thread writes back exactly the slots it read
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ib = i + b * TPF;
#pragma unroll
                for (int u = 0; u < 8; ++u) zf[PHYS(ib + 256 * u)] = a[8 * b + u];
            }
            __syncthreads();
        }
        if constexpr (LOG2N == 10) {        // pass 3: radix 4, P = 256, four butterflies per thread
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ib = i + b * TPF;            // 0..255, k = ib
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float2 v = zf[PHYS(ib + t * 256)];
                    a[4 * b + t] = (t == 0) ? v : cmul(v, TW(t * ib));
                }
                dft4r(a[4 * b], a[4 * b + 1], a[4 * b + 2], a[4 * b + 3]);
            }
            // in place: every thread writes back exactly the slots it read
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ib = i + b * TPF;
#pragma unroll
                for (int u = 0; u < 4; ++u) zf[PHYS(ib + 256 * u)] = a[4 * b + u];
            }
            __syncthreads();
        }
        if constexpr (LOG2N == 12) {        // pass 3: radix 16, P = 256, k = i
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float2 v = zf[PHYS(i + t * TPF)];
                a[t] = (t == 0) ? v : cmul(v, TW(t * i));
            }
            dft16(a, o);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 16; ++u) zf[PHYS(i + 256 * u)] = o[u];
            __syncthreads();
        }
        } else { __syncthreads(); }
        if (fetch_next) {
            float* hnext = hk + ((k + 1) & TWO) * L;
#pragma unroll
            for (int j = 0; j < HN; ++j) {
                const int n = tid + THREADS * j;
                if (n < L) hnext[n] = hn[j];
            }
        }
        MT_TICK(2);
        // split the packed pair, store X[f][w][r][k][c..c+1]
        if (p.dbg & 4) continue;
        if constexpr (PL && !LONG && NF >= 4) {
            {
                // Planes format: a thread takes FOUR channel pairs (8 channels) of one frequency, so that every plane leaves
                // as one 16-byte store (16 lanes = a 256-byte tile row of the four planes): eight LDS reads, the
                // conjugate-symmetry split, the two-piece f16 split (the channel scales are already on the samples).
                // TR (four or more groups of eight channels per workgroup, i.e. N <= 256): the four lanes of a quad take four CONSECUTIVE frequencies of one
                // channel group and transpose their 4 planes x 4 frequencies before the stores, so that store a of a lane is
                // plane (lane & 3) of frequency a of the quad: the 16 lanes of four quads write one whole 256-byte tile row per
                // instruction, where plane-per-instruction stores left 64-byte pieces of sixteen rows (measured with a
                // wrong-contents variant of the same volume: 1.94 -> 1.80 ms at cfg3).
                constexpr int NG = NF / 4, FSTEP = THREADS / NG;
                constexpr bool TR = NG >= 4;
                const int j = tid & 3;
                // (Round 5, tried and dropped: starting the four 16-lane rows of a wave 16 frequencies apart instead of 4 -- a bank distance of
                //  17 instead of 4 in the skewed exchange buffer for the eight LDS reads of a lane -- changed nothing: 1.92 / 1.93 against
                //  1.91 / 1.94 ms per transform incl. the scale pass, A/B of two libraries on one box.  The store loop is not bank-bound.)
                const int grp = TR ? (tid >> 2) % NG : tid % NG, fq = TR ? (tid / (4 * NG)) * 4 + j : tid / NG, cg = c0 + 8 * grp;
                const int64_t row0 = ((int64_t)w * p.R + r) * p.K + k, rows_f = (int64_t)p.W * p.R * p.K;
                const bool in_tile = cg < ((C + 31) & ~31);
                unsigned char* dst0 = p.P + row0 * p.row_bytes + (cg >> 5) * 256 + ((cg & 31) >> 3) * 16 + (TR ? j * 64 : 0);
                if (in_tile)
                for (int f = fq; (TR ? f - j : f) <= N / 2; f += FSTEP) {
                    u32x4_t rh, rm, ih, im;
                    float2 u1q[4], u2q[4];         // all eight LDS reads in flight before the first split (one wait instead of four round trips)
                    const int fr = f <= N / 2 ? f : N / 2;        // (TR: a lane past the Nyquist bin reads it again; its column is never stored)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float2* zp = z + (4 * grp + q) * ZS;
                        u1q[q] = zp[PHYS(fr)];
                        u2q[q] = zp[PHYS((N - fr) & (N - 1))];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float2 u1 = u1q[q], u2 = u2q[q];
                        float2 A = make_float2(u1.x + u2.x, u1.y - u2.y);       // (Z[f] + conj Z[N-f]) / 2, the half already in the samples
                        float2 B = make_float2(u1.y + u2.y, u2.x - u1.x);       // (Z[f] - conj Z[N-f]) / (2 i)
                        if (any_flag) {
                            const int pp = 2 * (4 * grp + q);
                            const bool qna = nbf[pp] != 0, qnb = nbf[pp + 1] != 0;
                            if (!qna && nzf[pp] == 0) A = make_float2(0.f, 0.f);
                            if (!qnb && nzf[pp + 1] == 0) B = make_float2(0.f, 0.f);
                            if (qna) A = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
                            if (qnb) B = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
                        }
                        unsigned h, m;
                        mt_split2(A.x, B.x, h, m);
                        rh[q] = h; rm[q] = m;
                        mt_split2(A.y, B.y, h, m);
                        ih[q] = h; im[q] = m;
                    }
                    if (p.dbg & 1) continue;
                    if constexpr (TR) {
                        // items (rh, rm, ih, im) = planes 0 .. 3 of this lane's frequency  ->  plane j of the quad's frequencies 0 .. 3
                        mt_quad_xchg<0xB1>(rh, rm, (j & 1) != 0);
                        mt_quad_xchg<0xB1>(ih, im, (j & 1) != 0);
                        mt_quad_xchg<0x4E>(rh, ih, (j & 2) != 0);
                        mt_quad_xchg<0x4E>(rm, im, (j & 2) != 0);
                        const int f0 = f - j;
                        unsigned char* dst = dst0 + (int64_t)f0 * rows_f * p.row_bytes;
                        const int64_t fs = rows_f * p.row_bytes;
                        *reinterpret_cast<u32x4_t*>(dst) = rh;
                        if (f0 + 1 <= N / 2) *reinterpret_cast<u32x4_t*>(dst + fs) = rm;
                        if (f0 + 2 <= N / 2) *reinterpret_cast<u32x4_t*>(dst + 2 * fs) = ih;
                        if (f0 + 3 <= N / 2) *reinterpret_cast<u32x4_t*>(dst + 3 * fs) = im;
                    } else {
                        unsigned char* dst = dst0 + (int64_t)f * rows_f * p.row_bytes;
                        *reinterpret_cast<u32x4_t*>(dst) = rh;
                        *reinterpret_cast<u32x4_t*>(dst + 64) = rm;
                        *reinterpret_cast<u32x4_t*>(dst + 128) = ih;
                        *reinterpret_cast<u32x4_t*>(dst + 192) = im;
                    }
                }
                continue;
            }
        }
        float2* Xk = p.X + (((int64_t)w * p.R + r) * p.K + k) * C + c0;
        // F * NF = 8 * 256 + NF outputs: eight full rounds (LDS reads batched four at a time ahead of the stores) and
        // the Nyquist row on the first NF threads.
        const int pr = tid & (NF - 1), fb = tid / NF, c = c0 + 2 * pr;
        if (c < C) {
            const float2* zp = z + pr * ZS;
            const bool last = tid < NF;
            float2 zn = make_float2(0.f, 0.f);
            if (last) zn = zp[PHYS(N / 2)];
            float2* dst0 = Xk + 2 * pr;
            const float ia = pair_scale(mxc[2 * pr], true), ib = pair_scale(mxc[2 * pr + 1], true);       // back to the samples' units
            // (the silent / non-finite channel overrides cost eight selects per row of a thread -- a tenth of the kernel's
            //  VALU instructions -- and almost never apply: a wave without a flagged channel takes the loop without them)
            auto put_any = [&](auto flagged, int f, float2 u1, float2 u2) {
                float2 A = make_float2((u1.x + u2.x) * ia, (u1.y - u2.y) * ia);       // (Z[f] + conj Z[N-f]) / 2, the half already in the samples
                float2 B = make_float2((u1.y + u2.y) * ib, (u2.x - u1.x) * ib);       // (Z[f] - conj Z[N-f]) / (2 i)
                if constexpr (decltype(flagged)::value) {
                    if (za) A = make_float2(0.f, 0.f);
                    if (zb) B = make_float2(0.f, 0.f);
                    if (na) A = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
                    if (nb) B = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
                }
                if ((p.dbg & 1) && A.x != 12345.f) return;
                if constexpr (ROWSTORE) {
                    float2* row = p.Z + (((((int64_t)w * p.Rc + (r - p.r_off)) * p.K + k) * C + c) * (int64_t)(N / 2 + 1));
                    row[f] = A;
                    if (c + 1 < C) row[(N / 2 + 1) + f] = B;
                    return;
                }
                float2* dst = dst0 + (int64_t)f * sF;
                if (vec_ok) {
                    // non-temporal only where the NF pairs of a frequency row make whole 128-byte lines (NF >= 8: up to 1024
                    // samples); the 64- / 32-byte pieces of 2048 / 4096 samples NEED the write-back L2 to merge them
                    // (4096 samples: 0.91 -> 1.73 ms with the hint)
                    if (NF < 8 || (p.dbg & 16)) *reinterpret_cast<float4*>(dst) = make_float4(A.x, A.y, B.x, B.y);
                    else sc_stream_store(dst, A, B);
                } else {
                    dst[0] = A;
                    if (c + 1 < C) dst[1] = B;
                }
            };
            auto store_all = [&](auto flagged) {
#pragma unroll
                for (int h = 0; h < 8; h += 4) {
                    float2 z1[4], z2[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int f = fb + (h + it) * (THREADS / NF);
                        z1[it] = zp[PHYS(f)];
                        z2[it] = zp[PHYS((N - f) & (N - 1))];
                    }
#pragma unroll
                    for (int it = 0; it < 4; ++it) put_any(flagged, fb + (h + it) * (THREADS / NF), z1[it], z2[it]);
                }
                if (last) put_any(flagged, N / 2, zn, zn);
            };
            if (any_flag) store_all(std::true_type{}); else store_all(std::false_type{});
        }
        MT_TICK(3);
        // the barrier at the top of the next taper orders these reads before pass 1 rewrites z
    }
#undef XBAR
#undef PHYS
}

__global__ void twiddle_kernel(float2* tw, int N) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= N) return;
    double s, c;
    sincospi(-2.0 * (double)m / (double)N, &s, &c);
    tw[m] = make_float2((float)c, (float)s);
}

extern "C" int sc_fft_twiddles_f32(int64_t N, void* d_tw, void* stream) {
    SC_REQUIRE(d_tw != nullptr && N >= 1, "bad twiddle request");
    hipLaunchKernelGGL(twiddle_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float2*)d_tw, (int)N);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

template <int LOG2N, int THREADS = 256, bool PL = false>
static int launch_mt16_(const MtArgs& a_in, hipStream_t stream) {
    constexpr int N = 1 << LOG2N;
    constexpr int TPF = N / 16, NF = THREADS / TPF, CT = 2 * NF;
    constexpr bool LONG = LOG2N >= 11;
    constexpr size_t xt_b = LONG ? 0 : (size_t)N * (CT + 2) * 4, z_b = (size_t)NF * (N + N / 16 + 1) * 8;
    constexpr size_t uni = (xt_b > z_b ? xt_b : z_b), red_b = (size_t)(2 * THREADS + 2 * CT) * 8;
    static_assert(LONG || (uni + (size_t)N * 16 <= 160 * 1024 && uni + red_b <= 160 * 1024), "LDS budget exceeded");
    static_assert(!LONG || (z_b >= (size_t)4 * THREADS * 8 && z_b + (64 + N / 64) * 8 <= 160 * 1024), "LDS budget exceeded");
    auto lds = [&](size_t kh, size_t L) { size_t t = (size_t)N * 8 + kh * L * 4; return uni + (t > red_b ? t : red_b); };
    // not resident = two buffers: the next taper is parked while the current one is in use
    // Keep all K tapers in LDS when that does not cost a resident workgroup per CU.
    constexpr size_t cu_lds = 160 * 1024 - 16 * ((CT * 4 + 15) / 16);     // minus the kernel's static zero-channel flags
    MtArgs a = a_in;
    const size_t one = lds(TPF <= 64 ? 2 : 1, a.L), all = lds(a.K, a.L);
    a.kh = (all <= cu_lds && cu_lds / all == cu_lds / one) ? a.K : 1;
    { const char* d = sc_switch(SC_SW_MTFFT_DEBUG); a.dbg = d ? atoi(d) : 0; }
    const size_t shmem = a.kh == a.K ? all : one;
    auto k = mtfft16_kernel<LOG2N, THREADS, PL>;
    if constexpr (LOG2N < 11) SC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    if constexpr (LOG2N >= 11) {
        // long windows: row-major spectra of a range of trials into a stream-ordered scratch (<= 2 GB), then one tiled
        // transpose per window into the frequency-major X
        const size_t lds_long = z_b + (64 + N / 64) * sizeof(float2);
        SC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_long));
        if constexpr (LOG2N == 11 || THREADS != 256) {
            // 2048 samples: 32-byte pieces of a frequency row per workgroup, and the four tiles that complete a 128-byte
            // line run back to back on one XCD -- its L2 merges them: 4.2 ms for the volume the row store + transpose
            // below takes 4.7 ms for.  (At 4096 samples, 16-byte pieces of eight tiles, the transpose still wins: 9.7
            // against 10.6 ms.)
            a.Z = nullptr; a.r_off = 0; a.Rc = a.R;
            const int64_t n_ct = (a.C + CT - 1) / CT, groups8 = ((int64_t)a.W * a.R + 7) / 8 * 8;
            hipLaunchKernelGGL(k, dim3((unsigned)(groups8 * n_ct)), dim3(THREADS), lds_long, stream, a);
            SC_CHECK_HIP(hipGetLastError());
            return SC_OK;
        }
        const int64_t F = N / 2 + 1, rows_trial = (int64_t)a.K * a.C, batch = (int64_t)a.W * a.R * rows_trial;
        int64_t rc = ((int64_t)2 << 30) / ((int64_t)a.W * rows_trial * F * 8);
        rc = rc < 1 ? 1 : (rc > a.R ? a.R : rc);
        float2* Z = nullptr;
        if (hipMallocAsync((void**)&Z, (size_t)(a.W * rc * rows_trial * F) * sizeof(float2), stream) != hipSuccess) {
            sc_set_error("multitaper FFT (N=%d): scratch allocation failed", N);
            return SC_ENOMEM;
        }
        a.Z = Z;
        int rc_ret = SC_OK;
        for (int64_t r0 = 0; r0 < a.R && rc_ret == SC_OK; r0 += rc) {
            const int64_t n = a.R - r0 < rc ? a.R - r0 : rc;
            a.r_off = (int)r0; a.Rc = (int)n;
            const int64_t n_ct = (a.C + CT - 1) / CT, groups8 = (a.W * n + 7) / 8 * 8;
            hipLaunchKernelGGL(k, dim3((unsigned)(groups8 * n_ct)), dim3(THREADS), lds_long, stream, a);
            for (int64_t w = 0; w < a.W && rc_ret == SC_OK; ++w)
                rc_ret = sc_internal_rows_to_bins(Z + w * n * rows_trial * F, a.X, n * rows_trial, F, batch,
                                                  (w * a.R + r0) * rows_trial, N / 2, stream);
        }
        (void)hipFreeAsync(Z, stream);
        if (rc_ret != SC_OK) return rc_ret;
        SC_CHECK_HIP(hipGetLastError());
        return SC_OK;
    }
    if constexpr (LOG2N >= 10) {        // one-dimensional grid, channel tiles of a (window, trial) on one XCD (see the kernel)
        a.r_off = 0; a.Rc = a.R;
        const int64_t n_ct = (a.C + CT - 1) / CT, groups8 = ((int64_t)a.W * a.R + 7) / 8 * 8;
        hipLaunchKernelGGL(k, dim3((unsigned)(groups8 * n_ct)), dim3(THREADS), shmem, stream, a);
        SC_CHECK_HIP(hipGetLastError());
        return SC_OK;
    }
    dim3 grid((unsigned)((a.C + CT - 1) / CT), (unsigned)a.R, (unsigned)a.W);
    hipLaunchKernelGGL(k, grid, dim3(THREADS), shmem, stream, a);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

template <int LOG2N, int THREADS = 256>
static int launch_mt16(const MtArgs& a, hipStream_t stream) {
    if constexpr (LOG2N <= 10) { if (a.P) return launch_mt16_<LOG2N, THREADS, true>(a, stream); }
    return launch_mt16_<LOG2N, THREADS, false>(a, stream);
}

// Powers of two that bring a channel whose largest finite magnitude has the bit pattern mx into [1, 2) (and back): the two channels
// of a packed pair enter their shared complex transform at the same magnitude, so that the weaker one keeps its own float32
// rounding (see mtfft16_kernel).  1 for zero / denormal / huge maxima.
__device__ __forceinline__ float mt_pair_scale(unsigned mx, bool inverse) {
    const unsigned E = mx >> 23;
    return (E >= 1u && E <= 253u) ? __uint_as_float((inverse ? E : 254u - E) << 23) : 1.f;
}

// ----------------------------------------------------------------------------------------
// Window lengths off the power-of-two list whose only prime factors are 2, 3 and 5 (250, 500, 1000, 200, 300, 1500 ...:
// what scipy.fft.next_fast_len, transforms.py:1024-1036, hands the reference for the usual sampling rates): the same
// fusion -- window extraction + detrend + taper + real FFT + transposed store in ONE kernel -- with a mixed-radix
// Stockham transform in LDS instead of the register-resident radix-16 passes.  rocFFT took these lengths in three
// passes over HBM (tapered windows out, transform, transposed copy: 8.8 ms at 250 samples for the cfg3 volume).
// Workgroup = (window, trial, tile of CT = 2 NF channels), 512 threads.  The detrended window tile stays in LDS for all
// tapers; per taper the NF packed channel pairs (z = x_c h + i x_{c+1} h) go through log-many passes of radix 5 / 4 / 3 /
// 2 butterflies between two LDS buffers (autosort: natural order out, no bit reversal), twiddles from an LDS copy of
// the fp64-rounded table exp(-2 pi i m / N), then the pair is separated by conjugate symmetry and stored like the
// radix-16 kernel stores (DC / Nyquist exactly real).
struct MxArgs {
    const float* x;
    const float* tapers;   // [K][L], already divided by fs
    const float2* tw;      // [N] exp(-2 pi i m / N)
    float2* X;             // [F][W][R][K][C]
    int T, R, C, L, step, W, K, detrend;
    int N, NF, n_pass;
    int radix[16];
    // planes-format output (mtfft_mixed_kernel only; see MtArgs): when P is set the spectra go there INSTEAD of X
    unsigned char* P;
    const float* scale;    // [C] powers of two
    int64_t row_bytes;
};
// two scaled reals (channels c, c + 1 of one component) -> their f16 pieces h and m = x - h, one dword each (sc_fused2.hip: f2_split2)
__device__ __forceinline__ void mx_split2(float x0, float x1, unsigned& h, unsigned& m) {
    typedef _Float16 mx_h2 __attribute__((ext_vector_type(2)));
    const mx_h2 hv = {(_Float16)x0, (_Float16)x1};
    const mx_h2 mv = {(_Float16)(x0 - (float)hv[0]), (_Float16)(x1 - (float)hv[1])};
    h = __builtin_bit_cast(unsigned, hv);
    m = __builtin_bit_cast(unsigned, mv);
}

__device__ __forceinline__ float2 mx_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 mx_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// one radix-R butterfly of a Stockham pass: inputs src[b + t m] (twiddled by W^(t k tw_step), k = b mod Ls), DFT_R in registers
template <int R>
__device__ __forceinline__ void mx_bfly_compute(const float2* __restrict__ src, const float2* __restrict__ tw, int b, int m,
                                                int Ls, int tw_step, float2 (&v)[R]) {
    const int k = b % Ls;
#pragma unroll
    for (int t = 0; t < R; ++t) {
        v[t] = src[b + t * m];
        if (t > 0 && Ls > 1) v[t] = cmul(v[t], tw[t * k * tw_step]);
    }
    if constexpr (R == 2) {
        const float2 a = v[0], c = v[1];
        v[0] = mx_add(a, c); v[1] = mx_sub(a, c);
    } else if constexpr (R == 3) {
        constexpr float S3 = 0.86602540378443865f;
        const float2 s = mx_add(v[1], v[2]), d = mx_sub(v[1], v[2]);
        const float2 t = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
        v[0] = mx_add(v[0], s);
        v[1] = make_float2(t.x + S3 * d.y, t.y - S3 * d.x);      // t - i S3 d
        v[2] = make_float2(t.x - S3 * d.y, t.y + S3 * d.x);      // t + i S3 d
    } else if constexpr (R == 4) {
        dft4r(v[0], v[1], v[2], v[3]);
    } else {
        constexpr float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f;
        constexpr float S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;
        const float2 a1 = mx_add(v[1], v[4]), a2 = mx_add(v[2], v[3]), b1 = mx_sub(v[1], v[4]), b2 = mx_sub(v[2], v[3]);
        const float2 p1 = make_float2(v[0].x + C1 * a1.x + C2 * a2.x, v[0].y + C1 * a1.y + C2 * a2.y);
        const float2 p2 = make_float2(v[0].x + C2 * a1.x + C1 * a2.x, v[0].y + C2 * a1.y + C1 * a2.y);
        const float2 q1 = make_float2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
        const float2 q2 = make_float2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
        v[0] = mx_add(v[0], mx_add(a1, a2));
        v[1] = make_float2(p1.x + q1.y, p1.y - q1.x);            // p1 - i q1
        v[4] = make_float2(p1.x - q1.y, p1.y + q1.x);            // p1 + i q1
        v[2] = make_float2(p2.x + q2.y, p2.y - q2.x);            // p2 - i q2
        v[3] = make_float2(p2.x - q2.y, p2.y + q2.x);            // p2 + i q2
    }
}
// ... and its outputs: dst[(b - k) R + k + t Ls] (autosort: natural order after the last pass)
template <int R>
__device__ __forceinline__ void mx_bfly_store(float2* __restrict__ dst, int b, int Ls, const float2 (&v)[R]) {
    const int k = b % Ls, base = (b - k) * R + k;
#pragma unroll
    for (int t = 0; t < R; ++t) dst[base + t * Ls] = v[t];
}
template <int R>
__device__ __forceinline__ void mx_butterfly(const float2* __restrict__ src, float2* __restrict__ dst, const float2* __restrict__ tw,
                                             int b, int m, int Ls, int tw_step) {
    float2 v[R];
    mx_bfly_compute<R>(src, tw, b, m, Ls, tw_step, v);
    mx_bfly_store<R>(dst, b, Ls, v);
}

// A transform whose length is known at compile time, ONE WAVE per packed channel pair, in place: every lane takes the
// butterflies b = lane, lane + 64, ... of the pass into registers, the wave meets (LDS executes a wave's instructions in
// order: no workgroup barrier), and writes them back to the same buffer.  Radix, butterfly count and twiddle stride are
// constants: the index arithmetic compiles to multiply-shift sequences and the rounds unroll.
template <int N, int LS>
__device__ __forceinline__ void mx_passes_wave(float2* z, const float2* tw, int lane) {
    if constexpr (LS < N) {
        constexpr int rem = N / LS;
        constexpr int R = rem % 5 == 0 ? 5 : (rem % 4 == 0 ? 4 : (rem % 3 == 0 ? 3 : 2));
        constexpr int m = N / R, tw_step = N / (LS * R), ROUNDS = (m + 63) / 64;
        float2 v[ROUNDS][R];
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int b = lane + 64 * j;
            if (b < m) mx_bfly_compute<R>(z, tw, b, m, LS, tw_step, v[j]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int b = lane + 64 * j;
            if (b < m) mx_bfly_store<R>(z, b, LS, v[j]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        mx_passes_wave<N, LS * R>(z, tw, lane);
    }
}

// The passes of a transform whose length is known at compile time: radix, butterfly count and twiddle stride are
// constants, the index arithmetic of every butterfly (q / m, b % Ls) compiles to multiply-shift sequences and the loops
// unroll.  (With run-time N the same kernel spends most of its instructions on integer division: 8.0 ms against
// rocFFT's 8.7 at 250 samples; the common lengths are instantiated, the rest take the run-time version.)
template <int N, int LS>
__device__ __forceinline__ float2* mx_passes(float2* src, float2* dst, const float2* tw, int NF, int tid) {
    if constexpr (LS >= N) {
        return src;
    } else {
        constexpr int rem = N / LS;
        constexpr int R = rem % 5 == 0 ? 5 : (rem % 4 == 0 ? 4 : (rem % 3 == 0 ? 3 : 2));
        constexpr int m = N / R, tw_step = N / (LS * R);
        const int total = NF * m;
        for (int q = tid; q < total; q += 512) {
            const int pr = q / m, b = q - pr * m;
            mx_butterfly<R>(src + pr * N, dst + pr * N, tw, b, m, LS, tw_step);
        }
        __syncthreads();
        return mx_passes<N, LS * R>(dst, src, tw, NF, tid);
    }
}

template <int NT>
__global__ void __launch_bounds__(512) mtfft_mixed_kernel(MxArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = NT > 0 ? NT : p.N;
    const int NF = p.NF, CT = 2 * NF, XS = CT + 2, L = p.L, C = p.C;
    const int lnf = 31 - __clz(NF), lct = lnf + 1;                // NF and CT are powers of two: shifts, not divisions
    float2* zA = reinterpret_cast<float2*>(smem);                 // [NF][N]
    float2* zB = zA + NF * N;                                     // [NF][N]
    float2* tw = zB + NF * N;                                     // [N]
    float* tile = reinterpret_cast<float*>(tw + N);               // [L][XS]
    double* red = reinterpret_cast<double*>(tile + ((L * XS + 1) & ~1));   // [2][512] + trend [2][CT]
    __shared__ int nzf[32], nbf[32];                              // channel not identically zero / holds a non-finite sample (see mtfft16_kernel)
    __shared__ unsigned mxc[32];                                  // largest finite |detrended sample| of the channel in this window
    const int tid = threadIdx.x;
    if (tid < 32) { nzf[tid] = 0; nbf[tid] = 0; mxc[tid] = 0u; }
    const int c0 = blockIdx.x * CT, r = blockIdx.y, w = blockIdx.z;
    const int64_t RC = (int64_t)p.R * C;
    const float* xw = p.x + ((int64_t)w * p.step * p.R + r) * C + c0;
    for (int idx = tid; idx < L * CT; idx += 512) {
        const int l = idx >> lct, cc = idx & (CT - 1);
        tile[l * XS + cc] = (c0 + cc < C) ? xw[(int64_t)l * RC + cc] : 0.f;
    }
    for (int i = tid; i < N; i += 512) tw[i] = p.tw[i];
    __syncthreads();
    if (p.detrend != SC_DETREND_NONE) {
        // trend sums in fp64, SL slices of the window per channel, then one thread per channel
        const int SL = 512 >> lct, cc = tid & (CT - 1), sl = tid >> lct;
        double s = 0.0, st = 0.0;
        if (sl < SL)
            for (int l = sl; l < L; l += SL) {
                const double v = (double)tile[l * XS + cc];
                s += v;
                st += v * (double)(l + 1);
            }
        red[tid] = s;
        red[512 + tid] = st;
        __syncthreads();
        if (tid < CT) {
            double sum = 0.0, sumt = 0.0;
            for (int q = 0; q < SL; ++q) { sum += red[q * CT + tid]; sumt += red[512 + q * CT + tid]; }
            sumt /= (double)L;
            const double n = (double)L;
            double a = 0.0, b;
            if (p.detrend == SC_DETREND_CONSTANT) {
                b = sum / n;
            } else {            // least-squares line on abscissa (l + 1) / L  (transforms.py:1903-1909)
                const double St = (n + 1.0) * 0.5, Stt = (n + 1.0) * (2.0 * n + 1.0) / (6.0 * n);
                const double den = n * Stt - St * St;
                a = (den != 0.0) ? (n * sumt - St * sum) / den : 0.0;
                b = (sum - a * St) / n;
            }
            red[1024 + tid] = a;
            red[1024 + CT + tid] = b;
        }
        __syncthreads();
        const double invL = 1.0 / (double)L;
        for (int idx = tid; idx < L * CT; idx += 512) {
            const int l = idx >> lct, cc2 = idx & (CT - 1);
            const double tt = (double)(l + 1) * invL;
            tile[l * XS + cc2] = (float)((double)tile[l * XS + cc2] - (red[1024 + cc2] * tt + red[1024 + CT + cc2]));
        }
        __syncthreads();
    }
    {
        const int SL = 512 >> lct, cc = tid & (CT - 1), sl = tid >> lct;
        // (integer tests on the bit patterns: OR of the magnitudes != 0 <=> some sample is not zero; largest magnitude with
        //  the exponent field all ones <=> NaN or infinity)
        unsigned orv = 0u, mxv = 0u;
        if (sl < SL)
            for (int l = sl; l < L; l += SL) { const unsigned u = __float_as_uint(tile[l * XS + cc]) & 0x7fffffffu; orv |= u; mxv = mxv > u ? mxv : u; }
        const bool nz = orv != 0u, bad = mxv >= 0x7f800000u;
        if (nz) nzf[cc] = 1;
        if (bad) nbf[cc] = 1;
        if (nz && !bad) atomicMax(&mxc[cc], mxv);
        __syncthreads();
        // a channel with a NaN / infinity leaves the packed transform (see mtfft16_kernel): zeros in, NaN bins out
        if (sl < SL && nbf[cc]) {
            for (int l = sl; l < L; l += SL) tile[l * XS + cc] = 0.f;
            nzf[cc] = 0;           // the store loop writes exact zeros for it; the fix-up after it writes the NaNs
        } else if (sl < SL) {      // pair normalisation (mt_pair_scale): exact, undone at the store
            const float sc = mt_pair_scale(mxc[cc], false);
            if (sc != 1.f)
                for (int l = sl; l < L; l += SL) tile[l * XS + cc] *= sc;
        }
        __syncthreads();
    }
    bool any_bad = false;
    for (int q = 0; q < CT; ++q) any_bad |= nbf[q] != 0;          // uniform over the workgroup
    const int F = N / 2 + 1;
    const int64_t sF = (int64_t)p.W * p.R * p.K * C;
    const bool vec_ok = (C % 2) == 0;
    for (int k = 0; k < p.K; ++k) {
        const float* hk = p.tapers + (int64_t)k * L;
        for (int idx = tid; idx < NF * N; idx += 512) {
            const int pr = idx / N, n = idx - pr * N;
            float2 v = make_float2(0.f, 0.f);
            if (n < L) {
                const float h = hk[n];
                const float2 xv = *reinterpret_cast<const float2*>(tile + n * XS + 2 * pr);
                v = make_float2(xv.x * h, xv.y * h);
            }
            zA[idx] = v;
        }
        __syncthreads();
        float2* src = zA;
        if constexpr (NT > 0) {
            src = mx_passes<(NT > 0 ? NT : 2), 1>(zA, zB, tw, NF, tid);
        } else {
            float2* dst = zB;
            int Ls = 1;
            for (int ps = 0; ps < p.n_pass; ++ps) {
                const int R = p.radix[ps], m = N / R, tw_step = N / (Ls * R);
                for (int q = tid; q < NF * m; q += 512) {
                    const int pr = q / m, b = q - pr * m;
                    const float2* s0 = src + pr * N;
                    float2* d0 = dst + pr * N;
                    if (R == 5) mx_butterfly<5>(s0, d0, tw, b, m, Ls, tw_step);
                    else if (R == 4) mx_butterfly<4>(s0, d0, tw, b, m, Ls, tw_step);
                    else if (R == 3) mx_butterfly<3>(s0, d0, tw, b, m, Ls, tw_step);
                    else mx_butterfly<2>(s0, d0, tw, b, m, Ls, tw_step);
                }
                __syncthreads();
                float2* t = src; src = dst; dst = t;
                Ls *= R;
            }
        }
        // split the packed pair: A[f] = (Z[f] + conj Z[N-f]) / 2, B[f] = (Z[f] - conj Z[N-f]) / (2 i); store X[f][w][r][k][c..c+1]
        float2* Xk = p.X + (((int64_t)w * p.R + r) * p.K + k) * C + c0;
        // planes-format output (round 6: every 2^a 3^b 5^c length this kernel takes -- 384, 768, 96 ... -- so that the matrix-pipe stage B
        // serves them): row [f][w][r][k], tile of 32 channels, planes Re h, Re m, Im h, Im m of 64 bytes; a channel pair is one dword per
        // plane.  The workgroup that holds the last channel also zeroes the rest of its 32-channel tile (stage B stages whole tiles).
        const int64_t prow0 = ((int64_t)w * p.R + r) * p.K + k, prowF = (int64_t)p.W * p.R * p.K;
        const int c_pad = (c0 + CT >= C) ? ((C + 31) & ~31) : 0;     // zero channels [C, c_pad) (C is even on this path)
        for (int idx = tid; idx < F * NF; idx += 512) {
            const int f = idx >> lnf, pr = idx & (NF - 1), c = c0 + 2 * pr;
            if (c >= C) continue;
            const float2 u1 = src[pr * N + f], u2 = src[pr * N + (f == 0 ? 0 : N - f)];
            const float ha = 0.5f * mt_pair_scale(mxc[2 * pr], true), hb = 0.5f * mt_pair_scale(mxc[2 * pr + 1], true);
            float2 A = make_float2(ha * (u1.x + u2.x), ha * (u1.y - u2.y));
            float2 B = make_float2(hb * (u1.y + u2.y), hb * (u2.x - u1.x));
            if (nzf[2 * pr] == 0) A = make_float2(0.f, 0.f);         // identically zero channel: exact zeros
            if (nzf[2 * pr + 1] == 0) B = make_float2(0.f, 0.f);
            if (p.P) {
                const float qn = __int_as_float(0x7fc00000);
                if (nbf[2 * pr]) A = make_float2(qn, qn);             // a NaN / infinity in the channel: its bins are NaN
                if (nbf[2 * pr + 1]) B = make_float2(qn, qn);
                const float sa = p.scale[c], sb = p.scale[c + 1];
                unsigned h, m;
                unsigned* d = reinterpret_cast<unsigned*>(p.P + (prow0 + (int64_t)f * prowF) * p.row_bytes + (c >> 5) * 256 + (c & 31) * 2);
                mx_split2(A.x * sa, B.x * sb, h, m);
                d[0] = h; d[16] = m;
                mx_split2(A.y * sa, B.y * sb, h, m);
                d[32] = h; d[48] = m;
                continue;
            }
            float2* d = Xk + (int64_t)f * sF + 2 * pr;
            if (vec_ok) {
                *reinterpret_cast<float4*>(d) = make_float4(A.x, A.y, B.x, B.y);
            } else {
                d[0] = A;
                if (c + 1 < C) d[1] = B;
            }
        }
        if (p.P && c_pad > C) {
            const int npad = (c_pad - C) >> 1;                       // channel pairs to zero per row
            for (int idx = tid; idx < F * npad; idx += 512) {
                const int f = idx / npad, c = C + 2 * (idx - f * npad);
                unsigned* d = reinterpret_cast<unsigned*>(p.P + (prow0 + (int64_t)f * prowF) * p.row_bytes + (c >> 5) * 256 + (c & 31) * 2);
                d[0] = 0u; d[16] = 0u; d[32] = 0u; d[48] = 0u;
            }
        }
        if (any_bad && !p.P) {       // rare: a channel of this tile held a NaN / infinity -- its bins become NaN (same thread, same
                             // addresses as the store loop above, so the order is the program's)
            const float qn = __int_as_float(0x7fc00000);
            for (int idx = tid; idx < F * NF; idx += 512) {
                const int f = idx >> lnf, pr = idx & (NF - 1), c = c0 + 2 * pr;
                if (c >= C) continue;
                float2* d = Xk + (int64_t)f * sF + 2 * pr;
                if (nbf[2 * pr]) d[0] = make_float2(qn, qn);
                if (nbf[2 * pr + 1] && c + 1 < C) d[1] = make_float2(qn, qn);
            }
        }
        __syncthreads();     // the next taper refills zA
    }
}

// The common lengths up to 1000 samples: N and the number of packed pairs NF are compile-time, ONE WAVE per pair runs the
// whole transform in place on its own slice of a single LDS buffer (mx_passes_wave), so the passes need no workgroup
// barrier; only the conjugate-symmetry split + store (coalesced across the pairs of a frequency row) and the refill
// for the next taper are fenced by one each.  64 NF threads per workgroup.
template <int N, int NF>
__global__ void __launch_bounds__(64 * NF) mtfft_mixed_wave_kernel(MxArgs p) {
    constexpr int NT = 64 * NF, CT = 2 * NF, XS = CT + 2, F = N / 2 + 1;
    constexpr int LCT = CT == 32 ? 5 : (CT == 16 ? 4 : (CT == 8 ? 3 : 2)), LNF = LCT - 1;
    extern __shared__ __align__(16) unsigned char smem[];
    float2* z = reinterpret_cast<float2*>(smem);                  // [NF][N]; the detrend scratch aliases it
    double* red = reinterpret_cast<double*>(smem);                // [2][NT] + trend [2][CT]
    float2* tw = z + (NF * N > (2 * NT + 2 * CT) ? NF * N : (2 * NT + 2 * CT));   // [N]
    float* tile = reinterpret_cast<float*>(tw + N);               // [L][XS]
    __shared__ int nzf[CT], nbf[CT];                              // channel not identically zero / holds a non-finite sample (see mtfft16_kernel)
    __shared__ unsigned mxc[CT];                                  // largest finite |detrended sample| of the channel in this window
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < CT) { nzf[tid] = 0; nbf[tid] = 0; mxc[tid] = 0u; }
    const int L = p.L, C = p.C;
    int c0, r, w;
    if constexpr (NF < 8) {
        // 64-byte pieces of a frequency row per workgroup: the two channel tiles that complete a 128-byte line run back
        // to back on ONE XCD (block b runs on XCD b % 8), whose L2 merges them -- see mtfft16_kernel
        const int n_ct = (C + CT - 1) / CT;
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int g = (j / n_ct) * 8 + xcd;
        if (g >= p.W * p.R) return;
        c0 = (j % n_ct) * CT; w = g / p.R; r = g - w * p.R;
    } else {
        c0 = blockIdx.x * CT; r = blockIdx.y; w = blockIdx.z;
    }
    const int64_t RC = (int64_t)p.R * C;
    const float* xw = p.x + ((int64_t)w * p.step * p.R + r) * C + c0;
    for (int idx = tid; idx < L * CT; idx += NT) {
        const int l = idx >> LCT, cc = idx & (CT - 1);
        tile[l * XS + cc] = (c0 + cc < C) ? xw[(int64_t)l * RC + cc] : 0.f;
    }
    for (int i = tid; i < N; i += NT) tw[i] = p.tw[i];
    __syncthreads();
    if (p.detrend != SC_DETREND_NONE) {
        constexpr int SL = NT / CT;
        const int cc = tid & (CT - 1), sl = tid >> LCT;
        double s = 0.0, st = 0.0;
        for (int l = sl; l < L; l += SL) {
            const double v = (double)tile[l * XS + cc];
            s += v;
            st += v * (double)(l + 1);
        }
        red[tid] = s;
        red[NT + tid] = st;
        __syncthreads();
        if (tid < CT) {
            double sum = 0.0, sumt = 0.0;
            for (int q = 0; q < SL; ++q) { sum += red[q * CT + tid]; sumt += red[NT + q * CT + tid]; }
            sumt /= (double)L;
            const double n = (double)L;
            double a = 0.0, b;
            if (p.detrend == SC_DETREND_CONSTANT) {
                b = sum / n;
            } else {            // least-squares line on abscissa (l + 1) / L  (transforms.py:1903-1909)
                const double St = (n + 1.0) * 0.5, Stt = (n + 1.0) * (2.0 * n + 1.0) / (6.0 * n);
                const double den = n * Stt - St * St;
                a = (den != 0.0) ? (n * sumt - St * sum) / den : 0.0;
                b = (sum - a * St) / n;
            }
            red[2 * NT + tid] = a;
            red[2 * NT + CT + tid] = b;
        }
        __syncthreads();
        const double invL = 1.0 / (double)L;
        for (int idx = tid; idx < L * CT; idx += NT) {
            const int l = idx >> LCT, cc2 = idx & (CT - 1);
            const double tt = (double)(l + 1) * invL;
            tile[l * XS + cc2] = (float)((double)tile[l * XS + cc2] - (red[2 * NT + cc2] * tt + red[2 * NT + CT + cc2]));
        }
        __syncthreads();                                          // the scratch is free: z takes its place
    }
    {
        constexpr int SL = NT / CT;
        const int cc = tid & (CT - 1), sl = tid >> LCT;
        // (integer tests on the bit patterns: OR of the magnitudes != 0 <=> some sample is not zero; largest magnitude with
        //  the exponent field all ones <=> NaN or infinity)
        unsigned orv = 0u, mxv = 0u;
        for (int l = sl; l < L; l += SL) { const unsigned u = __float_as_uint(tile[l * XS + cc]) & 0x7fffffffu; orv |= u; mxv = mxv > u ? mxv : u; }
        const bool nz = orv != 0u, bad = mxv >= 0x7f800000u;
        if (nz) nzf[cc] = 1;
        if (bad) nbf[cc] = 1;
        if (nz && !bad) atomicMax(&mxc[cc], mxv);
        __syncthreads();
        // a channel with a NaN / infinity leaves the packed transform (see mtfft16_kernel): zeros in, NaN bins out
        if (nbf[cc]) {
            for (int l = sl; l < L; l += SL) tile[l * XS + cc] = 0.f;
            nzf[cc] = 0;           // the store loop writes exact zeros for it; the fix-up below overwrites them
        } else {                   // pair normalisation (mt_pair_scale): exact, undone at the store
            const float sc = mt_pair_scale(mxc[cc], false);
            if (sc != 1.f)
                for (int l = sl; l < L; l += SL) tile[l * XS + cc] *= sc;
        }
        __syncthreads();
    }
    bool any_bad = false;
#pragma unroll
    for (int q = 0; q < CT; ++q) any_bad |= nbf[q] != 0;          // uniform over the workgroup
    const int64_t sF = (int64_t)p.W * p.R * p.K * C;
    const bool vec_ok = (C % 2) == 0;
    float2* zw = z + wave * N;                                    // this wave's pair
    for (int k = 0; k < p.K; ++k) {
        const float* hk = p.tapers + (int64_t)k * L;
        for (int n = lane; n < N; n += 64) {
            float2 v = make_float2(0.f, 0.f);
            if (n < L) {
                const float h = hk[n];
                const float2 xv = *reinterpret_cast<const float2*>(tile + n * XS + 2 * wave);
                v = make_float2(xv.x * h, xv.y * h);
            }
            zw[n] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        mx_passes_wave<N, 1>(zw, tw, lane);
        __syncthreads();                                          // every pair transformed
        float2* Xk = p.X + (((int64_t)w * p.R + r) * p.K + k) * C + c0;
        // (planes-format output: see mtfft_mixed_kernel)
        const int64_t prow0 = ((int64_t)w * p.R + r) * p.K + k, prowF = (int64_t)p.W * p.R * p.K;
        const int c_pad = (c0 + CT >= C) ? ((C + 31) & ~31) : 0;
        for (int idx = tid; idx < F * NF; idx += NT) {
            const int f = idx >> LNF, pr = idx & (NF - 1), c = c0 + 2 * pr;
            if (c >= C) continue;
            const float2 u1 = z[pr * N + f], u2 = z[pr * N + (f == 0 ? 0 : N - f)];
            const float ha = 0.5f * mt_pair_scale(mxc[2 * pr], true), hb = 0.5f * mt_pair_scale(mxc[2 * pr + 1], true);
            float2 A = make_float2(ha * (u1.x + u2.x), ha * (u1.y - u2.y));
            float2 B = make_float2(hb * (u1.y + u2.y), hb * (u2.x - u1.x));
            if (nzf[2 * pr] == 0) A = make_float2(0.f, 0.f);         // identically zero channel: exact zeros
            if (nzf[2 * pr + 1] == 0) B = make_float2(0.f, 0.f);
            if (p.P) {
                const float qn = __int_as_float(0x7fc00000);
                if (nbf[2 * pr]) A = make_float2(qn, qn);
                if (nbf[2 * pr + 1]) B = make_float2(qn, qn);
                const float sa = p.scale[c], sb = p.scale[c + 1];
                unsigned h, m;
                unsigned* d = reinterpret_cast<unsigned*>(p.P + (prow0 + (int64_t)f * prowF) * p.row_bytes + (c >> 5) * 256 + (c & 31) * 2);
                mx_split2(A.x * sa, B.x * sb, h, m);
                d[0] = h; d[16] = m;
                mx_split2(A.y * sa, B.y * sb, h, m);
                d[32] = h; d[48] = m;
                continue;
            }
            float2* d = Xk + (int64_t)f * sF + 2 * pr;
            if (vec_ok) {
                if constexpr (NF >= 8) sc_stream_store(d, A, B);        // whole 128-byte lines per frequency row
                else *reinterpret_cast<float4*>(d) = make_float4(A.x, A.y, B.x, B.y);
            } else {
                d[0] = A;
                if (c + 1 < C) d[1] = B;
            }
        }
        if (p.P && c_pad > C) {
            const int npad = (c_pad - C) >> 1;
            for (int idx = tid; idx < F * npad; idx += NT) {
                const int f = idx / npad, c = C + 2 * (idx - f * npad);
                unsigned* d = reinterpret_cast<unsigned*>(p.P + (prow0 + (int64_t)f * prowF) * p.row_bytes + (c >> 5) * 256 + (c & 31) * 2);
                d[0] = 0u; d[16] = 0u; d[32] = 0u; d[48] = 0u;
            }
        }
        if (any_bad && !p.P) {       // rare: a channel of this tile held a NaN / infinity -- its bins become NaN (same thread, same
                             // addresses as the store loop above, so the order is the program's)
            const float qn = __int_as_float(0x7fc00000);
            for (int idx = tid; idx < F * NF; idx += NT) {
                const int f = idx >> LNF, pr = idx & (NF - 1), c = c0 + 2 * pr;
                if (c >= C) continue;
                float2* d = Xk + (int64_t)f * sF + 2 * pr;
                if (nbf[2 * pr]) d[0] = make_float2(qn, qn);
                if (nbf[2 * pr + 1] && c + 1 < C) d[1] = make_float2(qn, qn);
            }
        }
        __syncthreads();                                          // the next taper refills z
    }
}

template <int N, int NF>
static int launch_mixed_wave(const MxArgs& m_in, hipStream_t stream) {
    MxArgs m = m_in;
    m.NF = NF;
    constexpr int NT = 64 * NF, CT = 2 * NF;
    const size_t zb = (size_t)(NF * N > (2 * NT + 2 * CT) ? NF * N : (2 * NT + 2 * CT)) * 8;
    const size_t lds = zb + (size_t)N * 8 + (size_t)m.L * (CT + 2) * 4 + 16;
    auto k = mtfft_mixed_wave_kernel<N, NF>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)((m.C + CT - 1) / CT), (unsigned)m.R, (unsigned)m.W);
    if (NF < 8) grid = dim3((unsigned)((((int64_t)m.W * m.R + 7) / 8 * 8) * ((m.C + CT - 1) / CT)));
    hipLaunchKernelGGL(k, grid, dim3(NT), lds, stream, m);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

// N = 2^a 3^b 5^c, not a power of two, 8 <= N <= 2048: the radix list (5s, 4s, 3s, then a 2) or 0 if N has another factor
static int mx_radices(int64_t N, int* radix) {
    if (N < 8 || N > 2048 || (N & (N - 1)) == 0) return 0;
    int n = 0;
    int64_t rem = N;
    while (rem % 5 == 0) { radix[n++] = 5; rem /= 5; }
    while (rem % 4 == 0) { radix[n++] = 4; rem /= 4; }
    while (rem % 3 == 0) { radix[n++] = 3; rem /= 3; }
    while (rem % 2 == 0) { radix[n++] = 2; rem /= 2; }
    return rem == 1 ? n : 0;
}

static int launch_mixed(const MtArgs& a, int64_t N, hipStream_t stream) {
    MxArgs m;
    m.x = a.x; m.tapers = a.tapers; m.tw = a.tw; m.X = a.X;
    m.T = a.T; m.R = a.R; m.C = a.C; m.L = a.L; m.step = a.step; m.W = a.W; m.K = a.K; m.detrend = a.detrend;
    m.N = (int)N;
    m.P = a.P; m.scale = a.scale; m.row_bytes = a.row_bytes;
    m.n_pass = mx_radices(N, m.radix);
    int nf = 16;
    while (nf > 2 && (int64_t)nf * N > 4096) nf >>= 1;
    if (N <= 512 && (int64_t)nf * N > 2048) nf >>= 1;       // short windows: 128-byte store segments are enough, two workgroups per CU
    while (nf > 1 && 2 * (nf >> 1) >= a.C) nf >>= 1;           // few channels: no wider than the data
    m.NF = nf;
    const int CT = 2 * nf;
    const size_t lds = (size_t)2 * nf * N * 8 + (size_t)N * 8 + (((size_t)a.L * (CT + 2) + 1) & ~(size_t)1) * 4 +
                       (size_t)(1024 + 2 * CT) * 8;
    if (lds > 160 * 1024 - 128) { sc_set_error("multitaper FFT (N=%lld): window tile does not fit LDS", (long long)N); return SC_EUNSUPPORTED; }
    switch (N) {        // one wave per channel pair, in place (mtfft_mixed_wave_kernel): the common lengths up to 1000 samples
    // (round 6: the 2^a 3 lengths -- 0.75 / 1.5 s at 128 ... 512 Hz -- with them; planes-format requests of 200 ... 1000 never come here)
    case 96: return launch_mixed_wave<96, 8>(m, stream);
    case 192: return launch_mixed_wave<192, 8>(m, stream);
    case 384: return launch_mixed_wave<384, 8>(m, stream);
    case 768: return launch_mixed_wave<768, 4>(m, stream);
    case 960: return launch_mixed_wave<960, 4>(m, stream);
    case 200: return launch_mixed_wave<200, 8>(m, stream);
    case 250: return launch_mixed_wave<250, 8>(m, stream);
    case 300: return launch_mixed_wave<300, 8>(m, stream);
    case 400: return launch_mixed_wave<400, 8>(m, stream);
    case 500: return launch_mixed_wave<500, 8>(m, stream);
    case 600: return launch_mixed_wave<600, 4>(m, stream);
    case 750: return launch_mixed_wave<750, 4>(m, stream);
    case 800: return launch_mixed_wave<800, 4>(m, stream);
    case 1000: return launch_mixed_wave<1000, 4>(m, stream);
    default: break;
    }
    dim3 grid((unsigned)((a.C + CT - 1) / CT), (unsigned)a.R, (unsigned)a.W);
#define MX_CASE(NN)                                                                                             \
    case NN:                                                                                                    \
        (void)hipFuncSetAttribute((const void*)mtfft_mixed_kernel<NN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(mtfft_mixed_kernel<NN>, grid, dim3(512), lds, stream, m);                            \
        break;
    switch (N) {        // the lengths next_fast_len hands out for window durations in round numbers of ms at 200 Hz ... 2 kHz
        MX_CASE(1200) MX_CASE(1250) MX_CASE(1500) MX_CASE(2000)
    default:
        MX_CASE(0)
    }
#undef MX_CASE
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

static int mt_wide() {
    const char* e = sc_switch(SC_SW_MTFFT_WIDE);
    return e ? atoi(e) : 1;
}

extern "C" int sc_multitaper_fft_planes_supported(int64_t L, int64_t N, int64_t C) {
    // (2048 and 4096 samples: the anti-phase kernel of sc_mtfft_long.hip only; N = 10 RM RF: sc_mtfft_mixed.hip only)
    if (!(L >= 1 && L <= N && C >= 2 && (C % 2) == 0)) return 0;
    if (N >= 64 && N <= 4096 && (N & (N - 1)) == 0) return 1;
    if (sc_internal_mtfft_mix_has(N)) return 1;
    int radix[16];
    return mx_radices(N, radix) > 0 ? 1 : 0;          // (round 6) every other 2^a 3^b 5^c <= 2048: the Stockham kernel's planes store
}

extern "C" int sc_multitaper_fft_supported(int64_t L, int64_t N) {
    if (L < 1 || L > N) return 0;
    if (N >= 64 && N <= 4096 && (N & (N - 1)) == 0) return 1;      // radix-16 kernel
    int radix[16];
    return mx_radices(N, radix) > 0 ? 1 : 0;                          // mixed-radix kernel: 2^a 3^b 5^c up to 2048
}

// planes != nullptr: the planes-format output of sc_multitaper_fft_planes_f32
static int mtfft_run(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L,
                     int64_t step, int64_t W, int64_t N, const float* d_tapers, int64_t K,
                     int detrend_type, const void* d_twiddles, void* d_X, void* d_P, const float* d_scale, void* stream) {
    ScTimed timed_("mtfft_fused", stream);
    SC_REQUIRE(d_x && d_tapers && d_twiddles && (d_X || d_P), "NULL device pointer");
    SC_REQUIRE(T >= 1 && R >= 1 && C >= 1 && L >= 1 && step >= 1 && W >= 1 && K >= 1, "dimensions must be positive");
    SC_REQUIRE((W - 1) * step + L <= T, "windows exceed the time series");
    SC_REQUIRE(detrend_type >= 0 && detrend_type <= 2, "unknown detrend_type");
    SC_REQUIRE(R <= 65535 && W <= 65535, "too many trials/windows for one launch");
    if (!sc_multitaper_fft_supported(L, N)) {
        sc_set_error("fused multitaper FFT needs L <= N and N a power of two in 64 ... 4096 or 2^a 3^b 5^c in 8 ... 2048 "
                     "(got L=%lld N=%lld); use sc_taper_windows_f32 + sc_fft_execute", (long long)L, (long long)N);
        return SC_EUNSUPPORTED;
    }
    MtArgs a{d_x, d_tapers, (const float2*)d_twiddles, (float2*)d_X, (int)T, (int)R, (int)C, (int)L,
             (int)step, (int)W, (int)K, detrend_type};
    a.P = (unsigned char*)d_P; a.scale = d_scale; a.row_bytes = 256 * ((C + 31) / 32);
    hipStream_t s = (hipStream_t)stream;
    // long windows with enough work to fill the chip -- and every planes-format request beyond 1024 samples: anti-phase
    // half-workgroups (sc_mtfft_long.hip)
    const bool pow2 = (N & (N - 1)) == 0;
    const bool use_long = pow2 && (sc_internal_mtfft_long_applies(N, C, W * R) || (d_P && N >= 2048));
    if (d_P) {
        if (!sc_multitaper_fft_planes_supported(L, N, C)) {
            sc_set_error("planes-format multitaper FFT needs an even number of signals and N a power of two in 64 ... 4096 or "
                         "2^a 3^b 5^c in 8 ... 2048 (got C=%lld N=%lld)",
                         (long long)C, (long long)N);
            return SC_EUNSUPPORTED;
        }
        // A workgroup writes the CT channels it transforms (round-1..3 kernels: CT = 2 * threads / (N / 16), 16 at 512 samples and
        // with the 512-thread workgroups of 1024, 8 with 256 threads at 1024); where the workgroups do not cover the last 32-channel
        // tile the uncovered part must read as zeros (stage B stages whole tiles)
        const int64_t threads = (N == 1024 && mt_wide() && C >= 16) ? 512 : 256, ct = pow2 ? 2 * threads / (N / 16) : 1;
        // (the Stockham kernel of the other 2^a 3^b 5^c lengths zeroes the rest of its last tile itself)
        const int64_t covered = !pow2 ? (sc_internal_mtfft_mix_has(N) ? sc_internal_mtfft_mix_coverage(N, C, true) : (C + 31) / 32 * 32)
                                      : (use_long ? sc_internal_mtfft_long_coverage(N, C) : (C + ct - 1) / ct * ct);
        if (covered < (C + 31) / 32 * 32)
            SC_CHECK_HIP(hipMemsetAsync(d_P, 0, (size_t)((N / 2 + 1) * W * R * K) * (size_t)a.row_bytes, s));
    }
    // the lengths N = 10 RM RF with enough work to fill the chip -- and every planes-format request: sc_mtfft_mixed.hip
    if (!pow2 && sc_internal_mtfft_mix_has(N) && (d_P || sc_internal_mtfft_mix_applies(N, C, W * R)))
        return sc_internal_mtfft_mix(d_x, T, R, C, L, step, W, N, d_tapers, K, detrend_type, d_twiddles, d_X, d_P, d_scale, s);
    if (!pow2) return launch_mixed(a, N, s);
    if (use_long) return sc_internal_mtfft_long(d_x, T, R, C, L, step, W, N, d_tapers, K, detrend_type, d_twiddles, d_X, d_P, d_scale, s);
    switch (N) {
    case 64: return launch_mt16<6>(a, s);
    case 128: return launch_mt16<7>(a, s);
    case 256: return launch_mt16<8>(a, s);
    case 512: return launch_mt16<9>(a, s);
    // 512-thread workgroups from 1024 samples on: 16 / 8 / 4 channels per workgroup, i.e. 128- / 64- / 32-byte pieces of a
    // frequency row per store group (2.5 -> 2.7, 1.6 -> 2.1, 1.3 -> 1.8 TB/s at N = 1024 / 2048 / 4096, the last without the
    // row store + transpose pass; tools/stage_a_wide.py).  SC_MTFFT_WIDE=0 (diagnostic): the 256-thread workgroups.
    case 1024: return mt_wide() && C >= 16 ? launch_mt16<10, 512>(a, s) : launch_mt16<10>(a, s);
    case 2048: return mt_wide() && C >= 8 ? launch_mt16<11, 512>(a, s) : launch_mt16<11>(a, s);
    case 4096: return mt_wide() && C >= 4 ? launch_mt16<12, 512>(a, s) : launch_mt16<12>(a, s);
    }
    return SC_EUNSUPPORTED;
}

extern "C" int sc_multitaper_fft_f32(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L,
                                     int64_t step, int64_t W, int64_t N, const float* d_tapers, int64_t K,
                                     int detrend_type, const void* d_twiddles, void* d_X, void* stream) {
    SC_REQUIRE(d_X, "NULL device pointer");
    return mtfft_run(d_x, T, R, C, L, step, W, N, d_tapers, K, detrend_type, d_twiddles, d_X, nullptr, nullptr, stream);
}

extern "C" int sc_multitaper_fft_planes_f32(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L,
                                            int64_t step, int64_t W, int64_t N, const float* d_tapers, int64_t K,
                                            int detrend_type, const void* d_twiddles, const float* d_scale, void* d_P,
                                            void* stream) {
    SC_REQUIRE(d_P && d_scale, "NULL device pointer");
    return mtfft_run(d_x, T, R, C, L, step, W, N, d_tapers, K, detrend_type, d_twiddles, nullptr, d_P, d_scale, stream);
}
