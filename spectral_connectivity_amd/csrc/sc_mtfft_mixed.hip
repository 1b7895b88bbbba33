// sc_mtfft_mixed.hip -- stage A for the window lengths next_fast_len hands out in an ordinary lab setting that are NOT powers of
// two: N = 10 * RM * RF (100, 150, 160, 200, 240, 250, 300, 320, 360, 400, 450, 480, 500, 600, 750, 800, 900, 1000, 1200, 1250, 1500,
// 1600, 1800, 2000 samples: 0.1 ... 2 s windows at 1 kHz, round durations at 100 Hz ... 2 kHz): window extraction + detrend + DPSS taper multiply + real FFT + transposed store of the one-sided spectra
// X[f][w][r][k][c] -- or of the planes format of sc_fused2.hip, so that the matrix-pipe stage B serves these lengths too
// (reference: transforms.py:1147-1171 sliding windows, :1311-1405 _multitaper_fft with n_fft = next_fast_len(L) :1024-1036,
//  :1798-1915 detrend).
//
// Why a third kernel.  The round-2 kernel for these lengths (mtfft_mixed_wave_kernel, sc_mtfft.hip) runs a Stockham pass per prime-ish
// factor IN LDS -- five read + write sweeps of the sequence per transform at 1000 samples, four at 250 -- with one wave per channel
// pair and 4 ... 8 waves per compute unit: 1.1 TB/s of spectra at 1000 samples, 2.4-2.5 at 200 / 250, where the power-of-two
// lengths reach 3.5-4.4; and it has no planes output, which left stage B on the round-3 kernels (5.0 against 3.5 ms).
// Here the transform is the register-resident scheme of sc_mtfft_long.hip with a radix-10 first pass: N / 10 threads per packed
// transform (two real channels), every thread owns the ten samples i + t N/10 of its pair for ALL tapers, and a transform is
//   pass 1   radix 10 in registers, its ten outputs to ten contiguous slots of the exchange buffer (five 16-byte pieces 80 bytes
//            apart across the lanes: conflict-free),
//   pass 2   radix RM (10, 5 or 2: a divisor of 10, so a thread's ten values are 10 / RM whole butterflies), twiddles from a
//            RM x 10 table,
//   pass 3   radix RF (anything small: 2 ... 25), IN PLACE -- every thread writes back the slots it read --, twiddles W_N^(t b)
//            from a table laid out [t][b] so that a thread's RF factors sit at constant offsets from one base;
// two exchanges instead of four or five sweeps.  A workgroup is two halves in ANTI-PHASE (sc_mtfft_long.hip): half 0 runs the passes
// of taper k while half 1 splits and stores its taper k - 1, then they swap.  ALIGNED: a transform never straddles a wave (N / 10
// <= 64: 64 / (N / 10) transforms per wave, the other lanes idle) -- the exchanges then need no workgroup barrier, one barrier per
// slot; otherwise the lanes are packed and the passes' barriers are matched by barriers between the store chunks of the other half.
// Arithmetic as in sc_mtfft.hip: two real channels per complex sequence, pair normalised per window by powers of two (planes
// output: the channel scales on the samples), halved samples, fp64 trend sums in a fixed order, twiddles from the fp64-rounded table.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "sc_common.h"
#include "sc_mtfft_bfly.h"

struct MixArgs {
    const float* x;        // [T][R][C]
    const float* tapers;   // [K][L], already divided by fs
    const float2* tw;      // [N] exp(-2 pi i m / N)
    float2* X;             // [F][W][R][K][C]
    int R, C, L, step, W, K, detrend;
    int vec;               // rows can be read in 16-byte pieces (C % 4 == 0, x 16-byte aligned)
    unsigned char* P;      // planes-format output (see sc_mtfft_long.hip); when set the spectra go there INSTEAD of X
    const float* scale;    // [C] powers of two
    int64_t row_bytes;
    int dbg;               // SC_MTFFT_DEBUG (results WRONG when set): 1 or 4 = no split / store loop, 2 = no passes; 8 = no priority for the
                           // passes, 32 = no super-tiles (results right)
};

// ---- small DFTs with compile-time twiddles -------------------------------------------------------------------------------------
namespace mr {
constexpr double PI = 3.141592653589793238462643383279502884;
constexpr double tsin(double x) {            // |x| <= pi / 4
    const double x2 = x * x;
    double term = x, s = x;
    for (int n = 1; n < 14; ++n) { term *= -x2 / (double)((2 * n) * (2 * n + 1)); s += term; }
    return s;
}
constexpr double tcos(double x) {
    const double x2 = x * x;
    double term = 1.0, s = 1.0;
    for (int n = 1; n < 14; ++n) { term *= -x2 / (double)((2 * n - 1) * (2 * n)); s += term; }
    return s;
}
// exp(-2 pi i j / R) = wre - i wim_pos ... returned as (cos, -sin); quadrant reduction in integers, so the multiples of a quarter
// turn are exact
constexpr double wre(int j, int R) {
    j %= R;
    const int m = (8 * j + R) / (2 * R);     // nearest quarter turn
    const double phi = 2.0 * PI * (double)(4 * j - m * R) / (double)(4 * R);
    switch (m & 3) {
    case 0: return tcos(phi);
    case 1: return -tsin(phi);
    case 2: return -tcos(phi);
    default: return tsin(phi);
    }
}
constexpr double wim(int j, int R) {         // imaginary part of exp(-2 pi i j / R) = -sin
    j %= R;
    const int m = (8 * j + R) / (2 * R);
    const double phi = 2.0 * PI * (double)(4 * j - m * R) / (double)(4 * R);
    switch (m & 3) {
    case 0: return -tsin(phi);
    case 1: return -tcos(phi);
    case 2: return tsin(phi);
    default: return tcos(phi);
    }
}
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float2 add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

template <int R> struct Fact {               // R = A * B: A-point transforms first
    static constexpr int A = (R % 5 == 0 && R > 5) ? 5 : (R % 4 == 0 && R > 4) ? 4 : (R % 3 == 0 && R > 3) ? 3 : (R % 2 == 0 && R > 2) ? 2 : R;
};

// forward DFT of R points, natural order in and out, in place
template <int R>
__device__ __forceinline__ void dft(float2 (&v)[R]) {
    if constexpr (R == 1) {
    } else if constexpr (R == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = add(a, b); v[1] = sub(a, b);
    } else if constexpr (R == 3) {
        constexpr float S3 = 0.86602540378443865f;
        const float2 s = add(v[1], v[2]), d = sub(v[1], v[2]);
        const float2 t = make_float2(v[0].x - 0.5f * s.x, v[0].y - 0.5f * s.y);
        v[0] = add(v[0], s);
        v[1] = make_float2(t.x + S3 * d.y, t.y - S3 * d.x);
        v[2] = make_float2(t.x - S3 * d.y, t.y + S3 * d.x);
    } else if constexpr (R == 4) {
        dft4r(v[0], v[1], v[2], v[3]);
    } else if constexpr (R == 5) {
        constexpr float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f;
        constexpr float S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;
        const float2 a1 = add(v[1], v[4]), a2 = add(v[2], v[3]), b1 = sub(v[1], v[4]), b2 = sub(v[2], v[3]);
        const float2 p1 = make_float2(v[0].x + C1 * a1.x + C2 * a2.x, v[0].y + C1 * a1.y + C2 * a2.y);
        const float2 p2 = make_float2(v[0].x + C2 * a1.x + C1 * a2.x, v[0].y + C2 * a1.y + C1 * a2.y);
        const float2 q1 = make_float2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
        const float2 q2 = make_float2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
        v[0] = add(v[0], add(a1, a2));
        v[1] = make_float2(p1.x + q1.y, p1.y - q1.x);
        v[4] = make_float2(p1.x - q1.y, p1.y + q1.x);
        v[2] = make_float2(p2.x + q2.y, p2.y - q2.x);
        v[3] = make_float2(p2.x - q2.y, p2.y + q2.x);
    } else {
        // Cooley-Tukey, t = t1 B + t2, u = u1 + A u2:  X[u1 + A u2] = sum_t2 W_B^(t2 u2) W_R^(t2 u1) sum_t1 x[t1 B + t2] W_A^(t1 u1)
        constexpr int A = Fact<R>::A, B = R / A;
        static_assert(A < R, "prime radix without a butterfly");
        float2 y[B][A];
        static_for<B>([&](auto t2c) {
            constexpr int t2 = decltype(t2c)::value;
            float2 s[A];
#pragma unroll
            for (int t1 = 0; t1 < A; ++t1) s[t1] = v[t1 * B + t2];
            dft<A>(s);
            static_for<A>([&](auto u1c) {
                constexpr int u1 = decltype(u1c)::value;
                constexpr int e = (t2 * u1) % R;
                if constexpr (e == 0) {
                    y[t2][u1] = s[u1];
                } else if constexpr (4 * e == R) {
                    y[t2][u1] = make_float2(s[u1].y, -s[u1].x);               // x (-i)
                } else if constexpr (2 * e == R) {
                    y[t2][u1] = make_float2(-s[u1].x, -s[u1].y);
                } else if constexpr (4 * e == 3 * R) {
                    y[t2][u1] = make_float2(-s[u1].y, s[u1].x);               // x i
                } else {
                    constexpr float c = (float)wre(e, R), sn = (float)wim(e, R);
                    y[t2][u1] = cmulc(s[u1], c, sn);
                }
            });
        });
        static_for<A>([&](auto u1c) {
            constexpr int u1 = decltype(u1c)::value;
            float2 s[B];
#pragma unroll
            for (int t2 = 0; t2 < B; ++t2) s[t2] = y[t2][u1];
            dft<B>(s);
#pragma unroll
            for (int u2 = 0; u2 < B; ++u2) v[u1 + A * u2] = s[u2];
        });
    }
}
}  // namespace mr

typedef unsigned mx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned mx_u32x2 __attribute__((ext_vector_type(2)));
// two scaled reals -> the dwords of their leading and trailing f16 pieces (as in sc_mtfft_long.hip)
__device__ __forceinline__ void mx_split2(float x0, float x1, unsigned& h, unsigned& m) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(x0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(x1), "v"(h));
}

// Geometry of one instantiation, shared by the kernel and its launcher.
template <int N_, int RM_, int RF_, int HT_, bool ALIGNED_, int BP_, int ZPAD_, int RP_ = 10, bool GRP_ = false>
struct MixGeo {
    // RP: points per thread = radix of pass 1.  10 everywhere but where 20 puts a transform into ONE wave (1000 / 1200 samples: 50 / 60
    // threads per transform -- the exchanges then need no workgroup barrier, a half holds eight channel pairs instead of four, and a
    // half's store pieces are whole 128-byte lines)
    static constexpr int N = N_, RM = RM_, RF = RF_, HT = HT_, RP = RP_;
    static constexpr bool ALIGNED = ALIGNED_;
    static_assert(N == RP * (RM ? RM : 1) * RF, "N = RP RM RF");
    static_assert(RM == 0 || RP % RM == 0, "the middle radix divides the first");
    static_assert(RP % 2 == 0, "16-byte pieces of pass 1");
    static_assert(HT % 64 == 0, "a half is whole waves");
    static constexpr int TPF = N / RP;                               // threads per transform
    static constexpr int GL = ALIGNED ? (TPF <= 64 ? 64 : (TPF + 63) / 64 * 64) : TPF;       // lanes of a group of transforms
    static constexpr int TPG = (ALIGNED && TPF <= 64) ? 64 / TPF : 1;                         // transforms per group
    static constexpr int NF_RAW = (HT / GL) * TPG;
    static constexpr int NF = NF_RAW / 2 * 2;                         // transforms (channel pairs) per half
    static_assert(NF >= 2, "a half holds at least two channel pairs");
    static constexpr int CTH = 2 * NF, CT = 2 * CTH;                 // channels per half / per workgroup
    static constexpr int LS = RP * (RM ? RM : 1);                    // butterflies of the last pass = its input stride
    static constexpr bool WAVE_LOCAL = ALIGNED && TPF <= 64;
    // GROUP_LOCAL: a transform is GL / 64 WHOLE waves (lanes aligned, more than 64 threads per transform) that meet at a counter in
    // LDS instead of the workgroup barrier -- the exchanges of a transform then concern its own two to four waves only, a slot has ONE
    // workgroup barrier like the wave-local geometries, and the storing half is not chopped into chunks to match the passes' barriers
    static constexpr bool GROUP_LOCAL = GRP_ && ALIGNED && TPF > 64;
    static constexpr bool SLOT1 = WAVE_LOCAL || GROUP_LOCAL;         // one workgroup barrier per slot, the taper double-buffered
    static constexpr int NB = SLOT1 ? 1 : (RM ? 4 : 2);              // workgroup barriers of one slot
    // Exchange buffer of a transform (float2): phys(idx) = idx + (idx / LS) BP -- BP pad elements behind every block of LS = 10 RM.
    // What the pad is for: pass 2 writes runs of ten consecutive elements LS apart; with LS + BP = 10 (mod 16) the runs of a
    // 16-lane store group tile the 32 banks (BP = 6 at RM = 10 or 2; RM = 5 keeps 0: its runs start mid-decade), every other access
    // of the passes is unit-stride across the lanes or (pass 1) ten contiguous elements per lane, written as five 16-byte pieces 80
    // bytes apart -- conflict-free as they are.  ZPAD spreads the transforms of a half over the banks for the store loop.  Both from a
    // model of the LDS banks over every access of a slot (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the first layout -- idx + idx /
    // 10, 0.33 at 250 samples -- agreed with it to 2 %): profiles/r06_stage_a_mixed_lds.txt.
    static constexpr int BP = BP_;
    static_assert(BP % 2 == 0 && ZPAD_ % 2 == 0, "16-byte pieces");
    static constexpr int ZS = N + ((N - 1) / LS) * BP + ZPAD_;
    // super-tiles: the workgroups whose pieces complete a 128-byte line run back to back on one XCD -- 16 channels of the complex64
    // rows, 32 of the planes rows (a line there is two planes of a 32-channel tile)
    static constexpr int SUP = (CTH >= 16 || 16 % CTH != 0) ? 1 : 16 / CTH;
    static constexpr int SUP_PL = SUP;      // (32-channel super-tiles for the planes rows -- what the power-of-two kernel gained 0.6-1.1 ms from --
                                            //  changed nothing here at 250 ... 1200 samples and cost 0.15 ms at 800: r06_stage_a_planes_sup.txt)
    static constexpr int RS = CT + 2;                                // padded row of the window tile (floats)
    static constexpr size_t z_bytes = (size_t)2 * NF * ZS * 8;
    static constexpr size_t tile_bytes = (size_t)(N / 2) * RS * 4;
    static constexpr int G1Q = TPF <= 4 ? 1 : (TPF <= 16 ? 4 : (TPF <= 64 ? 8 : (TPF <= 144 ? 12 : 16)));   // trend sums: first-level run
    static constexpr int G1 = (TPF + G1Q - 1) / G1Q;
    static constexpr size_t red_bytes = (size_t)2 * NF * (TPF + G1) * 4 * 8;
    static constexpr size_t un_bytes0 = z_bytes > tile_bytes ? z_bytes : tile_bytes;
    static constexpr size_t un_bytes = ((un_bytes0 > red_bytes ? un_bytes0 : red_bytes) + 15) / 16 * 16;
    static constexpr size_t lds = un_bytes + (size_t)((RM ? RM * RP : 0) + N) * 8 + (size_t)N * 4 * (SLOT1 ? 2 : 1);
};

template <class GEO, bool PL>
__global__ void __launch_bounds__(2 * GEO::HT, 2 * GEO::HT <= 512 ? 4 : 1) mtfft_mix_kernel(MixArgs p) {
    constexpr int N = GEO::N, RM = GEO::RM, RF = GEO::RF, HT = GEO::HT, RP = GEO::RP;
    constexpr int THREADS = 2 * HT, TPF = GEO::TPF, GL = GEO::GL, TPG = GEO::TPG, NF = GEO::NF, CTH = GEO::CTH, CT = GEO::CT;
    constexpr bool WAVE_LOCAL = GEO::WAVE_LOCAL, GROUP_LOCAL = GEO::GROUP_LOCAL, SLOT1 = GEO::SLOT1;
    constexpr int NB = GEO::NB, ZS = GEO::ZS, LS = GEO::LS, SUP = PL ? GEO::SUP_PL : GEO::SUP, RS = GEO::RS, LINE_CH = 16;
    constexpr int F = N / 2 + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    float2* zall = reinterpret_cast<float2*>(smem);                               // [2][NF][ZS]
    float2* T2 = reinterpret_cast<float2*>(smem + GEO::un_bytes);                 // [RM][RP]  W_(10 RM)^(t k)
    float2* TF = T2 + (RM ? RM * RP : 0);                                         // [RF][LS]  W_N^(t b)
    float* tap = reinterpret_cast<float*>(TF + N);                                // [N] the taper in use (zeros from L on); WAVE_LOCAL: [2][N]
    __shared__ int nzf[CT], nbf[CT];
    __shared__ unsigned mxc[CT];
    __shared__ unsigned gcnt[16];                     // GROUP_LOCAL: arrivals at the group barriers, one counter per group of waves

    const int tid = threadIdx.x, half = tid / HT, ht = tid - half * HT;
    const int L = p.L, C = p.C, K = p.K;
    // items: as in sc_mtfft_long.hip -- workgroup b takes (window, trial, channel tile), the tiles of one (window, trial) on ONE XCD
    const bool sup = C > LINE_CH && SUP > 1 && !(p.dbg & 32);
    const int n_ct = sup ? (C + SUP * CT - 1) / (SUP * CT) * SUP : (C + CT - 1) / CT;
    int ch0[2], w, r;
    {
        const int m = blockIdx.x, xcd = m & 7, j = m >> 3, g = (j / n_ct) * 8 + xcd;
        if (g >= p.W * p.R) return;
        const int tile = j % n_ct;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            ch0[h] = sup ? (tile / SUP) * (SUP * CT) + h * (SUP * CTH) + (tile % SUP) * CTH : tile * CT + h * CTH;
        w = g / p.R; r = g - w * p.R;
    }
    const int chalf = ch0[half];

    // lane -> (transform of this half, butterfly index)
    const int grp = ht / GL, gl = ht - grp * GL, pf = grp * TPG + gl / TPF, i = gl % TPF;
    const bool valid = gl < TPG * TPF && pf < NF;
    const int pfc = valid ? pf : 0;                                              // (idle lanes compute on transform 0's addresses, never write)
    const int lp = 2 * (half * NF + pfc);
    float2* zh = zall + half * NF * ZS;
    float2* zf = zh + pfc * ZS;
    if (tid < CT) { nzf[tid] = 0; nbf[tid] = 0; mxc[tid] = 0u; }
    if (tid < 16) gcnt[tid] = 0u;
    const bool detr = p.detrend != SC_DETREND_NONE;
    auto pair_scale = [](unsigned mx, bool inverse) -> float {
        const unsigned E = mx >> 23;
        return (E >= 1u && E <= 253u) ? __uint_as_float((inverse ? E : 254u - E) << 23) : 1.f;
    };

    // ---- the window into registers: two half-window tiles [N / 2 rows][CT channels] through the exchange buffers ----
    float2 xs[RP];                                    // this thread's pass-1 inputs, all tapers: samples i + t TPF of its pair
    {
        constexpr int QR = CT / 4, V = CTH / 4;       // 16-byte pieces per row, per half
        constexpr int PIECES = (N / 2) * QR, PPT = (PIECES + THREADS - 1) / THREADS;
        static_assert(GEO::tile_bytes <= GEO::un_bytes, "tile does not fit");
        float* tile = reinterpret_cast<float*>(smem);
        const int64_t RC = (int64_t)p.R * C;
        const float* xw = p.x + ((int64_t)w * p.step * p.R + r) * C;
        float4 v[2][PPT];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int it = 0; it < PPT; ++it) {
                const int idx = tid + it * THREADS, row = idx / QR, q = idx - row * QR, c = ch0[q / V] + 4 * (q % V);
                const int n = hh * (N / 2) + row;
                v[hh][it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < PIECES && n < L && c < C) {
                    const float* src = xw + (int64_t)n * RC + c;
                    if (p.vec && c + 3 < C) {
                        v[hh][it] = *reinterpret_cast<const float4*>(src);
                    } else {
                        v[hh][it].x = src[0];
                        if (c + 1 < C) v[hh][it].y = src[1];
                        if (c + 2 < C) v[hh][it].z = src[2];
                        if (c + 3 < C) v[hh][it].w = src[3];
                    }
                }
            }
        }
        // the tables, while the window is in flight (every exponent below is < N: t < RF, b < LS, and t k RF < RM 10 RF)
        if constexpr (RM > 0) {
            for (int e = tid; e < RM * RP; e += THREADS) T2[e] = p.tw[(e / RP) * (e % RP) * RF];
        }
        for (int e = tid; e < N; e += THREADS) TF[e] = p.tw[(e / LS) * (e % LS)];
        for (int n = tid; n < N; n += THREADS) tap[n] = (n < L) ? p.tapers[n] : 0.f;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (hh == 1) __syncthreads();             // the reads of the first half are done
#pragma unroll
            for (int it = 0; it < PPT; ++it) {
                const int idx = tid + it * THREADS, row = idx / QR, q = idx - row * QR;
                if (idx < PIECES) {
                    float2* d = reinterpret_cast<float2*>(tile + row * RS + 4 * q);
                    d[0] = make_float2(v[hh][it].x, v[hh][it].y);
                    d[1] = make_float2(v[hh][it].z, v[hh][it].w);
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = (RP / 2) * hh; t < (RP / 2) * (hh + 1); ++t)
                xs[t] = *reinterpret_cast<const float2*>(tile + (i + (t - (RP / 2) * hh) * TPF) * RS + half * CTH + 2 * pfc);
        }
    }
    __syncthreads();                                  // the tile is consumed: the union is free
    double ab[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    if (detr) {
        // trend sums in fp64, in a fixed order: a thread's ten samples, runs of G1Q threads, then the G1 runs of the transform
        constexpr int G1Q = GEO::G1Q, G1 = GEO::G1;
        double* part = reinterpret_cast<double*>(smem);                           // [2 NF][TPF][4]
        double* runs = part + (size_t)2 * NF * TPF * 4;                           // [2 NF][G1][4]
        double ts[4] = {0.0, 0.0, 0.0, 0.0};          // sum x, sum x (l + 1) of channel a; the same of channel b
#pragma unroll
        for (int t = 0; t < RP; ++t) {
            const double l1 = (double)(i + t * TPF + 1);
            ts[0] += (double)xs[t].x; ts[1] += (double)xs[t].x * l1;
            ts[2] += (double)xs[t].y; ts[3] += (double)xs[t].y * l1;
        }
        double* mine = part + ((size_t)(half * NF + pfc) * TPF + i) * 4;
        if (valid) { mine[0] = ts[0]; mine[1] = ts[1]; mine[2] = ts[2]; mine[3] = ts[3]; }
        __syncthreads();
        if (valid && i < G1) {
            const double* src = part + ((size_t)(half * NF + pfc) * TPF + i * G1Q) * 4;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (int q = 0; q < G1Q && i * G1Q + q < TPF; ++q) { a0 += src[4 * q]; a1 += src[4 * q + 1]; a2 += src[4 * q + 2]; a3 += src[4 * q + 3]; }
            double* d = runs + ((size_t)(half * NF + pfc) * G1 + i) * 4;
            d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3;
        }
        __syncthreads();
        {
            const double* src = runs + (size_t)(half * NF + pfc) * G1 * 4;
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            for (int q = 0; q < G1; ++q) { s[0] += src[4 * q]; s[1] += src[4 * q + 1]; s[2] += src[4 * q + 2]; s[3] += src[4 * q + 3]; }
            const double n = (double)L;
            const double St = (n + 1.0) * 0.5, Stt = (n + 1.0) * (2.0 * n + 1.0) / (6.0 * n), den = n * Stt - St * St;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const double sum = s[2 * ch], sumt = s[2 * ch + 1] / n;
                double a = 0.0, b;
                if (p.detrend == SC_DETREND_CONSTANT) {
                    b = sum / n;
                } else {            // least-squares line on abscissa (l + 1) / L  (transforms.py:1903-1909)
                    a = (den != 0.0) ? (n * sumt - St * sum) / den : 0.0;
                    b = (sum - a * St) / n;
                }
                ab[ch][0] = a; ab[ch][1] = b;
            }
        }
        const double invL = 1.0 / (double)L;
#pragma unroll
        for (int t = 0; t < RP; ++t) {
            const double tt = (double)(i + t * TPF + 1) * invL;
            const float dx = (float)((double)xs[t].x - (ab[0][0] * tt + ab[0][1]));
            const float dy = (float)((double)xs[t].y - (ab[1][0] * tt + ab[1][1]));
            const bool in = i + t * TPF < L;          // (zero padding stays zero)
            xs[t].x = in ? dx : 0.f;
            xs[t].y = in ? dy : 0.f;
        }
        __syncthreads();                              // the sums are consumed: the exchange buffers are free
    }
    if (valid) {
        // flag 1: the channel is not identically zero; flag 2: it holds a NaN / infinity (such a channel leaves the packed
        // transform -- zeros in its place, its partner stays clean -- and its bins are written as NaN); largest finite magnitude
        // of every channel of this window for the pair normalisation (see sc_mtfft.hip)
        unsigned or0 = 0u, or1 = 0u, mx0 = 0u, mx1 = 0u;
#pragma unroll
        for (int t = 0; t < RP; ++t) {
            const unsigned u0 = __float_as_uint(xs[t].x) & 0x7fffffffu, u1 = __float_as_uint(xs[t].y) & 0x7fffffffu;
            or0 |= u0; or1 |= u1;
            mx0 = mx0 > u0 ? mx0 : u0; mx1 = mx1 > u1 ? mx1 : u1;
        }
        const bool n0 = or0 != 0u, n1 = or1 != 0u, b0 = mx0 >= 0x7f800000u, b1 = mx1 >= 0x7f800000u;
        if (n0) nzf[lp] = 1;
        if (n1) nzf[lp + 1] = 1;
        if (b0) nbf[lp] = 1;
        if (b1) nbf[lp + 1] = 1;
        if (!b0 && n0) atomicMax(&mxc[lp], mx0);
        if (!b1 && n1) atomicMax(&mxc[lp + 1], mx1);
    }
    __syncthreads();
    {
        if (nbf[lp]) {
#pragma unroll
            for (int t = 0; t < RP; ++t) xs[t].x = 0.f;
        }
        if (nbf[lp + 1]) {
#pragma unroll
            for (int t = 0; t < RP; ++t) xs[t].y = 0.f;
        }
        // halved (the 1/2 of the conjugate-symmetry split) and scaled into [1, 2) per channel: exact
        // (planes output: the channel scales -- powers of two -- go onto the samples instead)
        const int c = chalf + 2 * pfc;
        const float h0 = 0.5f * (PL ? (c < C ? p.scale[c] : 1.f) : pair_scale(mxc[lp], false));
        const float h1 = 0.5f * (PL ? (c + 1 < C ? p.scale[c + 1] : 1.f) : pair_scale(mxc[lp + 1], false));
#pragma unroll
        for (int t = 0; t < RP; ++t) { xs[t].x *= h0; xs[t].y *= h1; }
    }

    // ---- addresses of the passes: (one base per role) + constants wherever the geometry allows ----
    constexpr int BP = GEO::BP;
    auto phys = [](int idx) -> int { return BP ? idx + (idx / LS) * BP : idx; };                    // (LS is a constant: multiply + shift)
    float2* const zw1 = zf + phys(RP * i);            // pass 1 writes phys(RP i + u) = phys(RP i) + u: RP / 2 16-byte pieces
    // passes 2, 3 read phys(i + s TPF).  Where a block of LS elements is whole runs of TPF (LS % TPF == 0) the block of i + s TPF is
    // s / (LS / TPF) whatever i: a constant offset from zf + i, no register; otherwise one offset per s, computed once
    constexpr bool ZCONST = LS % TPF == 0;
    constexpr int NZ = ZCONST ? 1 : RP;
    int zr_tab[NZ];
    if constexpr (!ZCONST) {
#pragma unroll
        for (int s = 0; s < RP; ++s) zr_tab[s] = phys(i + s * TPF);
    }
    auto zr_off = [&](int s) -> int { return ZCONST ? i + s * TPF + (s / (ZCONST ? LS / TPF : 1)) * BP : zr_tab[ZCONST ? 0 : s]; };

    // The waves of one transform meet at a counter in LDS (GROUP_LOCAL): a wave's LDS instructions execute in order, so when the
    // partners see the arrival their exchange writes are there too; the counter only grows (no reset, compared modulo 2^32).
    unsigned epoch = 0u;
    unsigned* const gc = gcnt + half * 8 + grp;
    auto group_barrier = [&]() {
        epoch += (unsigned)(GL / 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((tid & 63) == 0) __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while ((int)(__hip_atomic_load(gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - epoch) < 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
#define XBAR()                                                      \
    do {                                                            \
        if constexpr (WAVE_LOCAL) {                                 \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
            __builtin_amdgcn_wave_barrier();                        \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
        } else if constexpr (GROUP_LOCAL) {                         \
            group_barrier();                                        \
        } else {                                                    \
            __syncthreads();                                        \
        }                                                           \
    } while (0)
    auto passes = [&](int k) {                        // NB workgroup barriers
        const float* tk = SLOT1 ? tap + (k & 1) * N + i : tap + i;
        if (p.dbg & 2) {
#pragma unroll
            for (int b = 0; b < NB; ++b) __syncthreads();
            return;
        }
        if (!(p.dbg & 8)) __builtin_amdgcn_s_setprio(1);      // the passes are the critical path of a slot: ahead of the storing half's VALU
        {
            float2 a[RP];
#pragma unroll
            for (int t = 0; t < RP; ++t) {
                const float h = tk[t * TPF];
                a[t] = make_float2(xs[t].x * h, xs[t].y * h);
            }
            mr::dft<RP>(a);
            if (valid) {
#pragma unroll
                for (int u = 0; u < RP; u += 2)
                    *reinterpret_cast<float4*>(zw1 + u) = make_float4(a[u].x, a[u].y, a[u + 1].x, a[u + 1].y);
            }
        }
        XBAR();                                                                       // 1
        if constexpr (RM > 0) {
            constexpr int J2 = RP / RM;               // butterflies of this thread: b = i + j TPF, inputs s = j + t J2
            float2 a[RP];
#pragma unroll
            for (int s = 0; s < RP; ++s) a[s] = zf[zr_off(s)];
            float2 o[J2][RM];
            int wbase[J2];
#pragma unroll
            for (int j = 0; j < J2; ++j) {
                const int b = i + j * TPF, kk = b % RP;
#pragma unroll
                for (int t = 0; t < RM; ++t) o[j][t] = (t == 0) ? a[j] : cmul(a[j + t * J2], T2[t * RP + kk]);
                mr::dft<RM>(o[j]);
                wbase[j] = (b - kk) / RP * (LS + BP) + kk;                          // phys((b - kk) RM + kk + t RP) = wbase + 10 t
            }
            XBAR();                                                                   // 2
            if (valid) {
#pragma unroll
                for (int j = 0; j < J2; ++j) {
#pragma unroll
                    for (int t = 0; t < RM; ++t) zf[wbase[j] + RP * t] = o[j][t];
                }
            }
            XBAR();                                                                   // 3
        }
        {
            // last pass: radix RF in place, butterflies b = i + j TPF < LS, slots phys(b + t LS) = b + t (LS + BP)
            constexpr int JF = (LS + TPF - 1) / TPF, TS = LS + BP;
#pragma unroll
            for (int j = 0; j < JF; ++j) {
                const int b = i + j * TPF;
                if (JF * TPF == LS || b < LS) {
                    float2* zb = zf + zr_off(j);      // phys(i + j TPF): j < JF <= RP
                    const float2* tf = TF + b;
                    float2 q[RF];
#pragma unroll
                    for (int t = 0; t < RF; ++t) {
                        const float2 v = zb[t * TS];
                        q[t] = (t == 0) ? v : cmul(v, tf[t * LS]);
                    }
                    mr::dft<RF>(q);
                    if (valid) {
#pragma unroll
                        for (int u = 0; u < RF; ++u) zb[u * TS] = q[u];
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();                                                              // NB
    };

    // ---- the store side ----
    const bool vec_ok = (C % 2) == 0;
    const int64_t sF = (int64_t)p.W * p.R * K * C;
    // complex64: lane -> (pair spr of this half, first bin fb); FSTEP bins per round
    constexpr int FSTEP = HT / NF, MR = (F + FSTEP - 1) / FSTEP;
    const int spr = ht % NF, fb = ht / NF, sl = 2 * (half * NF + spr), cs = chalf + 2 * spr;
    unsigned fl;                                      // bit 0 / 1: channel a / b non-finite, 2 / 3: identically zero
    {
        const bool na = nbf[sl] != 0, nb = nbf[sl + 1] != 0;
        fl = (na ? 1u : 0u) | (nb ? 2u : 0u) | ((!na && nzf[sl] == 0) ? 4u : 0u) | ((!nb && nzf[sl + 1] == 0) ? 8u : 0u);
    }
    const float ia = pair_scale(mxc[sl], true), ib = pair_scale(mxc[sl + 1], true);      // back to the samples' units
    float2* const Xi = p.X + ((int64_t)w * p.R + r) * K * C + cs;                         // X[0][w][r][0][cs]
    const float2* const zst = zh + spr * ZS;
    auto put = [&](float2* dst, float2 u1, float2 u2) {
        float2 A = make_float2((u1.x + u2.x) * ia, (u1.y - u2.y) * ia);       // (Z[f] + conj Z[N-f]) / 2, the half already in the samples
        float2 B = make_float2((u1.y + u2.y) * ib, (u2.x - u1.x) * ib);       // (Z[f] - conj Z[N-f]) / (2 i)
        if (fl & 4u) A = make_float2(0.f, 0.f);
        if (fl & 8u) B = make_float2(0.f, 0.f);
        if (fl & 1u) A = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
        if (fl & 2u) B = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
        if (vec_ok) {
            *reinterpret_cast<float4*>(dst) = make_float4(A.x, A.y, B.x, B.y);
        } else {
            dst[0] = A;
            if (cs + 1 < C) dst[1] = B;
        }
    };
    // planes format: a thread takes G channel pairs of one frequency, every plane leaves as one 16-byte (8-byte) store
    constexpr int G = NF % 4 == 0 ? 4 : 2, NG = NF / G, FSP = HT / NG, PIT = (F + FSP - 1) / FSP;
    const int pgrp = ht % NG, pfq = ht / NG, pcg = chalf + 2 * G * pgrp;
    const int pchunk = pfq * NB / FSP;                // the store chunk of this lane (planes output, several barriers per slot)
    bool pflag = false;                               // some channel of this thread's group is silent or non-finite
    if constexpr (PL) {
#pragma unroll
        for (int q = 0; q < 2 * G; ++q) {
            const int e = 2 * (half * NF + G * pgrp) + q;
            pflag = pflag || nbf[e] != 0 || nzf[e] == 0;
        }
        pflag = __builtin_amdgcn_ballot_w64(pflag) != 0ull;       // wave-uniform
    }
    auto put_planes = [&](int k, int it) {
        using VT = std::conditional_t<G == 4, mx_u32x4, mx_u32x2>;
        const int f = pfq + it * FSP;
        if (pfq >= FSP || f > N / 2 || !(pcg < ((C + 31) & ~31))) return;    // (absent channels of a started tile are written: zeros)
        const int64_t rows_f = (int64_t)p.W * p.R * K;
        unsigned char* dst = p.P + (((int64_t)w * p.R + r) * K + k) * p.row_bytes + (pcg >> 5) * 256 + (pcg & 31) * 2
                             + (int64_t)f * rows_f * p.row_bytes;
        float2 u1q[G], u2q[G];                        // all LDS reads in flight before the first split
        const float2* zg = zh + (G * pgrp) * ZS;
        const int n2 = f == 0 ? 0 : N - f;
        const int i1 = phys(f), i2 = phys(n2);
#pragma unroll
        for (int q = 0; q < G; ++q) { u1q[q] = zg[q * ZS + i1]; u2q[q] = zg[q * ZS + i2]; }
        VT rh, rm, ih, im;
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const float2 u1 = u1q[q], u2 = u2q[q];
            float2 A = make_float2(u1.x + u2.x, u1.y - u2.y);
            float2 B = make_float2(u1.y + u2.y, u2.x - u1.x);
            if (pflag) {
                const int e = 2 * (half * NF + G * pgrp + q);
                const bool qna = nbf[e] != 0, qnb = nbf[e + 1] != 0;
                if (!qna && nzf[e] == 0) A = make_float2(0.f, 0.f);
                if (!qnb && nzf[e + 1] == 0) B = make_float2(0.f, 0.f);
                if (qna) A = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
                if (qnb) B = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
            }
            unsigned h, m;
            mx_split2(A.x, B.x, h, m);
            rh[q] = h; rm[q] = m;
            mx_split2(A.y, B.y, h, m);
            ih[q] = h; im[q] = m;
        }
        *reinterpret_cast<VT*>(dst) = rh;
        *reinterpret_cast<VT*>(dst + 64) = rm;
        *reinterpret_cast<VT*>(dst + 128) = ih;
        *reinterpret_cast<VT*>(dst + 192) = im;
    };
    // One store slot: taper k leaves in NB chunks; half 0 parks taper k + 1.  NB workgroup barriers.
    constexpr int TPT = (N + HT - 1) / HT;            // taper values per thread of a half
    auto store = [&](int k) {
        float2* Xk = Xi + (int64_t)k * C;
        const bool live = cs < C && fb < FSTEP && !(p.dbg & 5);
        const bool park = half == 0 && k + 1 < K;
        float hn[TPT];
        if (park) {
            const float* tp = p.tapers + (int64_t)(k + 1) * L;
#pragma unroll
            for (int j = 0; j < TPT; ++j) hn[j] = (ht + j * HT < L) ? tp[ht + j * HT] : 0.f;
        }
#pragma unroll
        for (int ch = 0; ch < NB; ++ch) {
            if constexpr (PL) {
                // (every chunk stores the frequencies of one NB-th of the lanes: the stream is spread over the slot)
                if (!(p.dbg & 5) && (SLOT1 || pchunk == ch)) {
#pragma unroll
                    for (int it = 0; it < PIT; ++it) put_planes(k, it);
                }
            } else if (live) {
                constexpr int MC = (MR + NB - 1) / NB;                               // rounds per chunk
                float2 z1[MC], z2[MC];
#pragma unroll
                for (int e = 0; e < MC; ++e) {
                    const int f = fb + (ch * MC + e) * FSTEP;
                    const int fc = f <= N / 2 ? f : 0, n2 = fc == 0 ? 0 : N - fc;
                    z1[e] = zst[phys(fc)];
                    z2[e] = zst[phys(n2)];
                }
#pragma unroll
                for (int e = 0; e < MC; ++e) {
                    const int f = fb + (ch * MC + e) * FSTEP;
                    if (ch * MC + e < MR && f <= N / 2) put(Xk + (int64_t)f * sF, z1[e], z2[e]);
                }
            }
            if constexpr (!SLOT1) __syncthreads();
            if (ch == 0 && park) {
                // not WAVE_LOCAL: the other half has read the taper in use (its first interval), replace it; WAVE_LOCAL: the other buffer
                float* tn = SLOT1 ? tap + ((k + 1) & 1) * N : tap;
#pragma unroll
                for (int j = 0; j < TPT; ++j)
                    if (ht + j * HT < N) tn[ht + j * HT] = hn[j];
            }
        }
        if constexpr (SLOT1) __syncthreads();
    };

    // Slot q of a half: taper q / 2, the passes in the even slots and the store in the odd ones; half 1 is one slot behind half 0.
#pragma nounroll
    for (int gs = 0; gs <= 2 * K; ++gs) {
        const int q = gs - half;
        if (q < 0 || q >= 2 * K) {
#pragma unroll
            for (int b = 0; b < NB; ++b) __syncthreads();
        } else if (!(q & 1)) {
            passes(q >> 1);
        } else {
            store(q >> 1);
        }
    }
#undef XBAR
}

// ---- launch ---------------------------------------------------------------------------------------------------------------------
template <class GEO, bool PL>
static int launch_mix_(MixArgs a, hipStream_t st) {
    static_assert(GEO::lds + 1024 <= 160 * 1024, "LDS budget exceeded");
    auto k = mtfft_mix_kernel<GEO, PL>;
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEO::lds));
    constexpr int SUP = PL ? GEO::SUP_PL : GEO::SUP, CT = GEO::CT, LINE_CH = 16;
    const int64_t n_ct = (a.C > LINE_CH && SUP > 1 && !(a.dbg & 32)) ? (a.C + SUP * CT - 1) / (SUP * CT) * SUP : (a.C + CT - 1) / CT;
    const int64_t groups8 = ((int64_t)a.W * a.R + 7) / 8 * 8;
    if (groups8 * n_ct >= ((int64_t)1 << 31)) {
        sc_set_error("multitaper FFT (N=%d): too many windows x trials for one launch", GEO::N);
        return SC_EINVAL;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(groups8 * n_ct)), dim3(2 * GEO::HT), GEO::lds, st, a);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
template <class GEO>
static int launch_mix(const MixArgs& a, hipStream_t st) {
    return a.P ? launch_mix_<GEO, true>(a, st) : launch_mix_<GEO, false>(a, st);
}
template <class GEO, bool PL = true>
static int64_t coverage_of(int64_t C, int dbg) {
    constexpr int SUP = PL ? GEO::SUP_PL : GEO::SUP, CT = GEO::CT, LINE_CH = 16;
    return (C > LINE_CH && SUP > 1 && !(dbg & 32)) ? (C + SUP * CT - 1) / (SUP * CT) * SUP * CT : (C + CT - 1) / CT * CT;
}

// The instantiated lengths and their geometries (threads of a half, lanes aligned to waves or packed, block pad, transform pad);
// SC_MTFFT_MIXED_GEO=g takes geometry g of a length where it has one (A/B on MI355X: profiles/r06_stage_a_mixed_ab.txt, which also
// holds the geometries that were tried and taken out again: 128- and 256-thread halves -- two to four workgroups per compute unit --
// 2.4-2.5 TB/s where geometry 0 has 2.9-3.0 at 200 / 250 samples; six waves per SIMD with the registers capped at 80: 2.4-2.7; RP = 20
// at 500 ... 1200 samples -- one wave per transform, eight pairs per half -- 1.6-2.5 TB/s against 2.0-2.9: its radix-20 pass needs
// more than the 128 registers a 1024-thread workgroup has).  X(N, RM, RF, g, HT, ALIGNED, BP, ZPAD, RP, GRP)
#define MIX_GEOS(X)                                     \
    X(200, 10, 2, 0, 256, true, 6, 6, 10, false)                   \
    X(200, 10, 2, 1, 320, false, 6, 6, 10, false)                  \
    X(200, 10, 2, 2, 512, true, 6, 6, 10, false)                   \
    X(250, 5, 5, 0, 256, true, 0, 2, 10, false)                    \
    X(250, 5, 5, 1, 448, false, 0, 0, 10, false)                   \
    X(250, 5, 5, 2, 512, true, 0, 0, 10, false)                    \
    X(300, 10, 3, 0, 256, true, 6, 6, 10, false)                   \
    X(300, 10, 3, 1, 256, false, 6, 6, 10, false)                  \
    X(300, 10, 3, 2, 512, true, 6, 6, 10, false)                   \
    X(400, 10, 4, 0, 256, true, 6, 6, 10, false)                   \
    X(400, 10, 4, 1, 320, false, 6, 6, 10, false)                  \
    X(400, 10, 4, 2, 512, true, 6, 2, 10, false)                   \
    X(500, 10, 5, 0, 256, true, 6, 12, 10, false)                  \
    X(500, 10, 5, 1, 448, false, 6, 6, 10, false)                  \
    X(500, 10, 5, 2, 512, true, 6, 0, 10, false)                   \
    X(600, 10, 6, 0, 256, true, 6, 2, 10, false)                   \
    X(600, 10, 6, 1, 512, false, 6, 6, 10, false)                  \
    X(600, 10, 6, 2, 512, true, 6, 6, 10, false)                   \
    X(750, 5, 15, 0, 320, false, 0, 10, 10, false)                 \
    X(750, 5, 15, 1, 512, true, 0, 10, 10, false)                  \
    X(750, 5, 15, 2, 512, true, 0, 10, 10, true)                  \
    X(800, 10, 8, 0, 320, false, 6, 14, 10, false)                 \
    X(800, 10, 8, 1, 512, true, 6, 14, 10, false)                  \
    X(800, 10, 8, 2, 512, true, 6, 14, 10, true)                  \
    X(1000, 10, 10, 0, 512, true, 6, 10, 10, false)                \
    X(1000, 10, 10, 1, 448, false, 6, 10, 10, false)               \
    X(1000, 10, 10, 3, 512, true, 6, 10, 10, true)               \
    X(1200, 10, 12, 0, 512, true, 6, 6, 10, false)                 \
    X(1200, 10, 12, 1, 512, false, 6, 6, 10, false)                \
    X(1200, 10, 12, 2, 512, true, 6, 6, 10, true)                \
    X(1500, 10, 15, 0, 320, false, 6, 0, 10, false)                \
    X(1500, 10, 15, 1, 512, false, 6, 0, 10, false)                \
    X(2000, 10, 20, 0, 448, false, 6, 14, 10, false)               \
    X(2000, 10, 20, 1, 512, true, 6, 14, 10, false)             \
    X(1000, 10, 5, 2, 384, true, 12, 4, 20, false)             \
    X(2000, 10, 10, 2, 512, true, 2, 6, 20, false)             \
    X(100, 0, 10, 0, 256, true, 0, 6, 10, false)               \
    X(150, 5, 3, 0, 256, true, 0, 0, 10, false)                \
    X(160, 2, 8, 0, 256, true, 6, 0, 10, false)                \
    X(240, 2, 12, 0, 256, true, 8, 12, 10, false)              \
    X(320, 2, 16, 0, 256, true, 0, 4, 10, false)               \
    X(360, 2, 18, 0, 256, true, 0, 0, 10, false)               \
    X(360, 2, 18, 1, 512, true, 0, 0, 10, false)               \
    X(450, 5, 9, 0, 512, true, 0, 2, 10, false)                \
    X(480, 2, 24, 0, 384, true, 8, 4, 10, false)               \
    X(900, 10, 9, 0, 512, true, 6, 4, 10, false)               \
    X(900, 10, 9, 1, 512, true, 6, 4, 10, true)               \
    X(1250, 5, 25, 0, 384, true, 0, 6, 10, false)              \
    X(1600, 10, 16, 0, 320, false, 6, 22, 10, false)           \
    X(1800, 10, 18, 0, 384, false, 6, 2, 10, false)

static int mix_dbg() {
    const char* d = sc_switch(SC_SW_MTFFT_DEBUG);
    return d ? atoi(d) : 0;
}
static bool mix_has_geo(int64_t N, int g) {
    switch (N * 8 + g) {
#define X(NN, RM, RF, GI, HT, AL, BP, ZP, RP, GRP) case NN * 8 + GI: return true;
        MIX_GEOS(X)
#undef X
    }
    return false;
}
// The geometry of a launch: SC_MTFFT_MIXED_GEO if set (and the length has it), else by output -- the planes format wants at least
// eight channel pairs per half where the length allows it (16-byte pieces of a plane from four halves never merged into lines in
// the L2: 4.6 against 2.7 ms at 500 samples), complex64 output the wave-local geometry (profiles/r06_stage_a_mixed_ab.txt)
static int mix_geo(int64_t N, bool planes) {
    const char* e = sc_switch(SC_SW_MTFFT_MIXED_GEO);
    if (e) {
        const int g = atoi(e);
        return (g > 0 && g < 8 && mix_has_geo(N, g)) ? g : 0;
    }
    if (N == 2000) return 2;          // RP = 20: 2.6 / 3.0 ms against 3.1 / 3.9 (radix 20 as the FINAL pass of RP = 10 spills)
    // 400 ... 600 samples: one transform per wave, eight pairs per half, one workgroup of sixteen waves per compute unit (geometry 2):
    // 2.09 / 1.86 ms against 2.22 / 2.21 at 500 / 600 samples, and the planes output 2.18-2.48 against 2.5-3.1 ms
    if (N == 400 || N == 500 || N == 600) return 2;
    // two waves per transform meeting at a counter in LDS instead of the workgroup barrier (GROUP_LOCAL): the planes output gains
    // 0.2-0.7 ms at 750 ... 1200 samples (its store slot is no longer cut into chunks), the complex64 output only at 900 ... 1200
    if (N == 1000) return 3;
    if (N == 1200) return 2;
    if (N == 900) return 1;
    if (planes) return N == 300 ? 2 : ((N == 750 || N == 800) ? 2 : ((N == 1500 || N == 360) ? 1 : 0));
    return 0;
}

bool sc_internal_mtfft_mix_has(int64_t N) { return mix_has_geo(N, 0); }

template <class GEO>
static int64_t tiles_of(int64_t C, int dbg) { return coverage_of<GEO, false>(C, dbg) / GEO::CT; }

// SC_MTFFT_MIXED=0: never (the round-2 kernels); =1: whatever the size (tests); unset: when the launch gives every compute unit a
// workgroup (a workgroup here holds 16 ... 48 channels; a small problem fills the chip better with the one-wave-per-pair kernels)
bool sc_internal_mtfft_mix_applies(int64_t N, int64_t C, int64_t groups) {
    const char* e = sc_switch(SC_SW_MTFFT_MIXED);
    if (e && atoi(e) == 0) return false;
    if (!sc_internal_mtfft_mix_has(N) || C < 1) return false;
    if (e && atoi(e) == 1) return true;
    int64_t n_ct = 1;
    switch (N * 8 + mix_geo(N, false)) {
#define X(NN, RM, RF, GI, HT, AL, BP, ZP, RP, GRP) case NN * 8 + GI: n_ct = tiles_of<MixGeo<NN, RM, RF, HT, AL, BP, ZP, RP, GRP>>(C, mix_dbg()); break;
        MIX_GEOS(X)
#undef X
    }
    return groups * n_ct >= 256;
}

int64_t sc_internal_mtfft_mix_coverage(int64_t N, int64_t C, bool planes) {
    const int dbg = mix_dbg();
    switch (N * 8 + mix_geo(N, planes)) {
#define X(NN, RM, RF, GI, HT, AL, BP, ZP, RP, GRP) case NN * 8 + GI: return coverage_of<MixGeo<NN, RM, RF, HT, AL, BP, ZP, RP, GRP>>(C, dbg);
        MIX_GEOS(X)
#undef X
    }
    return 0;
}

int sc_internal_mtfft_mix(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L, int64_t step, int64_t W, int64_t N,
                          const float* d_tapers, int64_t K, int detrend_type, const void* d_twiddles, void* d_X, void* d_P,
                          const float* d_scale, hipStream_t st) {
    MixArgs a{};
    a.x = d_x; a.tapers = d_tapers; a.tw = (const float2*)d_twiddles; a.X = (float2*)d_X;
    a.P = (unsigned char*)d_P; a.scale = d_scale; a.row_bytes = 256 * ((C + 31) / 32);
    SC_REQUIRE(!d_P || (d_scale && C % 2 == 0), "planes output needs channel scales and an even number of signals");
    a.R = (int)R; a.C = (int)C; a.L = (int)L; a.step = (int)step; a.W = (int)W; a.K = (int)K; a.detrend = detrend_type;
    a.vec = (C % 4 == 0 && ((uintptr_t)d_x & 15) == 0) ? 1 : 0;
    a.dbg = mix_dbg();
    SC_REQUIRE((W - 1) * step + L <= T, "windows exceed the time series");
    switch (N * 8 + mix_geo(N, d_P != nullptr)) {
#define X(NN, RM, RF, GI, HT, AL, BP, ZP, RP, GRP) case NN * 8 + GI: return launch_mix<MixGeo<NN, RM, RF, HT, AL, BP, ZP, RP, GRP>>(a, st);
        MIX_GEOS(X)
#undef X
    }
    return SC_EUNSUPPORTED;
}
