// sc_mtfft_long.hip -- stage A for power-of-two windows of 256 ... 4096 samples (written for the long ones, hence the name): window
// extraction + detrend + DPSS taper multiply + real FFT + transposed store of the one-sided spectra X[f][w][r][k][c] -- or of the
// planes format of sc_fused2.hip -- like sc_mtfft.hip
// (reference: transforms.py:1147-1171 sliding windows, :1311-1405 _multitaper_fft, :1798-1915 detrend).
//
// Why a second kernel.  At the long lengths one packed transform (two channels) fills 35 KB of LDS, so a workgroup holds few
// channels, and the round-3 kernels (mtfft16_kernel<10 / 11 / 12>) ran their three phases one after the other on every compute
// unit (SC_MTFFT_DEBUG ablation at the cfg3 volume, N = 4096: 2.38 ms = 0.87 load + detrend, 0.75 passes, 0.66 stores):
//  * the long-window loads took 8 bytes of 64 different rows per wave instruction (x is [time][trial][channel]: a workgroup needs
//    16 bytes of every 512-byte row): 0.6 TB/s of samples;
//  * the workgroups of a compute unit fell into step: all in the passes (sharing the VALU), then all in the store loop
//    (sharing the memory pipe), so nothing overlapped.
// Here a workgroup is TWO halves of HT threads that run in ANTI-PHASE by construction: while half 0 runs the radix-16 passes
// of taper k (VALU + LDS), half 1 splits and stores its taper k - 1 (memory pipe), then they swap -- the slot boundary is a
// workgroup barrier both halves reach, and the passes' inner barriers are matched by barriers between the store chunks of the
// other half (up to 1024 samples: one wave per transform, no inner barriers at all).  HT = 512 from 1024 samples on: one workgroup
// (16 waves, 139 KB of exchange buffers: one per half) owns the compute unit; HT = 256 at 256 / 512 samples: two workgroups share
// it.  Each half transforms NF = HT / (N / 16) channel pairs, a workgroup 4 NF channels.  The window reaches the registers through
// the exchange buffers, which are free until the first pass: two row-major half-window tiles, loaded with 16-byte pieces of the
// rows and read back per thread.
// (First form of this kernel: the series transposed to [trial][channel][time] by a pass of its own -- 0.25 ms of the 1.6 at
//  the cfg3 volume -- and read as contiguous channel rows; and several items per workgroup with the next item's prologue under
//  the last store slot, which gained nothing at 2048 samples and LOST 40 % at 4096: there the four workgroups whose 32-byte
//  pieces complete a line must stay in step, and they only do when they are dispatched together.)
// Arithmetic as in sc_mtfft.hip (two real channels per complex sequence, pair normalised per window by powers of two -- planes
// output: the channel scales on the samples --, halved samples, fp64 trend sums, register-resident passes through a skewed exchange
// buffer); pass-2 twiddles come from a 16 x 16 table, pass-3 twiddles from the product of two small tables, every table access at a
// constant offset from a base.  Measured: DESIGN.md section 4.1, profiles/r05_stage_a_*.txt.
#include <cstdlib>
#include <type_traits>
#include "sc_common.h"
#include "sc_mtfft_bfly.h"

struct LongArgs {
    const float* x;        // [T][R][C]
    const float* tapers;   // [K][L], already divided by fs
    const float2* tw;      // [N] exp(-2 pi i m / N)
    float2* X;             // [F][W][R][K][C]
    int R, C, L, step, W, K, detrend;
    int n_items;           // (window, trial) groups rounded up to a multiple of 8, times channel tiles
    int g_off, g_end;      // this launch takes the (window, trial) groups g_off ... g_end - 1 (long windows are launched in slices)
    int vec;               // rows can be read in 16-byte pieces (C % 4 == 0, x 16-byte aligned)
    // planes-format output (sc_fused2.hip: two f16 pieces per real, x * scale[c] = h + m, rows [f][w][r][k] of row_bytes; a row is
    // 256-byte tiles of 32 channels: planes Re h, Re m, Im h, Im m of 64 bytes each): when P is set the spectra go there INSTEAD of X
    unsigned char* P;
    const float* scale;    // [C] powers of two
    int64_t row_bytes;
    int dbg;               // SC_MTFFT_DEBUG (results WRONG when set): 1 or 4 = no split / store loop, 2 = no passes,
                           // 16 = non-temporal stores, 32 = no super-tiles, 64 = the other workgroup size at N <= 1024 (A/B)
};

// HT: threads of a half.  512 (one workgroup of 16 waves owns the compute unit) or 256 (two workgroups of 8 waves share it: half the
// channels per workgroup, but the prologue of one runs under the slots of the other).
typedef _Float16 ml_h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned ml_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ml_u32x2 __attribute__((ext_vector_type(2)));
// two scaled reals -> the dwords of their leading and trailing f16 pieces (x = h + m to 22 significant bits): one packed conversion
// and two mixed-precision fmas that read their f16 operand straight from the halves of h (as in sc_mtfft.hip)
__device__ __forceinline__ void ml_split2(float x0, float x1, unsigned& h, unsigned& m) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(x0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(x1), "v"(h));
}
// One step of a 4 x 4 transpose inside a quad of lanes: lanes with `hi` clear give b and take the partner's a into b, lanes with
// `hi` set give a and take the partner's b into a; the partner is lane ^ 1 (CTRL = quad_perm [1, 0, 3, 2]) or lane ^ 2 ([2, 3, 0, 1]).
template <int CTRL>
__device__ __forceinline__ void ml_quad_xchg(ml_u32x4& a, ml_u32x4& b, bool hi) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned give = hi ? a[d] : b[d];
        const unsigned take = (unsigned)__builtin_amdgcn_mov_dpp((int)give, CTRL, 0xf, 0xf, true);
        if (hi) a[d] = take; else b[d] = take;
    }
}

template <int LOG2N, int HT, bool PL, bool GRP = false>
__global__ void __launch_bounds__(2 * HT, HT == 512 ? 1 : 4) mtfft_long_kernel(LongArgs p) {
    constexpr int N = 1 << LOG2N;
    constexpr int THREADS = 2 * HT;
    constexpr int TPF = N / 16;          // threads per transform: 16 points each
    constexpr int NF = HT / TPF;         // transforms (channel pairs) per half
    constexpr int CTH = 2 * NF;          // channels per half
    constexpr int CT = 2 * CTH;          // channels per workgroup
    constexpr int ZS = N + N / 16 + 1;   // skewed exchange buffer per transform (float2), odd stride
    constexpr int WPF = TPF >= 64 ? TPF / 64 : 1;        // waves per transform (N <= 512: 64 / TPF transforms per wave)
    // One wave per transform (N = 1024): LDS executes a wave's instructions in order, so the exchanges between the passes need no
    // workgroup barrier -- ONE barrier per slot (the hand-over between the halves), the waves of a half run free in between, and
    // the taper is double-buffered instead of parked behind a barrier.
    constexpr bool WAVE_LOCAL = TPF <= 64;
    // GROUP_LOCAL (round 6, 2048 samples): the WPF waves of a transform meet at a counter in LDS instead of the workgroup barrier (as in
    // sc_mtfft_mixed.hip): a slot has one workgroup barrier like the wave-local lengths, the taper is double-buffered, and the store
    // slot is not cut into chunks -- the planes output gains most (its four stores per thread leave in one go).  4096 samples keep the
    // barriers: a second taper buffer does not fit beside its exchange buffers.
    constexpr bool GROUP_LOCAL = GRP && !WAVE_LOCAL;
    constexpr bool SLOT1 = WAVE_LOCAL || GROUP_LOCAL;
    constexpr int NB = SLOT1 ? 1 : 4;                                // workgroup barriers of one slot
    static_assert(LOG2N >= 8 && LOG2N <= 12, "256 ... 4096 samples");
    extern __shared__ __align__(16) unsigned char smem[];
    float2* zall = reinterpret_cast<float2*>(smem);             // [2][NF][ZS]
    // Twiddle tables, every access (a per-thread base) + (a compile-time constant):
    //   pass 2:  W_256^(t kk)                                  = T2[t][kk]
    //   pass 3:  W_N^(t ib), ib = 16 hi + lo < 256, t < M      = TH[t][hi] * TL[t][lo],  TH[t][hi] = W_N^(16 t hi), TL[t][lo] = W_N^(t lo)
    // (M = N / 256 is the radix of pass 3 -- no pass 3 at N = 256; at N = 4096 TH is T2)
    constexpr int M = N / 256;
    float2* T2 = zall + 2 * NF * ZS;                             // [16][16]
    float2* TH = LOG2N == 12 ? T2 : T2 + 256;                    // [M][16]
    float2* TL = TH + (LOG2N == 12 ? 256 : M * 16);              // [M][16]
    float* tap = reinterpret_cast<float*>(TL + M * 16);          // [N] the taper in use (zeros from L on); SLOT1: [2][N]
    __shared__ unsigned gcnt[16];                                // GROUP_LOCAL: arrivals at the group barriers, one counter per transform
    __shared__ int nzf[CT], nbf[CT];
    __shared__ unsigned mxc[CT];
    __shared__ double red[THREADS / 64][4];                      // trend sums per wave

    const int tid = threadIdx.x, half = tid / HT, ht = tid & (HT - 1), wv = tid >> 6;
    const int L = p.L, C = p.C, K = p.K;
    // Items.  Workgroup b takes item b = (window, trial, channel tile); block b runs on XCD b % 8, and the tiles of one (window, trial)
    // are dealt to ONE XCD (they share every row they read, and the pieces of a frequency row they write).
    // Channels of a tile: a half stores CTH channels = a 64-byte (N = 2048) or 32-byte (N = 4096) piece of every frequency row;
    // SUP = 16 / CTH tiles form a super-tile of SUP * CT channels in which the h-th halves of the SUP workgroups -- in the same
    // phase, on one XCD, dispatched back to back -- hold 16 ADJACENT channels: their pieces complete a 128-byte line in that XCD's
    // L2 within a store chunk, where the two halves of one workgroup are a phase apart (N = 4096: 2.8 -> 2.1 ms).  Up to 16
    // channels: no super-tiles.
    // Planes output (round 6): a 128-byte line of a tile row is two planes of THIRTY-TWO channels, so the pieces of 32 adjacent channels
    // must meet in the L2 -- super-tiles of 32 / CTH workgroups there (two at 512 / 1024 samples, where the complex64 output needs
    // none: before, a line waited for the other half of its own workgroup, a phase later).
    // (512 / 1024 / 2048 samples: 2.6 / 2.5 / 2.1 -> 1.8 / 1.4 / 1.5 ms at the cfg3 volume, the complex64 output's 1.6 / 1.4 / 1.3; 4096 samples
    //  keep four workgroups per super-tile: eight were slower, 2.58 against 2.40 ms; profiles/r06_stage_a_planes_sup.txt)
    constexpr int LINE_CH = (PL && LOG2N <= 11) ? 32 : 16;
    constexpr int SUP = CTH >= LINE_CH ? 1 : LINE_CH / CTH;
    const bool sup = C > LINE_CH && SUP > 1 && !(p.dbg & 32);
    const int n_ct = sup ? (C + SUP * CT - 1) / (SUP * CT) * SUP : (C + CT - 1) / CT;
    int ch0[2], w, r;                                 // first channel of either half
    {
        const int m = blockIdx.x, xcd = m & 7, j = m >> 3, g = p.g_off + (j / n_ct) * 8 + xcd;
        if (g >= p.g_end) return;
        const int tile = j % n_ct;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            ch0[h] = sup ? (tile / SUP) * (SUP * CT) + h * (SUP * CTH) + (tile % SUP) * CTH : tile * CT + h * CTH;
        w = g / p.R; r = g - w * p.R;
    }
    const int chalf = ch0[half];

    const int pf = ht / TPF, i = ht - pf * TPF;       // transform of this half and butterfly index
    const int lp = 2 * (half * NF + pf);              // this pair's slot in the flag arrays
    float2* zh = zall + half * NF * ZS;
    float2* zf = zh + pf * ZS;
    if (tid < CT) { nzf[tid] = 0; nbf[tid] = 0; mxc[tid] = 0u; }
    if (tid < 16) gcnt[tid] = 0u;
    static_assert(THREADS >= 256 + M * 16, "table fill");
    if (tid < 256) T2[tid] = p.tw[((tid >> 4) * (tid & 15)) * (N / 256)];
    else if (tid < 256 + M * 16) {
        const int e = tid - 256, t = e >> 4, x = e & 15;
        TL[e] = p.tw[t * x];
        if constexpr (LOG2N != 12) TH[e] = p.tw[16 * t * x];
    }
    for (int n = tid; n < N; n += THREADS) tap[n] = (n < L) ? p.tapers[n] : 0.f;
    const bool detr = p.detrend != SC_DETREND_NONE;
    auto pair_scale = [](unsigned mx, bool inverse) -> float {
        const unsigned E = mx >> 23;
        return (E >= 1u && E <= 253u) ? __uint_as_float((inverse ? E : 254u - E) << 23) : 1.f;
    };

    // ---- the window into registers: two half-window tiles [N / 2 rows][CT channels] through the exchange buffers ----
    float2 xs[16];                                    // this thread's pass-1 inputs, all tapers: samples i + t TPF of its pair
    {
        // padded row (floats): the per-thread float2 reads below are conflict-free -- 32 lanes = 32 rows of one pair (N >= 512: 2 RS
        // mod 64 banks apart with RS / 2 odd) or 16 rows of two pairs (N = 256: rows 4 banks apart, the pairs 2)
        constexpr int RS = LOG2N == 8 ? CT + 4 : CT + 2;
        constexpr int QR = CT / 4, V = CTH / 4;       // 16-byte pieces per row, per half
        static_assert((size_t)(N / 2) * RS * 4 <= (size_t)2 * NF * ZS * 8, "tile does not fit the exchange buffers");
        float* tile = reinterpret_cast<float*>(smem);
        const int64_t RC = (int64_t)p.R * C;
        const float* xw = p.x + ((int64_t)w * p.step * p.R + r) * C;
        // (all sixteen-byte pieces of BOTH halves in flight before the first one is parked: one memory latency per workgroup)
        float4 v[2][4];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {          // (N / 2) QR pieces = 4 per thread
                const int idx = tid + it * THREADS, row = idx / QR, q = idx - row * QR, c = ch0[q / V] + 4 * (q % V);
                const int n = hh * (N / 2) + row;
                v[hh][it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < L && c < C) {
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
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (hh == 1) __syncthreads();             // the reads of the first half are done
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = tid + it * THREADS, row = idx / QR, q = idx - row * QR;
                float2* d = reinterpret_cast<float2*>(tile + row * RS + 4 * q);
                d[0] = make_float2(v[hh][it].x, v[hh][it].y);
                d[1] = make_float2(v[hh][it].z, v[hh][it].w);
            }
            __syncthreads();
#pragma unroll
            for (int t = 8 * hh; t < 8 * hh + 8; ++t)
                xs[t] = *reinterpret_cast<const float2*>(tile + (i + (t - 8 * hh) * TPF) * RS + half * CTH + 2 * pf);
        }
    }
    double tsum[4] = {0.0, 0.0, 0.0, 0.0};            // sum x, sum x (l + 1) of channel a; the same of channel b
    if (detr) {
        // trend sums in fp64: a thread's 16 samples, then the lanes of the transform by shuffles (complete up to 1024 samples: a
        // transform is at most one wave); beyond, the waves of a transform in a fixed order below
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const double l1 = (double)(i + t * TPF + 1);
            tsum[0] += (double)xs[t].x; tsum[1] += (double)xs[t].x * l1;
            tsum[2] += (double)xs[t].y; tsum[3] += (double)xs[t].y * l1;
        }
#pragma unroll
        for (int m = (TPF < 64 ? TPF : 64) / 2; m > 0; m >>= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tsum[e] += __shfl_xor(tsum[e], m);
        }
        if constexpr (WPF > 1) {
            if ((tid & 63) == 0) { red[wv][0] = tsum[0]; red[wv][1] = tsum[1]; red[wv][2] = tsum[2]; red[wv][3] = tsum[3]; }
        }
    }
    __syncthreads();                                  // the tile is consumed (the exchange buffers are free), trend sums visible
    if (detr) {
        const double n = (double)L, invL = 1.0 / n;
        const double St = (n + 1.0) * 0.5, Stt = (n + 1.0) * (2.0 * n + 1.0) / (6.0 * n), den = n * Stt - St * St;
        double ab[2][2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            double sum = tsum[2 * ch], sumt = tsum[2 * ch + 1];
            if constexpr (WPF > 1) {
                const int wv0 = (half * HT + pf * TPF) >> 6;
                sum = 0.0; sumt = 0.0;
#pragma unroll
                for (int q = 0; q < WPF; ++q) { sum += red[wv0 + q][2 * ch]; sumt += red[wv0 + q][2 * ch + 1]; }
            }
            sumt /= n;
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
            const double tt = (double)(i + t * TPF + 1) * invL;
            const float dx = (float)((double)xs[t].x - (ab[0][0] * tt + ab[0][1]));
            const float dy = (float)((double)xs[t].y - (ab[1][0] * tt + ab[1][1]));
            const bool in = i + t * TPF < L;          // (zero padding stays zero)
            xs[t].x = in ? dx : 0.f;
            xs[t].y = in ? dy : 0.f;
        }
    }
    {
        // flag 1: the channel is not identically zero; flag 2: it holds a NaN / infinity (such a channel leaves the packed
        // transform -- zeros in its place, its partner stays clean -- and its bins are written as NaN); largest finite magnitude
        // of every channel of this window for the pair normalisation (see sc_mtfft.hip)
        unsigned or0 = 0u, or1 = 0u, mx0 = 0u, mx1 = 0u;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
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
            for (int t = 0; t < 16; ++t) xs[t].x = 0.f;
        }
        if (nbf[lp + 1]) {
#pragma unroll
            for (int t = 0; t < 16; ++t) xs[t].y = 0.f;
        }
        // halved (the 1/2 of the conjugate-symmetry split) and scaled into [1, 2) per channel: exact
        // (planes output: the channel scales -- powers of two -- go onto the samples instead: the two channels of a pair enter their
        //  transform at the same magnitude, and the store loop has no multiply left)
        const int c = chalf + 2 * pf;
        const float h0 = 0.5f * (PL ? (c < C ? p.scale[c] : 1.f) : pair_scale(mxc[lp], false));
        const float h1 = 0.5f * (PL ? (c + 1 < C ? p.scale[c + 1] : 1.f) : pair_scale(mxc[lp + 1], false));
#pragma unroll
        for (int t = 0; t < 16; ++t) { xs[t].x *= h0; xs[t].y *= h1; }
    }
    // the pair this thread stores (its first bin: fb), and what it needs to know about it
    const int spr = ht & (NF - 1), fb = ht / NF, sl = 2 * (half * NF + spr), cs = chalf + 2 * spr;
    unsigned fl;                                      // bit 0 / 1: channel a / b non-finite, 2 / 3: identically zero
    {
        const bool na = nbf[sl] != 0, nb = nbf[sl + 1] != 0;
        fl = (na ? 1u : 0u) | (nb ? 2u : 0u) | ((!na && nzf[sl] == 0) ? 4u : 0u) | ((!nb && nzf[sl + 1] == 0) ? 8u : 0u);
    }
    const float ia = pair_scale(mxc[sl], true), ib = pair_scale(mxc[sl + 1], true);      // back to the samples' units
    const bool vec_ok = (C % 2) == 0;
    const int64_t sF = (int64_t)p.W * p.R * K * C;
    float2* const Xi = p.X + ((int64_t)w * p.R + r) * K * C + cs + (int64_t)fb * sF;     // X[fb][w][r][0][cs]

    // Every LDS address below is (one base per role) + (compile-time constant): with phys(idx) = idx + idx / 16,
    //   pass 1 writes   phys(16 i + u)        = 17 i + u
    //   passes 2, 3 read phys(i + t TPF)      = phys(i) + t (TPF + TPF / 16)
    //   pass 2 writes   phys(j + 16 u)        = phys(j) + 17 u,  j = 16 (i - kk) + kk
    //   pass 3          phys(ib + 256 u)      = phys(ib) + 272 u
    //   the store loop  phys(fb + m FSTEP)    = phys(fb) + m (FSTEP + FSTEP / 16), and for the mirrored bin N - f, f > 0,
    //                   phys(N - fb - m FSTEP) = phys(N - fb) - m (FSTEP + FSTEP / 16)     (FSTEP = 512 / NF is a multiple of 16)
    constexpr int TS = TPF + TPF / 16, FSTEP = HT / NF, FS = FSTEP + FSTEP / 16;
    const int kk = i & 15;
    float2* const zw1 = zf + 17 * i;
    float2* const zr = zf + i + (i >> 4);
    float2* const zw2 = zf + (((i - kk) << 4) + kk) + (i - kk);
    const float2* const zs = zh + spr * ZS + fb + (fb >> 4);
    const float2* const zm = zh + spr * ZS + (N - fb) + ((N - fb) >> 4) - 7 * FS;      // mirrored bin of round m: zm[(7 - m) FS]
    const float2* const zm0 = fb == 0 ? zh + spr * ZS : zm + 7 * FS;                    // round 0: bin N - 0 is bin 0

    // ONE taper buffer serves both halves where a transform spans several waves: taper k is read by half 0 in the first interval of
    // slot 2k and by half 1 in the first interval of slot 2k + 1; behind that interval's barrier half 0 (storing then) replaces it
    // with the next taper.
    constexpr int TPT = N >= HT ? N / HT : 1;         // taper values per thread of a half
    const float2* const t2 = T2 + kk;                                                  // pass 2: t2[16 t]
    const float2* const th = TH + (i >> 4);                                            // pass 3, butterfly b: th[16 t + b TPF / 16]
    const float2* const tl = TL + kk;                                                  //         tl[16 t]

    unsigned epoch = 0u;
    unsigned* const gc = gcnt + half * NF + pf;
    auto group_barrier = [&]() {                      // (a wave's LDS instructions execute in order: the arrival is behind its exchange writes)
        epoch += (unsigned)WPF;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((tid & 63) == 0) __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while ((int)(__hip_atomic_load(gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - epoch) < 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
#define PBAR()                                                      \
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
        float2 a[16], o[16];
        const float* tk = SLOT1 ? tap + (k & 1) * N + i : tap + i;
        if (p.dbg & 2) {
#pragma unroll
            for (int b = 0; b < NB; ++b) __syncthreads();
            return;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float h = tk[t * TPF];
            a[t] = make_float2(xs[t].x * h, xs[t].y * h);
        }
        dft16(a, o);
#pragma unroll
        for (int u = 0; u < 16; ++u) zw1[u] = o[u];
        PBAR();                                                                       // 1
        {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int t = 4 * g; t < 4 * g + 4; ++t) {
                    const float2 v = zr[t * TS];
                    a[t] = (t == 0) ? v : cmul(v, t2[16 * t]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            dft16(a, o);
            PBAR();                                                                   // 2
#pragma unroll
            for (int u = 0; u < 16; ++u) zw2[17 * u] = o[u];
            PBAR();                                                                   // 3
        }
        if constexpr (LOG2N == 12) {        // pass 3: radix 16, P = 256
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int t = 4 * g; t < 4 * g + 4; ++t) {
                    const float2 v = zr[t * TS];
                    a[t] = (t == 0) ? v : cmul(v, cmul(th[16 * t], tl[16 * t]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            dft16(a, o);                    // (in place: every thread writes back exactly the slots it read -- no barrier between)
#pragma unroll
            for (int u = 0; u < 16; ++u) zr[272 * u] = o[u];
        } else if constexpr (LOG2N == 11) { // pass 3: radix 8, P = 256, two butterflies per thread, in place
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float2 q[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float2 v = zr[b * TS + 272 * t];
                    q[t] = (t == 0) ? v : cmul(v, cmul(th[16 * t + b * (TPF / 16)], tl[16 * t]));
                }
                dft8r(q);
#pragma unroll
                for (int u = 0; u < 8; ++u) zr[b * TS + 272 * u] = q[u];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (LOG2N == 10) { // pass 3: radix 4, P = 256, four butterflies per thread, in place
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float2 q[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float2 v = zr[b * TS + 272 * t];
                    q[t] = (t == 0) ? v : cmul(v, cmul(th[16 * t + b * (TPF / 16)], tl[16 * t]));
                }
                dft4r(q[0], q[1], q[2], q[3]);
#pragma unroll
                for (int u = 0; u < 4; ++u) zr[b * TS + 272 * u] = q[u];
            }
        } else if constexpr (LOG2N == 9) {  // pass 3: radix 2, P = 256, eight butterflies per thread, in place
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const float2 u0 = zr[b * TS], u1 = cmul(zr[b * TS + 272], cmul(th[16 + b * (TPF / 16)], tl[16]));
                zr[b * TS] = make_float2(u0.x + u1.x, u0.y + u1.y);
                zr[b * TS + 272] = make_float2(u0.x - u1.x, u0.y - u1.y);
            }
        }                                   // (N = 256: two passes)
        __syncthreads();                                                              // NB
    };

    // split the packed pair at a frequency and store X[f][w][r][k][cs .. cs + 1]
    auto put = [&](float2* dst, float2 u1, float2 u2) {
        float2 A = make_float2((u1.x + u2.x) * ia, (u1.y - u2.y) * ia);       // (Z[f] + conj Z[N-f]) / 2, the half already in the samples
        float2 B = make_float2((u1.y + u2.y) * ib, (u2.x - u1.x) * ib);       // (Z[f] - conj Z[N-f]) / (2 i)
        // (silent / non-finite channels: eight selects per row of a thread; nothing next to the passes of the other half)
        if (fl & 4u) A = make_float2(0.f, 0.f);
        if (fl & 8u) B = make_float2(0.f, 0.f);
        if (fl & 1u) A = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
        if (fl & 2u) B = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
        if (vec_ok) {
            // plain stores: the 32- / 64-byte pieces of a frequency row need the write-back L2 to merge them into lines
            if (p.dbg & 16) sc_stream_store(dst, A, B);
            else *reinterpret_cast<float4*>(dst) = make_float4(A.x, A.y, B.x, B.y);
        } else {
            dst[0] = A;
            if (cs + 1 < C) dst[1] = B;
        }
    };
    // Planes format: a thread takes G = 4 channel pairs (8 channels; 2 pairs at N = 4096) of one frequency, so that every plane leaves
    // as one 16-byte (8-byte) store: 2 G LDS reads, the conjugate-symmetry split, the two-piece f16 split; the four stores of a
    // thread go to the four 64-byte planes of one 256-byte tile row, back to back.  (sc_mtfft.hip transposes 4 planes x 4 frequencies
    // inside a quad of lanes first, so that 16 lanes write one whole tile row per instruction: 0.14 ms at cfg3 THERE, where the
    // store loop waits on the memory side; HERE the storing half shares the VALU with the half that runs the passes, and the 64
    // extra instructions per iteration cost more than the wider pieces return: 1.75 -> 1.66 ms at cfg3 without it, A/B of two
    // libraries on one box, -DML_QUAD_TR.)  Iteration `it` covers the FSP frequencies from it * FSP on; the last one holds the
    // Nyquist bin alone.
    constexpr int G = NF >= 4 ? 4 : NF, NG = NF / G, FSP = HT / NG, PIT = (N / 2) / FSP + 1;
#ifdef ML_QUAD_TR             // (A/B: the quad transposition of sc_mtfft.hip's planes store loop)
    constexpr bool TR = NG >= 4;
#else
    constexpr bool TR = false;
#endif
    const int pj = ht & 3, pgrp = TR ? (ht >> 2) % NG : ht % NG, pfq = TR ? (ht / (4 * NG)) * 4 + pj : ht / NG, pcg = chalf + 2 * G * pgrp;
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
        using VT = std::conditional_t<G == 4, ml_u32x4, ml_u32x2>;
        const int f = pfq + it * FSP;
        if (!(pcg < ((C + 31) & ~31)) || (TR ? f - pj : f) > N / 2) return;      // (absent channels of a started tile are written: zeros)
        const int64_t rows_f = (int64_t)p.W * p.R * K;
        unsigned char* dst0 = p.P + (((int64_t)w * p.R + r) * K + k) * p.row_bytes + (pcg >> 5) * 256 + (pcg & 31) * 2 + (TR ? pj * 64 : 0);
        const int fr = f <= N / 2 ? f : N / 2;        // (TR: a lane past the Nyquist bin reads it again; its column is never stored)
        float2 u1q[G], u2q[G];                        // all LDS reads in flight before the first split
        const float2* zg = zh + (G * pgrp) * ZS;
        const int i1 = fr + (fr >> 4), n2 = (N - fr) & (N - 1), i2 = n2 + (n2 >> 4);
#pragma unroll
        for (int q = 0; q < G; ++q) { u1q[q] = zg[q * ZS + i1]; u2q[q] = zg[q * ZS + i2]; }
        VT rh, rm, ih, im;
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const float2 u1 = u1q[q], u2 = u2q[q];
            float2 A = make_float2(u1.x + u2.x, u1.y - u2.y);       // (Z[f] + conj Z[N-f]) / 2, the half already in the samples
            float2 B = make_float2(u1.y + u2.y, u2.x - u1.x);       // (Z[f] - conj Z[N-f]) / (2 i)
            if (pflag) {
                const int e = 2 * (half * NF + G * pgrp + q);
                const bool qna = nbf[e] != 0, qnb = nbf[e + 1] != 0;
                if (!qna && nzf[e] == 0) A = make_float2(0.f, 0.f);
                if (!qnb && nzf[e + 1] == 0) B = make_float2(0.f, 0.f);
                if (qna) A = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
                if (qnb) B = make_float2(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000));
            }
            unsigned h, m;
            ml_split2(A.x, B.x, h, m);
            rh[q] = h; rm[q] = m;
            ml_split2(A.y, B.y, h, m);
            ih[q] = h; im[q] = m;
        }
        const int64_t fs = rows_f * p.row_bytes;
        if constexpr (TR) {
            // items (rh, rm, ih, im) = planes 0 .. 3 of this lane's frequency  ->  plane pj of the quad's frequencies 0 .. 3
            ml_quad_xchg<0xB1>(rh, rm, (pj & 1) != 0);
            ml_quad_xchg<0xB1>(ih, im, (pj & 1) != 0);
            ml_quad_xchg<0x4E>(rh, ih, (pj & 2) != 0);
            ml_quad_xchg<0x4E>(rm, im, (pj & 2) != 0);
            const int f0 = f - pj;
            unsigned char* dst = dst0 + (int64_t)f0 * fs;
            *reinterpret_cast<VT*>(dst) = rh;
            if (f0 + 1 <= N / 2) *reinterpret_cast<VT*>(dst + fs) = rm;
            if (f0 + 2 <= N / 2) *reinterpret_cast<VT*>(dst + 2 * fs) = ih;
            if (f0 + 3 <= N / 2) *reinterpret_cast<VT*>(dst + 3 * fs) = im;
        } else {
            unsigned char* dst = dst0 + (int64_t)f * fs;
            *reinterpret_cast<VT*>(dst) = rh;
            *reinterpret_cast<VT*>(dst + 64) = rm;
            *reinterpret_cast<VT*>(dst + 128) = ih;
            *reinterpret_cast<VT*>(dst + 192) = im;
        }
    };
    // One store slot: taper k leaves (F * NF = 8 * 512 + NF outputs per half, in four chunks); half 0 parks taper k + 1.  NB workgroup barriers.
    auto store = [&](int k) {
        float2* Xk = Xi + (int64_t)k * C;
        const int64_t sR = (int64_t)FSTEP * sF;       // one round further
        const bool live = cs < C && !(p.dbg & 5);
        const bool park = half == 0 && k + 1 < K;
        float hn[TPT];
        if (park) {
            const float* tp = p.tapers + ((int64_t)(k + 1) * L + ht);      // (one address per slot, constants from there)
#pragma unroll
            for (int j = 0; j < TPT; ++j) hn[j] = (ht + j * HT < L) ? tp[j * HT] : 0.f;
        }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if constexpr (PL) {
                if (!(p.dbg & 5)) {
#pragma unroll
                    for (int it = ch; it < (ch == 3 ? PIT : ch + 1); ++it) put_planes(k, it);
                }
            } else if (live) {
                float2 z1[2], z2[2];
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int m = 2 * ch + it;
                    z1[it] = zs[m * FS];
                    z2[it] = m == 0 ? zm0[0] : zm[(7 - m) * FS];
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) put(Xk + (2 * ch + it) * sR, z1[it], z2[it]);
                if (ch == 3 && ht < NF) {                                     // the Nyquist bin: fb = 0 here
                    const float2 zn = zs[8 * FS];
                    put(Xk + 8 * sR, zn, zn);
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
}

static bool long_enabled() {
    const char* e = sc_switch(SC_SW_MTFFT_LONG);
    return !e || atoi(e) != 0;
}

// N = 256 ... 4096 (SC_MTFFT_LONG=0: never; =512 / 1024 / 2048: from that many samples on -- A/B; =1: whatever the size), and enough
// (window, trial, channel tile) items to give every compute unit a workgroup: a workgroup here holds 4 x the channels of the
// round-3 kernels', so a small problem (BASELINE configs[1]: 100 trials x 32 channels = 100 items at N = 1024) fills the chip
// better with those.
bool sc_internal_mtfft_long_applies(int64_t N, int64_t C, int64_t groups) {
    if (!long_enabled()) return false;
    const char* e = sc_switch(SC_SW_MTFFT_LONG);
    const bool force = e && atoi(e) == 1;                            // SC_MTFFT_LONG=1: every length it has, whatever the size (tests)
    const int lo = (e && atoi(e) >= 256) ? atoi(e) : 256;
    if (!(N >= lo && N <= 4096 && (N & (N - 1)) == 0 && C >= 1)) return false;
    if (force) return true;
    const int64_t ct = 4 * ((N <= 512 ? 256 : 512) / (N / 16));
    return groups * ((C + ct - 1) / ct) >= 256;
}

// the channels the workgroups of a launch write (planes output: where this leaves part of the last 32-channel tile unwritten, the
// caller clears the buffer first)
int64_t sc_internal_mtfft_long_coverage(int64_t N, int64_t C) {
    // (called for the planes output only: its super-tiles span 32 channels)
    const int64_t ht = N <= 512 ? 256 : 512, nf = ht / (N / 16), ct = 4 * nf, line = N <= 2048 ? 32 : 16, sup = 2 * nf >= line ? 1 : line / (2 * nf);
    const char* d = sc_switch(SC_SW_MTFFT_DEBUG);
    const int dbg = d ? atoi(d) : 0;
    if (dbg & 64) return 0;                                           // (A/B of the other workgroup size: always clear)
    return (C > line && sup > 1 && !(dbg & 32)) ? (C + sup * ct - 1) / (sup * ct) * sup * ct : (C + ct - 1) / ct * ct;
}

template <int LOG2N, int HT, bool PL, bool GRP = false>
static int launch_long_(LongArgs a, hipStream_t st) {
    constexpr int N = 1 << LOG2N, TPF = N / 16, NF = HT / TPF, CT = 4 * NF, ZS = N + N / 16 + 1;
    constexpr size_t lds = (size_t)2 * NF * ZS * 8 + (256 + (LOG2N == 12 ? 256 : 2 * (N / 256) * 16)) * 8 + (size_t)N * 4 * ((TPF <= 64 || GRP) ? 2 : 1);
    static_assert((lds + 512) * (HT == 512 ? 1 : 2) <= 160 * 1024, "LDS budget exceeded");      // (+ the static flags and trend sums)
    auto k = mtfft_long_kernel<LOG2N, HT, PL, GRP>;
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    constexpr int LINE_CH = (PL && LOG2N <= 11) ? 32 : 16, SUP = 2 * NF >= LINE_CH ? 1 : LINE_CH / (2 * NF);      // = the kernel's
    const int64_t n_ct = (a.C > LINE_CH && SUP > 1 && !(a.dbg & 32)) ? (a.C + SUP * CT - 1) / (SUP * CT) * SUP : (a.C + CT - 1) / CT;
    // Slices (SC_MTFFT_SLICE=<dispatch rounds per launch>, default 0 = one launch; an A/B switch).  Round 5 saw N = 4096 fall from 2.7-2.8
    // TB/s at 3.7 GB of spectra to 1.96 at 11 GB, whatever the number of windows.  Round 6 (profiles/r06_stage_a_volume.txt): it is the
    // STORE stream and the FOOTPRINT -- with the passes off the stores of 3.7 GB take 0.72 ms (5.1 TB/s), those of 11 GB 4.05 ms (2.7
    // TB/s), while the passes and the prologue scale linearly (x 2.9 for x 3 the volume) -- and not the drift of the super-tile partners
    // over a long launch, which was the suspicion this switch was written to test: launches of 6, 12 or 24 dispatch rounds change
    // nothing (1.96-1.98 TB/s).  Every wave instruction of a store slot writes 32-byte pieces of 64 different frequency rows, megabytes
    // apart: 2049 pages per slot and half, which stops fitting the address-translation reach somewhere between 4 and 11 GB of output.
    const int64_t groups = (int64_t)a.W * a.R;
    int64_t slice_groups = groups;
    {
        const char* e = sc_switch(SC_SW_MTFFT_SLICE);
        const int64_t rounds = e ? atoll(e) : 0;
        if (rounds > 0) slice_groups = (rounds * 256 / n_ct + 7) / 8 * 8;
        if (slice_groups < 8) slice_groups = 8;
    }
    for (int64_t g0 = 0; g0 < groups; g0 += slice_groups) {
        const int64_t g1 = g0 + slice_groups < groups ? g0 + slice_groups : groups;
        const int64_t groups8 = (g1 - g0 + 7) / 8 * 8;
        if (groups8 * n_ct >= ((int64_t)1 << 31)) {
            sc_set_error("multitaper FFT (N=%d): too many windows x trials for one launch", N);
            return SC_EINVAL;
        }
        a.n_items = (int)(groups8 * n_ct);
        a.g_off = (int)g0; a.g_end = (int)g1;
        hipLaunchKernelGGL(k, dim3((unsigned)a.n_items), dim3(2 * HT), lds, st, a);
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

template <int LOG2N, int HT, bool GRP = false>
static int launch_long(const LongArgs& a, hipStream_t st) {
    return a.P ? launch_long_<LOG2N, HT, true, GRP>(a, st) : launch_long_<LOG2N, HT, false, GRP>(a, st);
}

int sc_internal_mtfft_long(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L, int64_t step, int64_t W, int64_t N,
                           const float* d_tapers, int64_t K, int detrend_type, const void* d_twiddles, void* d_X, void* d_P,
                           const float* d_scale, hipStream_t st) {
    LongArgs a{};
    a.x = d_x; a.tapers = d_tapers; a.tw = (const float2*)d_twiddles; a.X = (float2*)d_X;
    a.P = (unsigned char*)d_P; a.scale = d_scale; a.row_bytes = 256 * ((C + 31) / 32);
    SC_REQUIRE(!d_P || (d_scale && C % 2 == 0), "planes output needs channel scales and an even number of signals");
    a.R = (int)R; a.C = (int)C; a.L = (int)L; a.step = (int)step; a.W = (int)W; a.K = (int)K; a.detrend = detrend_type;
    a.vec = (C % 4 == 0 && ((uintptr_t)d_x & 15) == 0) ? 1 : 0;
    { const char* d = sc_switch(SC_SW_MTFFT_DEBUG); a.dbg = d ? atoi(d) : 0; }
    SC_REQUIRE((W - 1) * step + L <= T, "windows exceed the time series");
    switch (N) {
    case 256: return (a.dbg & 64) ? launch_long<8, 512>(a, st) : launch_long<8, 256>(a, st);
    case 512: return (a.dbg & 64) ? launch_long<9, 512>(a, st) : launch_long<9, 256>(a, st);
    case 1024: return (a.dbg & 64) ? launch_long<10, 256>(a, st) : launch_long<10, 512>(a, st);
    case 2048: return (a.dbg & 256) ? launch_long<11, 512>(a, st) : launch_long<11, 512, true>(a, st);      // (256: the barrier form, A/B)
    case 4096: return launch_long<12, 512>(a, st);
    }
    return SC_EUNSUPPORTED;
}
