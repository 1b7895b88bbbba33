// sc_mtfft_long.hip -- stage A for long power-of-two windows (N = 1024, 2048, 4096): window extraction + detrend + DPSS taper
// multiply + real FFT + transposed store of the one-sided spectra X[f][w][r][k][c], like sc_mtfft.hip
// (reference: transforms.py:1147-1171 sliding windows, :1311-1405 _multitaper_fft, :1798-1915 detrend).
//
// Why a second kernel.  At these lengths one packed transform (two channels) fills 35 KB of LDS, so a workgroup holds few
// channels, and the round-3 kernel (mtfft16_kernel<11 / 12>) ran its three phases one after the other on every compute unit
// (SC_MTFFT_DEBUG ablation at the cfg3 volume, N = 4096: 2.38 ms = 0.87 load + detrend, 0.75 passes, 0.66 stores):
//  * its loads took 8 bytes of 64 different rows per wave instruction (x is [time][trial][channel]: a workgroup needs 16 bytes
//    of every 512-byte row) -- 0.6 TB/s of samples;
//  * the two workgroups of a compute unit fell into step: both in the passes (sharing the VALU), then both in the store loop
//    (sharing the memory pipe), so nothing overlapped.
// Here:
//  1. a tiled transpose turns the series into xt[trial][channel][time] first (0.5 GB read + written once at the cfg3 volume;
//     stream-ordered scratch, <= 1 GiB at a time), so a wave reads 256 contiguous bytes of ONE channel per instruction;
//  2. a workgroup is TWO halves of 512 threads that run in ANTI-PHASE by construction: while half 0 runs the radix-16 passes
//     of taper k (VALU + LDS), half 1 splits and stores its taper k - 1 (memory pipe), then they swap -- the phase boundary is
//     a workgroup barrier both halves reach, and the passes' inner barriers are matched by barriers between the store chunks
//     of the other half.  One workgroup (16 waves, 139 KB of LDS: one exchange buffer per half) owns the compute unit.
// Each half transforms NF = 512 / (N / 16) channel pairs (8 / 4 / 2 at N = 1024 / 2048 / 4096), a workgroup 4 NF channels.
// Arithmetic as in sc_mtfft.hip (two real channels per complex sequence, pair normalised per window by powers of two, halved
// samples, fp64 trend sums, three register-resident passes through a skewed exchange buffer); pass-2 twiddles come from a
// 16 x 16 table, pass-3 twiddles from the product of two small tables, every table access at a constant offset from a base.
#include <cstdlib>
#include <type_traits>
#include "sc_common.h"
#include "sc_mtfft_bfly.h"

struct LongArgs {
    const float* xt;       // [Rc][C][Tt]: row (r - r_off, c) holds that channel's samples, time fastest
    const float* tapers;   // [K][L], already divided by fs
    const float2* tw;      // [N] exp(-2 pi i m / N)
    float2* X;             // [F][W][R][K][C]
    int64_t Tt;
    int R, C, L, step, W, K, detrend;
    int r_off, Rc;         // trials [r_off, r_off + Rc) of this launch
    int n_items;           // (window, trial) groups rounded up to a multiple of 8, times channel tiles
    int dbg;               // SC_MTFFT_DEBUG (results WRONG when set): 1 or 4 = no split / store loop, 2 = no passes,
                           // 16 = non-temporal stores, 32 = no super-tiles (A/B)
};

// x[t][col] (col = trial * C + channel, `ld` columns) -> xt[col - col0][t], t < T_used, 64 x 64 tiles through LDS
__global__ void __launch_bounds__(256) series_transpose_kernel(const float* __restrict__ x, float* __restrict__ xt, int64_t T_used,
                                                               int64_t ld, int64_t col0, int64_t ncols, int64_t Tt) {
    __shared__ float tile[64][65];
    const int64_t t0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t t = t0 + ly + 4 * j, c = c0 + lx;
        tile[ly + 4 * j][lx] = (t < T_used && c < ncols) ? x[t * ld + col0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t c = c0 + ly + 4 * j, t = t0 + lx;
        if (c < ncols && t < Tt) xt[c * Tt + t] = tile[lx][ly + 4 * j];
    }
}

template <int LOG2N>
__global__ void __launch_bounds__(1024, 1) mtfft_long_kernel(LongArgs p) {
    constexpr int N = 1 << LOG2N;
    constexpr int HT = 512;              // threads of a half
    constexpr int TPF = N / 16;          // threads per transform: 16 points each
    constexpr int NF = HT / TPF;         // transforms (channel pairs) per half
    constexpr int CTH = 2 * NF;          // channels per half
    constexpr int CT = 2 * CTH;          // channels per workgroup
    constexpr int ZS = N + N / 16 + 1;   // skewed exchange buffer per transform (float2), odd stride
    constexpr int NB = LOG2N == 12 ? 5 : 4;      // workgroup barriers of one slot
    constexpr int WPF = TPF / 64;        // waves per transform
    extern __shared__ __align__(16) unsigned char smem[];
    float2* zall = reinterpret_cast<float2*>(smem);             // [2][NF][ZS]
    // Twiddle tables, every access (a per-thread base) + (a compile-time constant):
    //   pass 2:  W_256^(t kk)                                  = T2[t][kk]
    //   pass 3:  W_N^(t ib), ib = 16 hi + lo < 256, t < M      = TH[t][hi] * TL[t][lo],  TH[t][hi] = W_N^(16 t hi), TL[t][lo] = W_N^(t lo)
    // (M = N / 256 is the radix of pass 3; at N = 4096 TH is T2)
    constexpr int M = N / 256;
    float2* T2 = zall + 2 * NF * ZS;                             // [16][16]
    float2* TH = LOG2N == 12 ? T2 : T2 + 256;                    // [M][16]
    float2* TL = TH + (LOG2N == 12 ? 256 : M * 16);              // [M][16]
    float* tap = reinterpret_cast<float*>(TL + M * 16);          // [N] the taper in use (zeros from L on)
    __shared__ int nzf[CT], nbf[CT];
    __shared__ unsigned mxc[CT];
    __shared__ double red[16][4];                                // trend sums per wave

    const int tid = threadIdx.x, half = tid >> 9, ht = tid & (HT - 1), wv = tid >> 6;
    const int L = p.L, C = p.C, K = p.K;
    // Items.  An item is (window, trial, channel tile); the workgroup walks items blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x
    // is a multiple of 8, so a workgroup stays on its XCD b % 8 and all tiles of a (window, trial) stay on ONE XCD).
    // Channels of a tile: a half stores CTH channels = a 64-byte (N = 2048) or 32-byte (N = 4096) piece of every frequency row;
    // SUP = 16 / CTH tiles form a super-tile of SUP * CT channels in which the h-th halves of the SUP workgroups -- in the same
    // phase, on one XCD -- hold 16 ADJACENT channels: their pieces complete a 128-byte line in that XCD's L2 within a store chunk,
    // where the two halves of one workgroup are a phase apart (N = 4096: 2.8 -> 2.1 ms).  Up to 16 channels: no super-tiles.
    constexpr int SUP = CTH >= 16 ? 1 : 16 / CTH;
    const bool sup = C > 16 && SUP > 1 && !(p.dbg & 32);
    const int n_ct = sup ? (C + SUP * CT - 1) / (SUP * CT) * SUP : (C + CT - 1) / CT;
    const int n_groups = p.W * p.Rc;
    auto item = [&](int m, int& chalf, int& w, int& r) -> bool {
        const int xcd = m & 7, j = m >> 3, g = (j / n_ct) * 8 + xcd;
        if (m >= p.n_items || g >= n_groups) return false;
        const int tile = j % n_ct;
        chalf = sup ? (tile / SUP) * (SUP * CT) + half * (SUP * CTH) + (tile % SUP) * CTH : tile * CT + half * CTH;
        w = g / p.Rc; r = p.r_off + (g - w * p.Rc);
        return true;
    };
    int n_mine = 0;                                   // (the valid items of a workgroup are a prefix of its walk)
    {
        int c_, w_, r_;
        for (int m = blockIdx.x; item(m, c_, w_, r_); m += gridDim.x) ++n_mine;
    }
    if (n_mine == 0) return;

    const int pf = ht / TPF, i = ht - pf * TPF;       // transform of this half and butterfly index
    const int lp = 2 * (half * NF + pf);              // this pair's slot in the flag arrays
    float2* zh = zall + half * NF * ZS;
    float2* zf = zh + pf * ZS;
    if (tid < 256) T2[tid] = p.tw[((tid >> 4) * (tid & 15)) * (N / 256)];
    else if (tid < 256 + M * 16) {
        const int e = tid - 256, t = e >> 4, x = e & 15;
        TL[e] = p.tw[t * x];
        if constexpr (LOG2N != 12) TH[e] = p.tw[16 * t * x];
    }
    for (int n = tid; n < N; n += 1024) tap[n] = (n < L) ? p.tapers[n] : 0.f;
    const bool detr = p.detrend != SC_DETREND_NONE;
    auto pair_scale = [](unsigned mx, bool inverse) -> float {
        const unsigned E = mx >> 23;
        return (E >= 1u && E <= 253u) ? __uint_as_float((inverse ? E : 254u - E) << 23) : 1.f;
    };

    // ---- what a half holds of its current item ----
    float2 xs[16];                                    // this thread's pass-1 inputs, all tapers
    const int spr = ht & (NF - 1), fb = ht / NF, sl = 2 * (half * NF + spr);      // the pair this thread stores, its first bin
    float2* Xi = nullptr;                             // X[fb][w][r][0][cs]
    int cs = 0;                                       // first channel of the stored pair
    unsigned fl = 0;                                  // bit 0 / 1: channel a / b non-finite, 2 / 3: identically zero
    float ia = 1.f, ib = 1.f;                         // back to the samples' units
    const bool vec_ok = (C % 2) == 0;
    const int64_t sF = (int64_t)p.W * p.R * K * C;

    // The prologue of an item in four steps with a workgroup barrier between consecutive ones.  For the first item both halves take
    // them before the slot loop; from then on a half takes them in its LAST store slot of the previous item (whose passes are
    // done: xs is free), one step per store chunk -- the samples travel while the previous spectra leave.
    auto pro_load = [&](int chalf, int w, int r) {    // step 1: samples in flight, flags of this half cleared
        const int cpair = chalf + 2 * pf;
        const float* row0 = p.xt + ((int64_t)(r - p.r_off) * C + cpair) * p.Tt + (int64_t)w * p.step + i;
        const bool h0 = cpair < C, h1 = cpair + 1 < C;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float2 v = make_float2(0.f, 0.f);
            if (i + t * TPF < L) {
                if (h0) v.x = row0[t * TPF];
                if (h1) v.y = row0[p.Tt + t * TPF];
            }
            xs[t] = v;
        }
        if (ht < CTH) { nzf[half * CTH + ht] = 0; nbf[half * CTH + ht] = 0; mxc[half * CTH + ht] = 0u; }
    };
    auto pro_sums = [&]() {                           // step 2: trend sums in fp64 -- a thread's 16 samples, its wave by shuffles
        if (!detr) return;
        double s0 = 0.0, t0 = 0.0, s1 = 0.0, t1 = 0.0;
        int iv = i + 1;
        asm volatile("" : "+v"(iv));                  // (keeps the sixteen sample positions out of the slot loop's invariants: registers)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const double l1 = (double)(iv + t * TPF);
            s0 += (double)xs[t].x; t0 += (double)xs[t].x * l1;
            s1 += (double)xs[t].y; t1 += (double)xs[t].y * l1;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            s0 += __shfl_xor(s0, m); t0 += __shfl_xor(t0, m);
            s1 += __shfl_xor(s1, m); t1 += __shfl_xor(t1, m);
        }
        if ((tid & 63) == 0) { red[wv][0] = s0; red[wv][1] = t0; red[wv][2] = s1; red[wv][3] = t1; }
    };
    auto pro_detrend = [&]() {                        // step 3: the waves of a transform in a fixed order, detrend, channel flags
        if (detr) {
            const int wv0 = (half * HT + pf * TPF) >> 6;
            const double n = (double)L, invL = 1.0 / n;
            const double St = (n + 1.0) * 0.5, Stt = (n + 1.0) * (2.0 * n + 1.0) / (6.0 * n), den = n * Stt - St * St;
            double ab[2][2];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                double sum = 0.0, sumt = 0.0;
#pragma unroll
                for (int q = 0; q < WPF; ++q) { sum += red[wv0 + q][2 * ch]; sumt += red[wv0 + q][2 * ch + 1]; }
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
            int iv = i + 1;
            asm volatile("" : "+v"(iv));              // (as above)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double tt = (double)(iv + t * TPF) * invL;
                const float dx = (float)((double)xs[t].x - (ab[0][0] * tt + ab[0][1]));
                const float dy = (float)((double)xs[t].y - (ab[1][0] * tt + ab[1][1]));
                const bool in = i + t * TPF < L;      // (zero padding stays zero)
                xs[t].x = in ? dx : 0.f;
                xs[t].y = in ? dy : 0.f;
            }
        }
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
    };
    auto pro_finish = [&](int chalf, int w, int r) {  // step 4: samples ready for the passes, the store state of the item
        if (nbf[lp]) {
#pragma unroll
            for (int t = 0; t < 16; ++t) xs[t].x = 0.f;
        }
        if (nbf[lp + 1]) {
#pragma unroll
            for (int t = 0; t < 16; ++t) xs[t].y = 0.f;
        }
        // halved (the 1/2 of the conjugate-symmetry split) and scaled into [1, 2) per channel: exact
        const float h0 = 0.5f * pair_scale(mxc[lp], false), h1 = 0.5f * pair_scale(mxc[lp + 1], false);
#pragma unroll
        for (int t = 0; t < 16; ++t) { xs[t].x *= h0; xs[t].y *= h1; }
        const bool na = nbf[sl] != 0, nb = nbf[sl + 1] != 0;
        fl = (na ? 1u : 0u) | (nb ? 2u : 0u) | ((!na && nzf[sl] == 0) ? 4u : 0u) | ((!nb && nzf[sl + 1] == 0) ? 8u : 0u);
        ia = pair_scale(mxc[sl], true); ib = pair_scale(mxc[sl + 1], true);
        cs = chalf + 2 * spr;
        Xi = p.X + ((int64_t)w * p.R + r) * K * C + cs + (int64_t)fb * sF;
    };

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

    // ONE taper buffer serves both halves: taper k is read by half 0 in the first interval of slot 2k and by half 1 in the first
    // interval of slot 2k + 1; behind that interval's barrier half 0 (storing then) replaces it with the next taper.
    constexpr int TPT = N / HT;                       // taper values per thread of a half
    const float2* const t2 = T2 + kk;                                                  // pass 2: t2[16 t]
    const float2* const th = TH + (i >> 4);                                            // pass 3, butterfly b: th[16 t + b TPF / 16]
    const float2* const tl = TL + kk;                                                  //         tl[16 t]

    auto passes = [&]() {                             // NB workgroup barriers
        float2 a[16], o[16];
        if (p.dbg & 2) {
#pragma unroll
            for (int b = 0; b < NB; ++b) __syncthreads();
            return;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float h = tap[i + t * TPF];
            a[t] = make_float2(xs[t].x * h, xs[t].y * h);
        }
        dft16(a, o);
#pragma unroll
        for (int u = 0; u < 16; ++u) zw1[u] = o[u];
        __syncthreads();                                                              // 1
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
            __syncthreads();                                                          // 2
#pragma unroll
            for (int u = 0; u < 16; ++u) zw2[17 * u] = o[u];
            __syncthreads();                                                          // 3
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
            dft16(a, o);
            __syncthreads();                                                          // 4
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
        } else {                            // pass 3: radix 4, P = 256, four butterflies per thread, in place
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
        }
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
    // One store slot: taper k of the current item leaves (F * NF = 8 * 512 + NF outputs per half, in four chunks); half 0 parks taper
    // k_next; with m_next >= 0 the half also takes the prologue of its next item.  NB workgroup barriers.
    auto store = [&](int k, int k_next, int m_next) {
        float2* Xk = Xi + (int64_t)k * C;
        const int64_t sR = (int64_t)FSTEP * sF;       // one round further
        const bool live = cs < C && !(p.dbg & 5);
        const bool park = half == 0 && k_next >= 0;
        int nc = 0, nw = 0, nr = 0;
        const bool pro = m_next >= 0 && item(m_next, nc, nw, nr);
        float hn[TPT];
        if (park) {
            const float* tp = p.tapers + ((int64_t)k_next * L + ht);       // (one address per slot, constants from there)
#pragma unroll
            for (int j = 0; j < TPT; ++j) hn[j] = (ht + j * HT < L) ? tp[j * HT] : 0.f;
        }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (live) {
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
            if (pro) {
                if (ch == 0) pro_load(nc, nw, nr);
                else if (ch == 1) pro_sums();
                else if (ch == 2) pro_detrend();
            }
            __syncthreads();
            if (ch == 0 && park) {                    // the other half has read the taper in use (its first interval): replace it
#pragma unroll
                for (int j = 0; j < TPT; ++j) tap[ht + j * HT] = hn[j];
            }
        }
        if constexpr (NB == 5) __syncthreads();
        if (pro) pro_finish(nc, nw, nr);              // (behind the last store of the current item: its store state is dead)
    };

    // ---- the first item: both halves together ----
    {
        int c_ = 0, w_ = 0, r_ = 0;
        item(blockIdx.x, c_, w_, r_);
        pro_load(c_, w_, r_);
        __syncthreads();                              // flags cleared, tables visible
        pro_sums();
        __syncthreads();
        pro_detrend();
        __syncthreads();
        pro_finish(c_, w_, r_);
    }
    // Slot q of a half: item q / 2K, taper (q % 2K) / 2, passes in the even slots and the store in the odd ones; half 1 is one slot behind half 0.
    const int n_slots = 2 * K * n_mine;
#pragma nounroll
    for (int gs = 0; gs <= n_slots; ++gs) {
        const int q = gs - half;
        if (q < 0 || q >= n_slots) {
#pragma unroll
            for (int b = 0; b < NB; ++b) __syncthreads();
            continue;
        }
        const int it = q / (2 * K), ph = q - it * 2 * K, k = ph >> 1;
        if (!(ph & 1)) {
            passes();
        } else {
            const bool more = it + 1 < n_mine;
            store(k, k + 1 < K ? k + 1 : (more ? 0 : -1), (k + 1 == K && more) ? (int)blockIdx.x + (it + 1) * (int)gridDim.x : -1);
        }
    }
}

static bool long_enabled() {
    const char* e = sc_switch(SC_SW_MTFFT_LONG);
    return !e || atoi(e) != 0;
}

bool sc_internal_mtfft_long_applies(int64_t N, int64_t C) {
    if (!long_enabled()) return false;
    const char* e = sc_switch(SC_SW_MTFFT_LONG);
    const int lo = e && atoi(e) >= 1024 ? atoi(e) : 2048;          // SC_MTFFT_LONG=1024: also the 1024-sample windows (A/B)
    return N >= lo && N <= 4096 && (N & (N - 1)) == 0 && C >= 1;
}

template <int LOG2N>
static int launch_long(LongArgs a, const float* d_x, int64_t T, hipStream_t st) {
    constexpr int N = 1 << LOG2N, TPF = N / 16, NF = 512 / TPF, CT = 4 * NF, ZS = N + N / 16 + 1;
    constexpr size_t lds = (size_t)2 * NF * ZS * 8 + (256 + (LOG2N == 12 ? 256 : 2 * (N / 256) * 16)) * 8 + (size_t)N * 4;
    static_assert(lds + 1024 <= 160 * 1024, "LDS budget exceeded");
    auto k = mtfft_long_kernel<LOG2N>;
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // the samples the windows cover, transposed for a range of trials at a time (scratch <= 1 GiB)
    const int64_t T_used = (int64_t)(a.W - 1) * a.step + a.L, Tt = (T_used + 63) / 64 * 64;
    SC_REQUIRE(T_used <= T, "windows exceed the time series");
    const char* mb = sc_switch(SC_SW_MTFFT_LONG_SCRATCH_MB);       // (diagnostic: a small scratch exercises the trial ranges)
    int64_t rc = (mb && atoi(mb) > 0 ? (int64_t)atoi(mb) << 20 : (int64_t)1 << 30) / (Tt * a.C * 4);
    rc = rc < 1 ? 1 : (rc > a.R ? a.R : rc);
    float* xt = nullptr;
    if (sc_internal_pool_alloc((void**)&xt, (size_t)(rc * a.C * Tt) * sizeof(float), st) != hipSuccess) {
        (void)hipGetLastError();
        sc_set_error("multitaper FFT (N=%d): scratch allocation failed", N);
        return SC_ENOMEM;
    }
    a.xt = xt; a.Tt = Tt;
    int rc_ret = SC_OK, dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    n_cu = n_cu < 8 ? 8 : n_cu / 8 * 8;
    for (int64_t r0 = 0; r0 < a.R; r0 += rc) {
        const int64_t n = a.R - r0 < rc ? a.R - r0 : rc, ncols = n * a.C;
        hipLaunchKernelGGL(series_transpose_kernel, dim3((unsigned)(Tt / 64), (unsigned)((ncols + 63) / 64)), dim3(256), 0, st,
                           d_x, xt, T_used, (int64_t)a.R * a.C, r0 * a.C, ncols, Tt);
        a.r_off = (int)r0; a.Rc = (int)n;
        constexpr int SUP = 2 * NF >= 16 ? 1 : 16 / (2 * NF);      // = the kernel's
        const int64_t n_ct = (a.C > 16 && SUP > 1 && !(a.dbg & 32)) ? (a.C + SUP * CT - 1) / (SUP * CT) * SUP : (a.C + CT - 1) / CT;
        const int64_t groups8 = ((int64_t)a.W * n + 7) / 8 * 8;
        if (groups8 * n_ct >= ((int64_t)1 << 31)) { sc_set_error("multitaper FFT (N=%d): too many windows x trials for one launch", N); rc_ret = SC_EINVAL; break; }
        a.n_items = (int)(groups8 * n_ct);
        // persistent workgroups, one per compute unit (a multiple of 8: a workgroup's items stay on its XCD)
        // items per workgroup (A/B through SC_MTFFT_DEBUG: 64 -> 1, 128 -> 2, 256 -> 4, 512 -> one workgroup per compute unit)
        const int64_t ipw = (a.dbg & 64) ? 1 : (a.dbg & 128) ? 2 : (a.dbg & 256) ? 4 : (a.dbg & 512) ? (a.n_items + n_cu - 1) / n_cu : 1;
        int64_t grid = ((a.n_items + ipw - 1) / ipw + 7) / 8 * 8;
        if (grid > a.n_items) grid = a.n_items;
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(1024), lds, st, a);
    }
    (void)hipFreeAsync(xt, st);
    if (rc_ret != SC_OK) return rc_ret;
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

int sc_internal_mtfft_long(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L, int64_t step, int64_t W, int64_t N,
                           const float* d_tapers, int64_t K, int detrend_type, const void* d_twiddles, void* d_X, hipStream_t st) {
    LongArgs a{};
    a.tapers = d_tapers; a.tw = (const float2*)d_twiddles; a.X = (float2*)d_X;
    a.R = (int)R; a.C = (int)C; a.L = (int)L; a.step = (int)step; a.W = (int)W; a.K = (int)K; a.detrend = detrend_type;
    { const char* d = sc_switch(SC_SW_MTFFT_DEBUG); a.dbg = d ? atoi(d) : 0; }
    switch (N) {
    case 1024: return launch_long<10>(a, d_x, T, st);
    case 2048: return launch_long<11>(a, d_x, T, st);
    case 4096: return launch_long<12>(a, d_x, T, st);
    }
    return SC_EUNSUPPORTED;
}
