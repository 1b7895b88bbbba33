// sc_mtfft_f64.hip -- stage A of the float64 engine as ONE kernel: window extraction + detrend + taper + real FFT +
// transposed store (transforms.py:1147-1171, 1311-1405, 1798-1915), complex128 spectra X[f][w][r][k][c].
//
// The float64 engine first took the three-pass route of the odd window lengths (sc_taper_windows_f64 -> double-precision
// rocFFT in chunks -> rows_to_bins transpose): every sample crosses HBM five times as a double, 14.9 ms for the cfg3
// volume against 1.5 ms for the float32 kernel.  This is the wave-per-pair scheme of mtfft_mixed_wave_kernel
// (sc_mtfft.hip) in doubles: a workgroup owns (window, trial, tile of 2 NF channels); the detrended window tile stays in
// LDS for all tapers; wave w packs channels (2 w, 2 w + 1) of the tile into one complex series, runs the whole
// autosort (Stockham) transform -- radix 5 / 4 / 3 / 2 passes, so powers of two and the next_fast_len lengths alike --
// IN PLACE on its own slice of LDS (a wave's LDS instructions execute in order: no workgroup barrier between passes),
// and the pairs are separated by conjugate symmetry on the way out (DC / Nyquist exactly real, an identically zero
// channel exactly zero), 32 bytes per pair and frequency row, 32 NF bytes contiguous per row.  Twiddles exp(-2 pi i m / N)
// are computed by the workgroup itself (sincospi in fp64, N values: 0.1 ms over the whole launch).
#include <cstdlib>
#include <cstring>
#include "sc_common.h"

typedef double2 zd;
__device__ __forceinline__ zd zd_add(zd a, zd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ zd zd_sub(zd a, zd b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ zd zd_mul(zd a, zd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

struct MdArgs {
    const double* x;
    const double* tapers;   // [K][L], already divided by fs
    zd* X;                  // [F][W][R][K][C]
    int T, R, C, L, step, W, K, detrend;
};

// one radix-R butterfly of a Stockham pass: inputs src[b + t m] (twiddled by W^(t k tw_step), k = b mod Ls), DFT_R in registers
template <int R>
__device__ __forceinline__ void md_bfly_compute(const zd* __restrict__ src, const zd* __restrict__ tw, int b, int m, int Ls,
                                                int tw_step, zd (&v)[R]) {
    const int k = b % Ls;
#pragma unroll
    for (int t = 0; t < R; ++t) {
        v[t] = src[b + t * m];
        if (t > 0 && Ls > 1) v[t] = zd_mul(v[t], tw[t * k * tw_step]);
    }
    if constexpr (R == 2) {
        const zd a = v[0], c = v[1];
        v[0] = zd_add(a, c); v[1] = zd_sub(a, c);
    } else if constexpr (R == 3) {
        constexpr double S3 = 0.86602540378443864676;
        const zd s = zd_add(v[1], v[2]), d = zd_sub(v[1], v[2]);
        const zd t = make_double2(v[0].x - 0.5 * s.x, v[0].y - 0.5 * s.y);
        v[0] = zd_add(v[0], s);
        v[1] = make_double2(t.x + S3 * d.y, t.y - S3 * d.x);      // t - i S3 d
        v[2] = make_double2(t.x - S3 * d.y, t.y + S3 * d.x);      // t + i S3 d
    } else if constexpr (R == 4) {
        const zd s02 = zd_add(v[0], v[2]), d02 = zd_sub(v[0], v[2]), s13 = zd_add(v[1], v[3]), d13 = zd_sub(v[1], v[3]);
        v[0] = zd_add(s02, s13);
        v[2] = zd_sub(s02, s13);
        v[1] = make_double2(d02.x + d13.y, d02.y - d13.x);        // d02 - i d13
        v[3] = make_double2(d02.x - d13.y, d02.y + d13.x);        // d02 + i d13
    } else {
        constexpr double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410;
        constexpr double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;
        const zd a1 = zd_add(v[1], v[4]), a2 = zd_add(v[2], v[3]), b1 = zd_sub(v[1], v[4]), b2 = zd_sub(v[2], v[3]);
        const zd p1 = make_double2(v[0].x + C1 * a1.x + C2 * a2.x, v[0].y + C1 * a1.y + C2 * a2.y);
        const zd p2 = make_double2(v[0].x + C2 * a1.x + C1 * a2.x, v[0].y + C2 * a1.y + C1 * a2.y);
        const zd q1 = make_double2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
        const zd q2 = make_double2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
        v[0] = zd_add(v[0], zd_add(a1, a2));
        v[1] = make_double2(p1.x + q1.y, p1.y - q1.x);            // p1 - i q1
        v[4] = make_double2(p1.x - q1.y, p1.y + q1.x);            // p1 + i q1
        v[2] = make_double2(p2.x + q2.y, p2.y - q2.x);            // p2 - i q2
        v[3] = make_double2(p2.x - q2.y, p2.y + q2.x);            // p2 + i q2
    }
}
// ... and its outputs: dst[(b - k) R + k + t Ls] (autosort: natural order after the last pass)
template <int R>
__device__ __forceinline__ void md_bfly_store(zd* __restrict__ dst, int b, int Ls, const zd (&v)[R]) {
    const int k = b % Ls, base = (b - k) * R + k;
#pragma unroll
    for (int t = 0; t < R; ++t) dst[base + t * Ls] = v[t];
}

#define MD_WAVE_SYNC()                                          \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)

// all passes of one transform, in place, by one wave: a lane pulls the butterflies b = lane, lane + 64, ... of the pass into
// registers, the wave meets, and writes them back
template <int N, int LS>
__device__ __forceinline__ void md_passes_wave(zd* z, const zd* tw, int lane) {
    if constexpr (LS < N) {
        constexpr int rem = N / LS;
        constexpr int R = rem % 5 == 0 ? 5 : (rem % 4 == 0 ? 4 : (rem % 3 == 0 ? 3 : 2));
        constexpr int m = N / R, tw_step = N / (LS * R), ROUNDS = (m + 63) / 64;
        // In place needs every input of the pass read before any output is written: all ROUNDS butterflies of a lane live
        // in registers at once (N = 1024, radix 4: 4 rounds x 4 x 4 registers = 64).
        zd v[ROUNDS][R];
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int b = lane + 64 * j;
            if (b < m) md_bfly_compute<R>(z, tw, b, m, LS, tw_step, v[j]);
        }
        MD_WAVE_SYNC();
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int b = lane + 64 * j;
            if (b < m) md_bfly_store<R>(z, b, LS, v[j]);
        }
        MD_WAVE_SYNC();
        md_passes_wave<N, LS * R>(z, tw, lane);
    }
}

template <int N, int NF>
__global__ void __launch_bounds__(64 * NF) mtfft_f64_kernel(MdArgs p) {
    constexpr int NT = 64 * NF, CT = 2 * NF, XS = CT + 1, F = N / 2 + 1;
    constexpr int LCT = CT == 32 ? 5 : (CT == 16 ? 4 : (CT == 8 ? 3 : 2)), LNF = LCT - 1;
    extern __shared__ __align__(16) unsigned char smem[];
    zd* z = reinterpret_cast<zd*>(smem);                          // [NF][N]; the detrend scratch aliases it
    double* red = reinterpret_cast<double*>(smem);                // [2][NT] + trend [2][CT]
    zd* tw = z + (NF * N > (NT + CT) ? NF * N : (NT + CT));       // [N]
    double* tile = reinterpret_cast<double*>(tw + N);             // [L][XS] (odd stride: the column walks of the trend sums)
    __shared__ int nzf[CT], nbf[CT];                              // channel not identically zero after the detrend / holds a non-finite sample
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < CT) { nzf[tid] = 0; nbf[tid] = 0; }
    const int L = p.L, C = p.C;
    const int c0 = blockIdx.x * CT, r = blockIdx.y, w = blockIdx.z;
    const int64_t RC = (int64_t)p.R * C;
    const double* xw = p.x + ((int64_t)w * p.step * p.R + r) * C + c0;
    for (int idx = tid; idx < L * CT; idx += NT) {
        const int l = idx >> LCT, cc = idx & (CT - 1);
        tile[l * XS + cc] = (c0 + cc < C) ? xw[(int64_t)l * RC + cc] : 0.0;
    }
    for (int i = tid; i < N; i += NT) {
        double s, c;
        sincospi(-2.0 * (double)i / (double)N, &s, &c);
        tw[i] = make_double2(c, s);
    }
    __syncthreads();
    constexpr int SL = NT / CT;
    const int cc = tid & (CT - 1), sl = tid >> LCT;
    if (p.detrend != SC_DETREND_NONE) {
        double s = 0.0, st = 0.0;
        for (int l = sl; l < L; l += SL) {
            const double v = tile[l * XS + cc];
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
            tile[l * XS + cc2] -= red[2 * NT + cc2] * tt + red[2 * NT + CT + cc2];
        }
        __syncthreads();                                          // the scratch is free: z takes its place
    }
    {
        // nzf: not identically zero; nbf: holds a NaN / infinity.  The reference transforms every channel on its own
        // (transforms.py:1402-1405), so a non-finite sample spoils that channel's spectrum only: such a channel leaves the
        // packed transform (zeros in its place: its partner stays clean) and its bins are written as NaN.
        unsigned long long orv = 0ull, mxv = 0ull;              // integer tests on the bit patterns (see mtfft16_kernel)
        for (int l = sl; l < L; l += SL) {
            const unsigned long long u = (unsigned long long)__double_as_longlong(tile[l * XS + cc]) & 0x7fffffffffffffffull;
            orv |= u; mxv = mxv > u ? mxv : u;
        }
        const bool nz = orv != 0ull, bad = mxv >= 0x7ff0000000000000ull;
        if (nz) nzf[cc] = 1;                                      // (plain stores of a constant: many threads may set the same flag)
        if (bad) nbf[cc] = 1;
        __syncthreads();
        if (nbf[cc]) {
            for (int l = sl; l < L; l += SL) tile[l * XS + cc] = 0.0;
        }
        __syncthreads();
    }
    // flags of the pair this thread stores (the store loop advances by NT, a multiple of NF: the pair never changes)
    const int spr = 2 * (tid & (NF - 1));
    const bool na = nbf[spr] != 0, nb = nbf[spr + 1] != 0, za = !na && nzf[spr] == 0, zb = !nb && nzf[spr + 1] == 0;
    const int64_t sF = (int64_t)p.W * p.R * p.K * C;
    zd* zw = z + wave * N;                                        // this wave's pair
    for (int k = 0; k < p.K; ++k) {
        const double* hk = p.tapers + (int64_t)k * L;
        for (int n = lane; n < N; n += 64) {
            zd v = make_double2(0.0, 0.0);
            if (n < L) {
                const double h = hk[n];
                v = make_double2(tile[n * XS + 2 * wave] * h, tile[n * XS + 2 * wave + 1] * h);
            }
            zw[n] = v;
        }
        MD_WAVE_SYNC();
        md_passes_wave<N, 1>(zw, tw, lane);
        __syncthreads();                                          // every pair transformed
        zd* Xk = p.X + (((int64_t)w * p.R + r) * p.K + k) * C + c0;
        for (int idx = tid; idx < F * NF; idx += NT) {
            const int f = idx >> LNF, pr = idx & (NF - 1), c = c0 + 2 * pr;
            if (c >= C) continue;
            const zd u1 = z[pr * N + f], u2 = z[pr * N + (f == 0 ? 0 : N - f)];
            zd A = make_double2(0.5 * (u1.x + u2.x), 0.5 * (u1.y - u2.y));
            zd B = make_double2(0.5 * (u1.y + u2.y), 0.5 * (u2.x - u1.x));
            if (za) A = make_double2(0.0, 0.0);
            if (zb) B = make_double2(0.0, 0.0);
            if (na) A = make_double2(__longlong_as_double(0x7ff8000000000000LL), __longlong_as_double(0x7ff8000000000000LL));
            if (nb) B = make_double2(__longlong_as_double(0x7ff8000000000000LL), __longlong_as_double(0x7ff8000000000000LL));
            zd* d = Xk + (int64_t)f * sF + 2 * pr;
            d[0] = A;                       // (16-byte halves of a 32-byte pair: the write-back L2 merges them -- no streaming hint)
            if (c + 1 < C) d[1] = B;
        }
        __syncthreads();                                          // the next taper refills z
    }
}

// ---- powers of two 64 ... 1024: the register-resident radix-16 scheme of the float32 kernel (sc_mtfft.hip) in doubles ------
// The wave-per-pair kernel above walks every transform through log4 N radix-4 passes in LDS: four write + read round trips of
// the whole sequence per taper at N = 256, 14 bytes of LDS traffic per byte stored -- 5.9 ms for the cfg3 volume, bound by
// the LDS pipe.  Here a thread owns 16 points per pass: a 256-point transform is two radix-16 butterflies in registers with
// ONE exchange through a skewed LDS buffer (phys = idx + idx / 16: conflict-free for the stride-16 writes and the stride-N/16
// reads), +radix 2 / 4 for 512 / 1024; every exchange stays inside one wavefront (N / 16 <= 64 threads per transform), so it
// needs no workgroup barrier; the thread's 16 x 2 detrended samples stay in registers for all tapers.  256 threads, 16 / 8 /
// 4 transforms (32 / 16 / 8 channels) per workgroup at N <= 256 / 512 / 1024.
__device__ __forceinline__ zd zd_mulc(zd a, double c, double s) { return make_double2(a.x * c - a.y * s, a.x * s + a.y * c); }
__device__ __forceinline__ void zd_dft4(zd& a0, zd& a1, zd& a2, zd& a3) {
    const zd b0 = zd_add(a0, a2), b1 = zd_sub(a0, a2), b2 = zd_add(a1, a3), b3 = make_double2(a1.y - a3.y, a3.x - a1.x);
    a0 = zd_add(b0, b2); a1 = zd_add(b1, b3); a2 = zd_sub(b0, b2); a3 = zd_sub(b1, b3);
}
// in: x[n], n = 4 n1 + n2 ; out: o[k], k = k1 + 4 k2   (forward DFT, exp(-2 pi i n k / 16))
__device__ __forceinline__ void zd_dft16(zd (&x)[16], zd (&o)[16]) {
    constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, H = 0.70710678118654752440;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) zd_dft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
    x[4 + 1] = zd_mulc(x[4 + 1], C1, -S1);  x[8 + 1] = zd_mulc(x[8 + 1], H, -H);    x[12 + 1] = zd_mulc(x[12 + 1], S1, -C1);
    x[4 + 2] = zd_mulc(x[4 + 2], H, -H);    x[8 + 2] = make_double2(x[8 + 2].y, -x[8 + 2].x);
    x[12 + 2] = zd_mulc(x[12 + 2], -H, -H);
    x[4 + 3] = zd_mulc(x[4 + 3], S1, -C1);  x[8 + 3] = zd_mulc(x[8 + 3], -H, -H);   x[12 + 3] = zd_mulc(x[12 + 3], -C1, S1);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        zd a0 = x[4 * k1], a1 = x[4 * k1 + 1], a2 = x[4 * k1 + 2], a3 = x[4 * k1 + 3];
        zd_dft4(a0, a1, a2, a3);
        o[k1] = a0; o[k1 + 4] = a1; o[k1 + 8] = a2; o[k1 + 12] = a3;
    }
}
__device__ __forceinline__ void zd_dft8(zd (&x)[8]) {     // natural order in and out
    constexpr double H = 0.70710678118654752440;
    zd e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    zd_dft4(e0, e1, e2, e3);
    zd_dft4(o0, o1, o2, o3);
    o1 = zd_mulc(o1, H, -H);
    o2 = make_double2(o2.y, -o2.x);
    o3 = zd_mulc(o3, -H, -H);
    x[0] = zd_add(e0, o0); x[4] = zd_sub(e0, o0);
    x[1] = zd_add(e1, o1); x[5] = zd_sub(e1, o1);
    x[2] = zd_add(e2, o2); x[6] = zd_sub(e2, o2);
    x[3] = zd_add(e3, o3); x[7] = zd_sub(e3, o3);
}

template <int LOG2N>
__global__ void __launch_bounds__(256, 2) mtfft16_f64_kernel(MdArgs p, int kh) {
    constexpr int N = 1 << LOG2N, TPF = N / 16, NF = 256 / TPF, CT = 2 * NF, XS = CT + 2, ZS = N + N / 16 + 1, F = N / 2 + 1;
    static_assert(TPF <= 64, "every exchange inside one wavefront");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr size_t XT_BYTES = (size_t)N * XS * 8, Z_BYTES = (size_t)NF * ZS * 16;
    constexpr size_t UNION_BYTES = XT_BYTES > Z_BYTES ? XT_BYTES : Z_BYTES;
    double* xt = reinterpret_cast<double*>(smem);                                 // [N][XS] window tile ...
    zd* z = reinterpret_cast<zd*>(smem);                                          // ... then [NF][ZS] exchange buffers
    double* red = reinterpret_cast<double*>(smem + UNION_BYTES);                  // [2][256] + trend [2][CT]; dead after the detrend:
    zd* tw = reinterpret_cast<zd*>(smem + UNION_BYTES);                           // [N] twiddles
    double* hk = reinterpret_cast<double*>(tw + N);                               // [kh][L] tapers (kh = K, or 2 buffers)
    __shared__ int nzf[CT], nbf[CT];
    const int tid = threadIdx.x;
    if (tid < CT) { nzf[tid] = 0; nbf[tid] = 0; }
    const int L = p.L, C = p.C;
    const int c0 = blockIdx.x * CT, r = blockIdx.y, w = blockIdx.z;
    const int64_t RC = (int64_t)p.R * C;
    const double* xw = p.x + ((int64_t)w * p.step * p.R + r) * C + c0;
    const int pf = tid / TPF, i = tid - pf * TPF;          // transform (channel pair) and butterfly index
    zd* zf = z + pf * ZS;
    const int64_t sF = (int64_t)p.W * p.R * p.K * C;
    const bool resident = kh == p.K;
    // window tile: 16-byte loads where the row is aligned
    if ((C & 1) == 0) {
        constexpr int V = CT / 2;
        for (int idx = tid; idx < L * V; idx += 256) {
            const int l = idx / V, cc = 2 * (idx - l * V);
            zd v = make_double2(0.0, 0.0);
            if (c0 + cc + 1 < C) v = *reinterpret_cast<const zd*>(xw + (int64_t)l * RC + cc);
            *reinterpret_cast<zd*>(xt + l * XS + cc) = v;
        }
    } else {
        for (int idx = tid; idx < L * CT; idx += 256) {
            const int l = idx / CT, cc = idx - l * CT;
            xt[l * XS + cc] = (c0 + cc < C) ? xw[(int64_t)l * RC + cc] : 0.0;
        }
    }
    __syncthreads();
    const bool detr = p.detrend != SC_DETREND_NONE;
    if (detr) {
        constexpr int SL = 256 / CT;
        const int cc = tid % CT, sl = tid / CT;
        double s = 0.0, st = 0.0;
        for (int l = sl; l < L; l += SL) {
            const double v = xt[l * XS + cc];
            s += v;
            st += v * (double)(l + 1);
        }
        red[tid] = s;
        red[256 + tid] = st;
        __syncthreads();
        if (tid < CT) {
            double sum = 0.0, sumt = 0.0;
            for (int q = 0; q < SL; ++q) { sum += red[q * CT + tid]; sumt += red[256 + q * CT + tid]; }
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
            red[512 + tid] = a;
            red[512 + CT + tid] = b;
        }
    }
    __syncthreads();                                          // tile and trend coefficients visible
    zd xs[16];                                                // this thread's pass-1 inputs, the same for every taper
    {
        const double invL = 1.0 / (double)L;
        const double a0 = detr ? red[512 + 2 * pf] : 0.0, a1 = detr ? red[512 + 2 * pf + 1] : 0.0;
        const double b0 = detr ? red[512 + CT + 2 * pf] : 0.0, b1 = detr ? red[512 + CT + 2 * pf + 1] : 0.0;
        unsigned long long or0 = 0ull, or1 = 0ull, mx0 = 0ull, mx1 = 0ull;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int n = i + t * TPF;
            zd v = make_double2(0.0, 0.0);
            if (n < L) {
                v = *reinterpret_cast<const zd*>(xt + n * XS + 2 * pf);
                if (detr) {           // the expression of the wave-per-pair kernel
                    const double tt = (double)(n + 1) * invL;
                    v.x -= a0 * tt + b0;
                    v.y -= a1 * tt + b1;
                }
            }
            xs[t] = v;
            const unsigned long long u0 = (unsigned long long)__double_as_longlong(v.x) & 0x7fffffffffffffffull;
            const unsigned long long u1 = (unsigned long long)__double_as_longlong(v.y) & 0x7fffffffffffffffull;
            or0 |= u0; or1 |= u1; mx0 = mx0 > u0 ? mx0 : u0; mx1 = mx1 > u1 ? mx1 : u1;
        }
        // nzf: the channel is not identically zero; nbf: it holds a NaN / infinity (see mtfft_f64_kernel)
        if (or0 != 0ull) nzf[2 * pf] = 1;
        if (or1 != 0ull) nzf[2 * pf + 1] = 1;
        if (mx0 >= 0x7ff0000000000000ull) nbf[2 * pf] = 1;
        if (mx1 >= 0x7ff0000000000000ull) nbf[2 * pf + 1] = 1;
    }
    __syncthreads();                                          // tile and detrend scratch consumed: their space is free
    if (nbf[2 * pf]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) xs[t].x = 0.0;
    }
    if (nbf[2 * pf + 1]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) xs[t].y = 0.0;
    }
    const int spr = 2 * (tid & (NF - 1));                     // the pair this thread stores
    const bool na = nbf[spr] != 0, nb = nbf[spr + 1] != 0, za = !na && nzf[spr] == 0, zb = !nb && nzf[spr + 1] == 0;
    for (int i2 = tid; i2 < N; i2 += 256) {
        double sn, cs;
        sincospi(-2.0 * (double)i2 / (double)N, &sn, &cs);
        tw[i2] = make_double2(cs, sn);
    }
    if (resident) {
        for (int i2 = tid; i2 < p.K * L; i2 += 256) hk[i2] = p.tapers[i2];
    } else {
        for (int i2 = tid; i2 < L; i2 += 256) hk[i2] = p.tapers[i2];              // taper 0 into buffer 0
    }
#define PHYS(idx) ((idx) + ((idx) >> 4))
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (int k = 0; k < p.K; ++k) {
        const double* hkk = resident ? hk + k * L : hk + (k & 1) * L;
        // the next taper travels L2 -> registers under this taper's passes and is parked in the other buffer before the
        // stores go out
        constexpr int HN = (N + 255) / 256;
        double hn[HN];
        const bool fetch_next = !resident && k + 1 < p.K;
        if (fetch_next) {
#pragma unroll
            for (int j = 0; j < HN; ++j) {
                const int n = tid + 256 * j;
                hn[j] = (n < L) ? p.tapers[(int64_t)(k + 1) * L + n] : 0.0;
            }
        }
        __syncthreads();     // taper k (and, first time, the twiddles) visible; the split of taper k - 1 done
        zd a[16], o[16];
        // pass 1: radix 16, inputs straight from registers
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int n = i + t * TPF;
            const double h = (n < L) ? hkk[n] : 0.0;
            a[t] = make_double2(xs[t].x * h, xs[t].y * h);
        }
        zd_dft16(a, o);
#pragma unroll
        for (int u = 0; u < 16; ++u) zf[PHYS(16 * i + u)] = o[u];
        MD_WAVE_SYNC();
        if constexpr (LOG2N < 8) {
            // N = 16 R2 (R2 = 4, 8): pass 2 is radix R2 with P = 16 -- thread i takes the outputs u = i + R2 b, in place
            constexpr int R2 = N / 16;
#pragma unroll
            for (int b = 0; b < 16 / R2; ++b) {
                const int u = i + R2 * b;
                zd q[R2];
#pragma unroll
                for (int j = 0; j < R2; ++j) {
                    const zd v = zf[PHYS(16 * j + u)];
                    q[j] = (j == 0) ? v : zd_mul(v, tw[j * u]);
                }
                if constexpr (R2 == 4) zd_dft4(q[0], q[1], q[2], q[3]); else zd_dft8(q);
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
        } else {
            // pass 2: radix 16, P = 16
            const int kk = i & 15;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const zd v = zf[PHYS(i + t * TPF)];
                a[t] = (t == 0) ? v : zd_mul(v, tw[t * kk * (N / 256)]);
            }
            zd_dft16(a, o);
            if constexpr (LOG2N != 8) MD_WAVE_SYNC();         // N = 256 writes back exactly the slots it read
            const int j = ((i - kk) << 4) + kk;
#pragma unroll
            for (int u = 0; u < 16; ++u) zf[PHYS(j + 16 * u)] = o[u];
            if constexpr (LOG2N == 8) __syncthreads(); else MD_WAVE_SYNC();
        }
        if constexpr (LOG2N == 9) {         // pass 3: radix 2, P = 256, eight butterflies per thread, in place
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int ib = i + b * TPF;
                const zd u0 = zf[PHYS(ib)], u1 = zd_mul(zf[PHYS(ib + 256)], tw[ib]);
                a[2 * b] = zd_add(u0, u1);
                a[2 * b + 1] = zd_sub(u0, u1);
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int ib = i + b * TPF;
                zf[PHYS(ib)] = a[2 * b];
                zf[PHYS(ib + 256)] = a[2 * b + 1];
            }
            __syncthreads();
        }
        if constexpr (LOG2N == 10) {        // pass 3: radix 4, P = 256, four butterflies per thread, in place
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ib = i + b * TPF;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const zd v = zf[PHYS(ib + t * 256)];
                    a[4 * b + t] = (t == 0) ? v : zd_mul(v, tw[t * ib]);
                }
                zd_dft4(a[4 * b], a[4 * b + 1], a[4 * b + 2], a[4 * b + 3]);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int ib = i + b * TPF;
#pragma unroll
                for (int u = 0; u < 4; ++u) zf[PHYS(ib + 256 * u)] = a[4 * b + u];
            }
            __syncthreads();
        }
        if (fetch_next) {
            double* hnext = hk + ((k + 1) & 1) * L;
#pragma unroll
            for (int j = 0; j < HN; ++j) {
                const int n = tid + 256 * j;
                if (n < L) hnext[n] = hn[j];
            }
        }
        // split the packed pair, store X[f][w][r][k][c .. c + 1]: eight rounds of 256 outputs + the Nyquist row
        zd* Xk = p.X + (((int64_t)w * p.R + r) * p.K + k) * C + c0;
        const int pr = tid & (NF - 1), fb = tid / NF, c = c0 + 2 * pr;
        if (c < C) {
            const zd* zp = z + pr * ZS;
            auto put = [&](int f, zd u1, zd u2) {
                zd A = make_double2(0.5 * (u1.x + u2.x), 0.5 * (u1.y - u2.y));
                zd B = make_double2(0.5 * (u1.y + u2.y), 0.5 * (u2.x - u1.x));
                if (za) A = make_double2(0.0, 0.0);
                if (zb) B = make_double2(0.0, 0.0);
                if (na) A = make_double2(qnan, qnan);
                if (nb) B = make_double2(qnan, qnan);
                zd* dst = Xk + (int64_t)f * sF + 2 * pr;
                dst[0] = A;                 // (16-byte halves of a 32-byte pair: the write-back L2 merges them -- no streaming hint;
                if (c + 1 < C) dst[1] = B;  //  with it the kernel takes 6.1 instead of 4.1 ms at cfg3)
            };
#pragma unroll
            for (int h = 0; h < 8; h += 4) {
                zd z1[4], z2[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int f = fb + (h + it) * TPF;
                    z1[it] = zp[PHYS(f)];
                    z2[it] = zp[PHYS((N - f) & (N - 1))];
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) put(fb + (h + it) * TPF, z1[it], z2[it]);
            }
            if (tid < NF) { const zd zn = zp[PHYS(N / 2)]; put(N / 2, zn, zn); }
        }
        // the barrier at the top of the next taper orders these reads before pass 1 rewrites z
    }
#undef PHYS
}

template <int LOG2N>
static int launch_md16(const MdArgs& m, hipStream_t stream) {
    constexpr int N = 1 << LOG2N, TPF = N / 16, NF = 256 / TPF, CT = 2 * NF;
    constexpr size_t xt_b = (size_t)N * (CT + 2) * 8, z_b = (size_t)NF * (N + N / 16 + 1) * 16;
    constexpr size_t uni = xt_b > z_b ? xt_b : z_b, red_b = (size_t)(512 + 2 * CT) * 8;
    auto lds = [&](size_t kh) { const size_t t = (size_t)N * 16 + kh * m.L * 8; return uni + (t > red_b ? t : red_b); };
    constexpr size_t cu_lds = 160 * 1024 - 256;
    const size_t two = lds(2), all = lds(m.K);
    if (two > cu_lds) { sc_set_error("float64 multitaper FFT (N=%d): tile does not fit LDS", N); return SC_EUNSUPPORTED; }
    const int kh = (all <= cu_lds && cu_lds / all == cu_lds / two) ? m.K : 2;        // all tapers resident when it costs no workgroup
    const size_t shmem = kh == m.K ? all : two;
    auto k = mtfft16_f64_kernel<LOG2N>;
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    dim3 grid((unsigned)((m.C + CT - 1) / CT), (unsigned)m.R, (unsigned)m.W);
    hipLaunchKernelGGL(k, grid, dim3(256), shmem, stream, m, kh);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

template <int N, int NF>
static int launch_md(const MdArgs& m, hipStream_t stream) {
    constexpr int NT = 64 * NF, CT = 2 * NF;
    const size_t zb = (size_t)(NF * N > (NT + CT) ? NF * N : (NT + CT)) * 16;
    const size_t lds = zb + (size_t)N * 16 + (size_t)m.L * (CT + 1) * 8 + 16;
    if (lds + 4 * CT > 160 * 1024) { sc_set_error("float64 multitaper FFT (N=%d): tile does not fit LDS", N); return SC_EUNSUPPORTED; }
    auto k = mtfft_f64_kernel<N, NF>;
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)((m.C + CT - 1) / CT), (unsigned)m.R, (unsigned)m.W);
    hipLaunchKernelGGL(k, grid, dim3(NT), lds, stream, m);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

// lengths with a compiled transform (powers of two and the lengths next_fast_len hands out); every other length keeps
// sc_taper_windows_f64 + sc_fft_execute_f64
extern "C" int sc_multitaper_fft_f64_supported(int64_t L, int64_t N) {
    if (L < 1 || L > N) return 0;
    switch (N) {
    case 64: case 128: case 256: case 512: case 1024: case 200: case 250: case 400: case 500: case 1000: return 1;
    default: return 0;
    }
}

extern "C" int sc_multitaper_fft_f64(const double* d_x, int64_t T, int64_t R, int64_t C, int64_t L, int64_t step,
                                     int64_t W, int64_t N, const double* d_tapers, int64_t K, int detrend_type,
                                     void* d_X, void* stream) {
    ScTimed timed_("mtfft_fused_f64", stream);
    SC_REQUIRE(d_x && d_tapers && d_X, "NULL device pointer");
    SC_REQUIRE(T >= 1 && R >= 1 && C >= 1 && L >= 1 && step >= 1 && W >= 1 && K >= 1, "dimensions must be positive");
    SC_REQUIRE((W - 1) * step + L <= T, "windows exceed the time series");
    SC_REQUIRE(detrend_type >= 0 && detrend_type <= 2, "unknown detrend_type");
    SC_REQUIRE(R <= 65535 && W <= 65535, "too many trials/windows for one launch");
    if (!sc_multitaper_fft_f64_supported(L, N)) {
        sc_set_error("fused float64 multitaper FFT: no compiled transform for N=%lld (L=%lld); use sc_taper_windows_f64 + "
                     "sc_fft_execute_f64", (long long)N, (long long)L);
        return SC_EUNSUPPORTED;
    }
    MdArgs m{d_x, d_tapers, (zd*)d_X, (int)T, (int)R, (int)C, (int)L, (int)step, (int)W, (int)K, detrend_type};
    hipStream_t s = (hipStream_t)stream;
    // powers of two: the radix-16 kernel (SC_MTFFT_F64=wave, diagnostic: the wave-per-pair kernel for every length)
    const char* sel = sc_switch(SC_SW_MTFFT_F64);
    const bool wave_only = sel && strcmp(sel, "wave") == 0;
    if (!wave_only) {
        switch (N) {
        case 64: return launch_md16<6>(m, s);
        case 128: return launch_md16<7>(m, s);
        case 256: return launch_md16<8>(m, s);
        case 512: return launch_md16<9>(m, s);
        case 1024: return launch_md16<10>(m, s);
        default: break;
        }
    }
    switch (N) {
    case 64: return launch_md<64, 8>(m, s);
    case 128: return launch_md<128, 8>(m, s);
    case 256: return launch_md<256, 8>(m, s);
    case 512: return launch_md<512, 4>(m, s);
    case 1024: return launch_md<1024, 4>(m, s);
    case 200: return launch_md<200, 8>(m, s);
    case 250: return launch_md<250, 8>(m, s);
    case 400: return launch_md<400, 4>(m, s);
    case 500: return launch_md<500, 4>(m, s);
    case 1000: return launch_md<1000, 4>(m, s);
    }
    return SC_EUNSUPPORTED;
}
