// sc_wilson_pair.hip -- pairwise spectral Granger with the WHOLE 2 x 2 Wilson iteration of a channel pair on one compute unit
// (round 5).
//
// The batched form of sc_wilson.hip keeps S, G and A of every (window, pair) problem in HBM -- 160 bytes per problem and
// two-sided bin -- and streams them through three kernels per iteration: at BASELINE configs[3] (2016 pairs x 4096 bins x 24
// iterations) that is 82 GB of state round trips for 0.5 GB of spectra, 17.3 of the 18.7 ms of the step, with a host
// synchronisation every four iterations.  Here one workgroup owns a problem from its cross-spectra to its converged factor:
//
//   * real time series give S(-f) = conj S(f), hence G(-f) = conj G(f) and A(-f) = conj A(f): only the N/2 + 1 non-negative
//     bins are kept -- G (4 complex) and S (4 doubles) of a thread's eight bins live in REGISTERS for the whole iteration,
//     nothing of the state is ever written to HBM;
//   * the four entries of a(n) = ifft(A) are REAL sequences, so the causal projection (minimum_phase_decomposition.py:96-142)
//     takes TWO complex transforms per direction instead of four: z1 = a00 + i a11, z2 = a01 + i a10 ("two for one"), masked in
//     the time domain (lag 0 halved, its strict lower triangle zeroed, lags >= N/2 zeroed), transformed back and split by
//     conjugate symmetry;
//   * the transforms are the register-resident fp64 radix-16 passes of sc_wilson_fft.h (16 points per thread, exchanges through
//     LDS): the N/16 threads of a problem run z1, then z2; 256 threads = one problem per workgroup at N = 4096, sixteen at
//     N = 256.  One workgroup per compute unit (139 KB of LDS for the two series), i.e. one wave per SIMD with the whole
//     512-register file to itself: G and S of eight bins per thread (192 registers) sit beside the transform's 128;
//   * G <- G A+, max |G - G_old| and the convergence test (minimum_phase_decomposition.py:145-181, :301-315) happen in the same
//     kernel: every problem stops at ITS convergence without the host looking (the batched form polls every four iterations).
//
// Arithmetic is the reference's (A = G^-1 (G^-1 S)^H + I in closed form, the same mask, the same update and test) in float64; what
// differs from the batched kernels is the rounding of the packed transforms (1e-13 of the factor, tests/test_gpu_parity.py).
// Applies to records of real series (N/2 + 1 accumulated bins) and N = 256 ... 4096 (round 6: also fourteen lengths 200 ... 4000 that are
// not powers of two, wilson_pair_mixed_kernel); everything else -- uploaded two-sided
// coefficients, other lengths, sc_wilson_factor_f64 -- stays on sc_wilson.hip.  SC_GRANGER_KERNEL=batched forces that path.
#include <string.h>
#include "sc_wilson_fft.h"

__device__ __forceinline__ cd pz_mul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cd pz_conj(cd a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cd pz_add(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd pz_sub(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cd pz_div(cd a, cd b) {
    const double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

struct PairArgs {
    ScRec accum;
    const int32_t* pairs;
    int64_t P, n_pairs, n_batch;
    int64_t floats_per_bin;
    int C, NB, n_tiles, p_csm;
    double n_obs;
    double* chol;            // [P][4]: l00, l10, l11 of the lag-0 covariance's Cholesky factor (identity after a restart), spare
    int32_t* batch_bad;      // [n_batch]
    cd* Ghalf;               // [P][4][N / 2 + 1]
    double* h0;              // [P][4]
    int32_t* n_iter;
    int32_t* status;
    int32_t* summary;        // [0] most iterations of a problem, [1] problems not converged, [2] problems restarted from the identity
    double tol;
    int max_iter;
};

// the pair's cross-spectra at one accumulated bin: s00, s11, Re s01, Im s01 (expectation = sum / n_obs)
__device__ __forceinline__ void pair_read_S(const PairArgs& a, int64_t g, int ci, int cj, int64_t F, int f, double (&s)[4]) {
    const ScRec rec = a.accum + (g * F + f) * a.floats_per_bin;
    auto rd = [&](int plane, int i, int j, bool* mirrored) -> double {
        int ti = i >> 4, tj = j >> 4, ii = i & 15, jj = j & 15;
        const bool m = (ti > tj) || (ti == tj && ii > jj);
        if (m) { int t = ti; ti = tj; tj = t; t = ii; ii = jj; jj = t; }
        *mirrored = m;
        return rec[((int64_t)plane * a.n_tiles + sc_tile_index(ti, tj, a.NB)) * SC_TILE_ELEMS + ii * 16 + jj];
    };
    bool m, mm;
    s[0] = rd(a.p_csm, ci, ci, &mm) / a.n_obs;
    s[1] = rd(a.p_csm, cj, cj, &mm) / a.n_obs;
    s[2] = rd(a.p_csm, ci, cj, &m) / a.n_obs;
    const double im = rd(a.p_csm + 1, ci, cj, &m) / a.n_obs;
    s[3] = m ? -im : im;
}

// A = G^-1 (G^-1 S)^H + I = G^-1 S G^-H + I at one frequency, closed form.  Against sc_wilson.hip's predict2x2: ONE reciprocal of the
// determinant instead of four complex divisions (an fp64 division is a dozen instructions, and this is the kernel's inner loop), and
// the Hermitian structure of A used -- a00, a11 are real (only their real parts are formed), a10 = conj(a01) is not computed at
// all: A[0] = (a00, 0), A[1] = a01, A[2] = a11 as (a11, 0).
__device__ __forceinline__ void pair_predict(const cd (&g)[4], const double (&sv)[4], cd (&A)[3]) {
    const cd s01 = make_double2(sv[2], sv[3]), s10 = pz_conj(s01);
    const cd det = pz_sub(pz_mul(g[0], g[3]), pz_mul(g[1], g[2]));
    const double rd = 1.0 / (det.x * det.x + det.y * det.y);
    const cd idet = make_double2(det.x * rd, -det.y * rd);
    const cd i00 = pz_mul(g[3], idet), i01 = pz_mul(make_double2(-g[1].x, -g[1].y), idet);
    const cd i10 = pz_mul(make_double2(-g[2].x, -g[2].y), idet), i11 = pz_mul(g[0], idet);
    // X = G^-1 S (s00, s11 real)
    const cd x00 = pz_add(make_double2(i00.x * sv[0], i00.y * sv[0]), pz_mul(i01, s10)), x01 = pz_add(pz_mul(i00, s01), make_double2(i01.x * sv[1], i01.y * sv[1]));
    const cd x10 = pz_add(make_double2(i10.x * sv[0], i10.y * sv[0]), pz_mul(i11, s10)), x11 = pz_add(pz_mul(i10, s01), make_double2(i11.x * sv[1], i11.y * sv[1]));
    // A = G^-1 X^H + I
    A[0] = make_double2(i00.x * x00.x + i00.y * x00.y + i01.x * x01.x + i01.y * x01.y + 1.0, 0.0);
    A[1] = pz_add(pz_mul(i00, pz_conj(x10)), pz_mul(i01, pz_conj(x11)));
    A[2] = make_double2(i10.x * x10.x + i10.y * x10.y + i11.x * x11.x + i11.y * x11.y + 1.0, 0.0);
}

// Lag-0 covariance R0 = mean over the N two-sided bins of Re S = (S(0) + S(N/2) + 2 sum_{0 < f < N/2} Re S(f)) / N and its
// Cholesky factor; a covariance that is not positive definite flags its batch (the windows of one pair: sc_wilson.hip k_init).
__global__ void __launch_bounds__(256) pair_lag0_kernel(PairArgs a, int64_t N) {
    __shared__ double red[3][256];
    const int64_t p = blockIdx.x, F = N / 2 + 1;
    const int64_t g = p / a.n_pairs, pr = p % a.n_pairs;
    const int ci = a.pairs[2 * pr], cj = a.pairs[2 * pr + 1];
    double r0 = 0, r1 = 0, r2 = 0;
    for (int f = threadIdx.x; f < (int)F; f += 256) {
        double s[4];
        pair_read_S(a, g, ci, cj, F, f, s);
        const double w = (f == 0 || f == (int)(N / 2)) ? 1.0 : 2.0;
        r0 += w * s[0]; r1 += w * s[1]; r2 += w * s[2];
    }
    red[0][threadIdx.x] = r0; red[1][threadIdx.x] = r1; red[2][threadIdx.x] = r2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st)
            for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double r00 = red[0][0] / (double)N, r11 = red[1][0] / (double)N, r01 = red[2][0] / (double)N;
        double l00 = sqrt(r00), l10 = r01 / l00;
        const double t = r11 - l10 * l10;
        double l11 = sqrt(t);
        const bool bad = !(r00 > 0.0) || !(t > 0.0);
        if (bad) { l00 = 1.0; l10 = 0.0; l11 = 1.0; atomicOr(a.batch_bad + p % a.n_batch, 1); }
        a.chol[p * 4] = l00; a.chol[p * 4 + 1] = l10; a.chol[p * 4 + 2] = l11;
    }
}
// every problem of a flagged batch starts from the identity (the expectation of the reference's random restart: sc_wilson.hip)
__global__ void pair_restart_kernel(PairArgs a) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.P || !a.batch_bad[p % a.n_batch]) return;
    a.chol[p * 4] = 1.0; a.chol[p * 4 + 1] = 0.0; a.chol[p * 4 + 2] = 1.0;
    atomicAdd(a.summary + 2, 1);
}

// HALVES = 1: 256 threads, the N/16 threads of a problem transform z1, then z2, and hold eight bins each (one wave per SIMD, the
// whole register file); HALVES = 2: 512 threads, the two halves of a problem's N/8 threads transform z1 and z2 side by side and hold
// four bins each (two waves per SIMD, 256 registers).
// Which one: 512 threads need the state AND the transform in 256 registers -- they fit at N = 256 (244 registers); from N = 512 on the
// third transform pass pushes 31 ... 209 registers to scratch, so the longer windows take the 256-thread form (456 registers at
// N = 4096, the overflow in AGPRs, no scratch).
template <int LOG2N, int HALVES>
__global__ void __launch_bounds__(256 * HALVES, HALVES) wilson_pair_kernel(PairArgs a) {
    constexpr int WP_THREADS = 256 * HALVES, WP_BINS = 8 / HALVES;
    constexpr int N = 1 << LOG2N, H = N / 2, TPF = N / 16, TPP = TPF * HALVES, PPW = WP_THREADS / TPP, ZS = N + N / 16, NHI = N / 64;
    constexpr int64_t F = N / 2 + 1;
    extern __shared__ __align__(16) unsigned char wp_smem[];
    cd* z = reinterpret_cast<cd*>(wp_smem);                               // [PPW][2][ZS]: staging (natural order) / transform exchange
    cd* lo = z + PPW * 2 * ZS;
    cd* hi = lo + 64;
    cd* gny = hi + NHI;                                                   // [PPW][4]: G at the Nyquist bin (thread 0 of the problem)
    double* sny = reinterpret_cast<double*>(gny + PPW * 4);              // [PPW][4]: S there
    unsigned long long* errs = reinterpret_cast<unsigned long long*>(sny + PPW * 4);   // [PPW]: max |G - G_old|^2 (bit pattern)
    const int tid = threadIdx.x, q = tid / TPP, j = tid % TPP;          // j: the thread's index inside its problem
    const int hh = j / TPF, i = j % TPF;                                 // (HALVES = 2) the series this thread transforms, its index there
    const int64_t p = (int64_t)blockIdx.x * PPW + q;
    const bool valid = p < a.P;
    if (tid < 64 + NHI) {
        const int m = tid < 64 ? tid : (tid - 64) * 64;
        double s, c;
        sincospi(-2.0 * (double)m / (double)N, &s, &c);
        (tid < 64 ? lo[tid] : hi[tid - 64]) = make_double2(c, s);
    }
    if (tid < PPW) errs[tid] = 0ull;
    cd* b1 = z + (q * 2) * ZS;              // z1 = a00 + i a11
    cd* b2 = b1 + ZS;                        // z2 = a01 + i a10

    // ---- the problem's state: this thread's bins f = j + TPP u (u < WP_BINS), the Nyquist bin with thread 0 -----------------------
    cd G[WP_BINS][4];
    double S[WP_BINS][4];
    int ci = 0, cj = 0;
    int64_t grp = 0;
    if (valid) {
        grp = p / a.n_pairs;
        const int64_t pr = p % a.n_pairs;
        ci = a.pairs[2 * pr]; cj = a.pairs[2 * pr + 1];
    }
    const double l00 = valid ? a.chol[p * 4] : 1.0, l10 = valid ? a.chol[p * 4 + 1] : 0.0, l11 = valid ? a.chol[p * 4 + 2] : 1.0;
#pragma unroll
    for (int u = 0; u < WP_BINS; ++u) {
        if (valid) pair_read_S(a, grp, ci, cj, F, j + TPP * u, S[u]);
        else { S[u][0] = 1.0; S[u][1] = 1.0; S[u][2] = 0.0; S[u][3] = 0.0; }
        G[u][0] = make_double2(l00, 0); G[u][1] = make_double2(l10, 0); G[u][2] = make_double2(0, 0); G[u][3] = make_double2(l11, 0);
    }
    if (j == 0) {
        double s[4] = {1.0, 1.0, 0.0, 0.0};
        if (valid) pair_read_S(a, grp, ci, cj, F, H, s);
#pragma unroll
        for (int e = 0; e < 4; ++e) sny[q * 4 + e] = s[e];
        gny[q * 4] = make_double2(l00, 0); gny[q * 4 + 1] = make_double2(l10, 0);
        gny[q * 4 + 2] = make_double2(0, 0); gny[q * 4 + 3] = make_double2(l11, 0);
    }
    // A(f) -> the packed spectra at f and (conjugate symmetry of the entries) at N - f:  z1 = a00 + i a11 with a00, a11 real is the
    // same at both; z2 = a01 + i a10 with a10 = conj(a01) is (re + im)(1 + i) at f and (re - im)(1 + i) at N - f
    auto stage = [&](int f, const cd (&A)[3]) {
        const cd z1 = make_double2(A[0].x, A[2].x);
        const double sp = A[1].x + A[1].y, sm = A[1].x - A[1].y;
        b1[f] = z1;
        b2[f] = make_double2(sp, sp);
        if (f != 0 && f != H) {
            b1[N - f] = z1;
            b2[N - f] = make_double2(sm, sm);
        }
    };
    __syncthreads();                          // (tables, errs, the Nyquist state)
#pragma unroll
    for (int u = 0; u < WP_BINS; ++u) {
        cd A[3];
        pair_predict(G[u], S[u], A);
        stage(j + TPP * u, A);
    }
    if (j == 0) {
        cd g[4], A[3];
        double s[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[e] = gny[q * 4 + e]; s[e] = sny[q * 4 + e]; }
        pair_predict(g, s, A);
        stage(H, A);
    }
    bool done = !valid;
    int n_it = 0;
    const double invN = 1.0 / (double)N, tol2 = a.tol * a.tol;
    for (int it = 0; it < a.max_iter; ++it) {
        __syncthreads();                      // z1, z2 staged
        // ---- a = ifft(z) = conj(fft(conj z)) / N, masked; A+ = fft(a+): z1, then z2 ---------------------------------------------
#pragma unroll 1
        for (int hs_ = 0; hs_ < 2 / HALVES; ++hs_) {
            const int h = HALVES == 2 ? hh : hs_;
            cd* zf = h ? b2 : b1;
            cd v[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const cd x = zf[i + t * TPF];
                v[t] = make_double2(x.x, -x.y);
            }
            wf_fft<LOG2N>(v, zf, lo, hi, i);
            // the mask in the time domain: lags n = j + t N/16 >= N/2 are exactly the registers t >= 8 -- written as constants, so
            // that the compiler drops the half of the inverse's last pass that computed them and the half of the forward's first pass
            // that would read them (zeros)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (t >= 8) { v[t] = make_double2(0.0, 0.0); continue; }
                double sr = invN, si = invN;
                if (t == 0 && i == 0) { sr *= 0.5; si = h ? 0.0 : 0.5 * si; }       // lag 0: halved; z2's imaginary part is a10, the strict lower triangle
                v[t] = make_double2(v[t].x * sr, -v[t].y * si);
            }
            wf_fft<LOG2N>(v, zf, lo, hi, i);
            __syncthreads();                  // the last pass has read the exchange buffer
#pragma unroll
            for (int t = 0; t < 16; ++t) zf[i + t * TPF] = v[t];
        }
        __syncthreads();
        // ---- split, G <- G A+, max |G - G_old|; the next A = predict(G) goes straight back into the staging buffers -------------
        double e2 = 0.0;
        auto update = [&](int f, cd (&g)[4], const double (&s)[4]) {
            const int fm = (N - f) & (N - 1);
            const cd p1 = b1[f], m1 = b1[fm], p2 = b2[f], m2 = b2[fm];
            cd Ap[4];
            Ap[0] = make_double2(0.5 * (p1.x + m1.x), 0.5 * (p1.y - m1.y));
            Ap[3] = make_double2(0.5 * (p1.y + m1.y), 0.5 * (m1.x - p1.x));
            Ap[1] = make_double2(0.5 * (p2.x + m2.x), 0.5 * (p2.y - m2.y));
            Ap[2] = make_double2(0.5 * (p2.y + m2.y), 0.5 * (m2.x - p2.x));
            if (!done) {
                cd n[4];
                n[0] = pz_add(pz_mul(g[0], Ap[0]), pz_mul(g[1], Ap[2])); n[1] = pz_add(pz_mul(g[0], Ap[1]), pz_mul(g[1], Ap[3]));
                n[2] = pz_add(pz_mul(g[2], Ap[0]), pz_mul(g[3], Ap[2])); n[3] = pz_add(pz_mul(g[2], Ap[1]), pz_mul(g[3], Ap[3]));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const cd d = pz_sub(n[k], g[k]);
                    e2 = fmax(e2, d.x * d.x + d.y * d.y);          // (the square root is taken once, of the maximum)
                    g[k] = n[k];
                }
            }
            cd A[3];
            pair_predict(g, s, A);
            stage(f, A);
        };
#pragma unroll
        for (int u = 0; u < WP_BINS; ++u) update(j + TPP * u, G[u], S[u]);
        if (j == 0) {
            cd g[4];
            double s[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { g[k] = gny[q * 4 + k]; s[k] = sny[q * 4 + k]; }
            update(H, g, s);
#pragma unroll
            for (int k = 0; k < 4; ++k) gny[q * 4 + k] = g[k];
        }
        if (!done && e2 > 0.0) atomicMax(errs + q, (unsigned long long)__double_as_longlong(e2));      // non-negative doubles order like their bits
        __syncthreads();
        if (!done) {
            ++n_it;
            // max |G - G_old| < tol (minimum_phase_decomposition.py:170-181), compared as squares
            if (__longlong_as_double((long long)errs[q]) < tol2) done = true;
        }
        const int running = __syncthreads_or(done ? 0 : 1);
        if (j == 0) errs[q] = 0ull;           // (the next atomics are several barriers away)
        if (!running) break;
    }
    // ---- results: G on the non-negative bins, H0 = Re ifft(G)[lag 0], iteration count and status ---------------------------------
    __syncthreads();
    double hs[4] = {0, 0, 0, 0};
    if (valid) {
        cd* Gp = a.Ghalf + p * 4 * F;
#pragma unroll
        for (int u = 0; u < WP_BINS; ++u) {
            const int f = j + TPP * u;
            const double w = f == 0 ? 1.0 : 2.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { Gp[k * F + f] = G[u][k]; hs[k] += w * G[u][k].x; }
        }
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const cd g = gny[q * 4 + k]; Gp[k * F + H] = g; hs[k] += g.x; }
        }
    }
    double* red = reinterpret_cast<double*>(z);                           // [WP_THREADS][4]
#pragma unroll
    for (int k = 0; k < 4; ++k) red[tid * 4 + k] = hs[k];
    __syncthreads();
    if (valid && j < 4) {
        double t = 0.0;
        for (int m = 0; m < TPP; ++m) t += red[(q * TPP + m) * 4 + j];       // fixed order
        a.h0[p * 4 + j] = t / (double)N;
    }
    if (valid && j == 0) {
        a.n_iter[p] = n_it;
        a.status[p] = done ? 1 : 0;
        atomicMax(a.summary, n_it);
        if (!done) atomicAdd(a.summary + 1, 1);
    }
}

// ---- windows that are NOT powers of two (round 6) -----------------------------------------------------------------------------------
// next_fast_len hands the reference 1000, 2000, 4000 ... samples in the ordinary lab setting (1 / 2 / 4 s at 1 kHz); those windows ran the
// batched kernels of sc_wilson.hip -- state in HBM, three kernels + two library transforms per iteration: 18.6 ms at the shape of BASELINE
// configs[3] with 4000 samples where the kernel above takes 3.1 at 4096.  Same kernel, other transform: N = P M with P = 16, 8, 4 or 2
// points per thread and M = N / P <= 256 threads per problem, M = 2^a 3^b 5^c.  A transform is Cooley-Tukey with the P-point DFT in
// registers first (over t: the thread's own points x[i + t M]), the twiddles W_N^(i k1), then P transforms of length M across the
// threads: Stockham passes of radix 5 / 4 / 3 / 2 IN LDS -- every thread takes its share of the pass's P M / R butterflies into
// registers, the workgroup meets, and the outputs go back to the same rows -- and a gather X[i + t M] = row (k mod P), entry k / P.
// The thread's points and bins are those of the kernel above (lags >= N / 2 are the registers t >= P / 2, bins f = j + M u), so the
// staging, the mask, the update and the convergence test are unchanged.
constexpr int wpm_radix(int M, int pass) {
    int m = M;
    for (int p = 0;; ++p) {
        const int r = (m % 5 == 0) ? 5 : (m % 4 == 0) ? 4 : (m % 3 == 0) ? 3 : (m % 2 == 0) ? 2 : 1;
        if (p == pass || r == 1) return r;
        m /= r;
    }
}
constexpr bool wpm_len_ok(int M) {
    int m = M;
    while (m % 5 == 0) m /= 5;
    while (m % 3 == 0) m /= 3;
    while (m % 2 == 0) m /= 2;
    return m == 1 && M >= 1 && M <= 512;
}
constexpr int wpm_row(int M) { return M | 1; }             // row length of the P rows in LDS (odd)

template <int R>
__device__ __forceinline__ void wpm_dft(cd (&v)[R]) {
    if constexpr (R == 2) {
        zdft2(v[0], v[1]);
    } else if constexpr (R == 3) {
        constexpr double S3 = 0.86602540378443864676;
        const cd s = make_double2(v[1].x + v[2].x, v[1].y + v[2].y), d = make_double2(v[1].x - v[2].x, v[1].y - v[2].y);
        const cd t = make_double2(v[0].x - 0.5 * s.x, v[0].y - 0.5 * s.y);
        v[0] = make_double2(v[0].x + s.x, v[0].y + s.y);
        v[1] = make_double2(t.x + S3 * d.y, t.y - S3 * d.x);      // t - i S3 d
        v[2] = make_double2(t.x - S3 * d.y, t.y + S3 * d.x);      // t + i S3 d
    } else if constexpr (R == 4) {
        zdft4(v[0], v[1], v[2], v[3]);
    } else {
        static_assert(R == 5, "radix 2, 3, 4 or 5");
        constexpr double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410;
        constexpr double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;
        const cd a1 = make_double2(v[1].x + v[4].x, v[1].y + v[4].y), a2 = make_double2(v[2].x + v[3].x, v[2].y + v[3].y);
        const cd b1 = make_double2(v[1].x - v[4].x, v[1].y - v[4].y), b2 = make_double2(v[2].x - v[3].x, v[2].y - v[3].y);
        const cd p1 = make_double2(v[0].x + C1 * a1.x + C2 * a2.x, v[0].y + C1 * a1.y + C2 * a2.y);
        const cd p2 = make_double2(v[0].x + C2 * a1.x + C1 * a2.x, v[0].y + C2 * a1.y + C1 * a2.y);
        const cd q1 = make_double2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
        const cd q2 = make_double2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
        v[0] = make_double2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
        v[1] = make_double2(p1.x + q1.y, p1.y - q1.x);            // p1 - i q1
        v[4] = make_double2(p1.x - q1.y, p1.y + q1.x);
        v[2] = make_double2(p2.x + q2.y, p2.y - q2.x);            // p2 - i q2
        v[3] = make_double2(p2.x - q2.y, p2.y + q2.x);
    }
}
// One Stockham pass of radix R over the P rows of length M: butterfly b of a row reads row[b + t M / R] (twiddled by W_(Ls R)^(t k),
// k = b mod Ls) and writes row[(b - k) R + k + t Ls] -- natural order after the last pass.  All threads of the problem must call.
template <int P, int M, int R>
__device__ __forceinline__ void wpm_pass(cd* z, const cd* lo, const cd* hi, int i, int Ls) {
    constexpr int N = P * M, MS = wpm_row(M), m = M / R, NBF = (P * m + M - 1) / M;
    cd v[NBF][R];
    const int twm = N / (Ls * R);
#pragma unroll
    for (int u = 0; u < NBF; ++u) {
        const int q = i + u * M;
        if (q < P * m) {
            const int row = q / m, b = q - row * m, k = b % Ls;
            const cd* src = z + row * MS;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                v[u][t] = src[b + t * m];
                if (t > 0 && Ls > 1) {
                    const int e = t * k * twm;
                    v[u][t] = zmul(v[u][t], zmul(lo[e & 63], hi[e >> 6]));
                }
            }
            wpm_dft<R>(v[u]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NBF; ++u) {
        const int q = i + u * M;
        if (q < P * m) {
            const int row = q / m, b = q - row * m, k = b % Ls;
            cd* dst = z + row * MS + (b - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) dst[t * Ls] = v[u][t];
        }
    }
    __syncthreads();
}
template <int P, int M, int PASS, int LS>
__device__ __forceinline__ void wpm_passes(cd* z, const cd* lo, const cd* hi, int i) {
    constexpr int R = wpm_radix(M, PASS);
    if constexpr (R > 1 && LS < M) {
        wpm_pass<P, M, R>(z, lo, hi, i, LS);
        wpm_passes<P, M, PASS + 1, LS * R>(z, lo, hi, i);
    }
}
// One forward transform of a problem's series: a[t] = x[i + t M] in, a[t] = X[i + t M] out; zf: P rows of wpm_row(M) elements.
template <int P, int M>
__device__ __forceinline__ void wpm_fft(cd (&a)[P], cd* zf, const cd* lo, const cd* hi, int i) {
    constexpr int MS = wpm_row(M);
    __syncthreads();                    // the previous reads of zf are done
    if constexpr (P == 16) {
        cd o[16];
        zdft16(a, o);
#pragma unroll
        for (int t = 0; t < 16; ++t) a[t] = o[t];
    } else if constexpr (P == 8) {
        zdft8(a);
    } else if constexpr (P == 4) {
        zdft4(a[0], a[1], a[2], a[3]);
    } else {
        zdft2(a[0], a[1]);
    }
#pragma unroll
    for (int k1 = 0; k1 < P; ++k1) {
        const int e = i * k1;                                   // < N
        zf[k1 * MS + i] = k1 == 0 ? a[0] : zmul(a[k1], zmul(lo[e & 63], hi[e >> 6]));
    }
    __syncthreads();
    wpm_passes<P, M, 0, 1>(zf, lo, hi, i);
#pragma unroll
    for (int t = 0; t < P; ++t) {
        const int k = i + t * M;
        a[t] = zf[(k % P) * MS + k / P];
    }
}

// TB: thread budget of a workgroup (256: one wave per SIMD with the whole register file, the form of the kernel above; 512: two waves per
// SIMD on 256 registers -- eight points and four bins per thread fit them, sixteen and eight do not)
template <int P, int M, int TB>
__global__ void __launch_bounds__(TB) wilson_pair_mixed_kernel(PairArgs a) {
    static_assert(wpm_len_ok(M) && M <= TB && (P == 16 || P == 8 || P == 4 || P == 2), "N = P M, M = 2^a 3^b 5^c <= TB");
    constexpr int N = P * M, H = N / 2, WP_BINS = P / 2, PPW = TB / M, WP_THREADS = PPW * M;
    constexpr int ZS = (P * wpm_row(M) > N ? P * wpm_row(M) : N), NHI = (N + 63) / 64;
    constexpr int64_t F = N / 2 + 1;
    extern __shared__ __align__(16) unsigned char wp_smem[];
    cd* z = reinterpret_cast<cd*>(wp_smem);                               // [PPW][2][ZS]: staging (natural order) / transform rows
    cd* lo = z + PPW * 2 * ZS;
    cd* hi = lo + 64;
    cd* gny = hi + NHI;
    double* sny = reinterpret_cast<double*>(gny + PPW * 4);
    unsigned long long* errs = reinterpret_cast<unsigned long long*>(sny + PPW * 4);
    const int tid = threadIdx.x, q = tid / M, j = tid % M, i = j;        // (launched with PPW M threads)
    const int64_t p = (int64_t)blockIdx.x * PPW + q;
    const bool valid = p < a.P;
    for (int t = tid; t < 64 + NHI; t += WP_THREADS) {
        const int m = t < 64 ? t : (t - 64) * 64;
        double s, c;
        sincospi(-2.0 * (double)m / (double)N, &s, &c);
        (t < 64 ? lo[t] : hi[t - 64]) = make_double2(c, s);
    }
    if (tid < PPW) errs[tid] = 0ull;
    cd* b1 = z + (q * 2) * ZS;              // z1 = a00 + i a11
    cd* b2 = b1 + ZS;                        // z2 = a01 + i a10
    cd G[WP_BINS][4];
    double S[WP_BINS][4];
    int ci = 0, cj = 0;
    int64_t grp = 0;
    if (valid) {
        grp = p / a.n_pairs;
        const int64_t pr = p % a.n_pairs;
        ci = a.pairs[2 * pr]; cj = a.pairs[2 * pr + 1];
    }
    const double l00 = valid ? a.chol[p * 4] : 1.0, l10 = valid ? a.chol[p * 4 + 1] : 0.0, l11 = valid ? a.chol[p * 4 + 2] : 1.0;
#pragma unroll
    for (int u = 0; u < WP_BINS; ++u) {
        if (valid) pair_read_S(a, grp, ci, cj, F, j + M * u, S[u]);
        else { S[u][0] = 1.0; S[u][1] = 1.0; S[u][2] = 0.0; S[u][3] = 0.0; }
        G[u][0] = make_double2(l00, 0); G[u][1] = make_double2(l10, 0); G[u][2] = make_double2(0, 0); G[u][3] = make_double2(l11, 0);
    }
    if (j == 0) {
        double s[4] = {1.0, 1.0, 0.0, 0.0};
        if (valid) pair_read_S(a, grp, ci, cj, F, H, s);
#pragma unroll
        for (int e = 0; e < 4; ++e) sny[q * 4 + e] = s[e];
        gny[q * 4] = make_double2(l00, 0); gny[q * 4 + 1] = make_double2(l10, 0);
        gny[q * 4 + 2] = make_double2(0, 0); gny[q * 4 + 3] = make_double2(l11, 0);
    }
    auto stage = [&](int f, const cd (&A)[3]) {
        const cd z1 = make_double2(A[0].x, A[2].x);
        const double sp = A[1].x + A[1].y, sm = A[1].x - A[1].y;
        b1[f] = z1;
        b2[f] = make_double2(sp, sp);
        if (f != 0 && f != H) {
            b1[N - f] = z1;
            b2[N - f] = make_double2(sm, sm);
        }
    };
    __syncthreads();
#pragma unroll
    for (int u = 0; u < WP_BINS; ++u) {
        cd A[3];
        pair_predict(G[u], S[u], A);
        stage(j + M * u, A);
    }
    if (j == 0) {
        cd g[4], A[3];
        double s[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[e] = gny[q * 4 + e]; s[e] = sny[q * 4 + e]; }
        pair_predict(g, s, A);
        stage(H, A);
    }
    bool done = !valid;
    int n_it = 0;
    const double invN = 1.0 / (double)N, tol2 = a.tol * a.tol;
    for (int it = 0; it < a.max_iter; ++it) {
        __syncthreads();                      // z1, z2 staged
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            cd* zf = h ? b2 : b1;
            cd v[P];
#pragma unroll
            for (int t = 0; t < P; ++t) {
                const cd x = zf[i + t * M];
                v[t] = make_double2(x.x, -x.y);
            }
            wpm_fft<P, M>(v, zf, lo, hi, i);
#pragma unroll
            for (int t = 0; t < P; ++t) {
                if (t >= P / 2) { v[t] = make_double2(0.0, 0.0); continue; }
                double sr = invN, si = invN;
                if (t == 0 && i == 0) { sr *= 0.5; si = h ? 0.0 : 0.5 * si; }
                v[t] = make_double2(v[t].x * sr, -v[t].y * si);
            }
            wpm_fft<P, M>(v, zf, lo, hi, i);
            __syncthreads();                  // the gather has read the rows
#pragma unroll
            for (int t = 0; t < P; ++t) zf[i + t * M] = v[t];
        }
        __syncthreads();
        double e2 = 0.0;
        auto update = [&](int f, cd (&g)[4], const double (&s)[4]) {
            const int fm = f == 0 ? 0 : N - f;
            const cd p1 = b1[f], m1 = b1[fm], p2 = b2[f], m2 = b2[fm];
            cd Ap[4];
            Ap[0] = make_double2(0.5 * (p1.x + m1.x), 0.5 * (p1.y - m1.y));
            Ap[3] = make_double2(0.5 * (p1.y + m1.y), 0.5 * (m1.x - p1.x));
            Ap[1] = make_double2(0.5 * (p2.x + m2.x), 0.5 * (p2.y - m2.y));
            Ap[2] = make_double2(0.5 * (p2.y + m2.y), 0.5 * (m2.x - p2.x));
            if (!done) {
                cd n[4];
                n[0] = pz_add(pz_mul(g[0], Ap[0]), pz_mul(g[1], Ap[2])); n[1] = pz_add(pz_mul(g[0], Ap[1]), pz_mul(g[1], Ap[3]));
                n[2] = pz_add(pz_mul(g[2], Ap[0]), pz_mul(g[3], Ap[2])); n[3] = pz_add(pz_mul(g[2], Ap[1]), pz_mul(g[3], Ap[3]));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const cd d = pz_sub(n[k], g[k]);
                    e2 = fmax(e2, d.x * d.x + d.y * d.y);
                    g[k] = n[k];
                }
            }
            cd A[3];
            pair_predict(g, s, A);
            stage(f, A);
        };
        // (update reads b[f] and b[N - f] and re-stages both: the bins of a thread are its own, the staging of bin f touches f and N - f only)
#pragma unroll
        for (int u = 0; u < WP_BINS; ++u) update(j + M * u, G[u], S[u]);
        if (j == 0) {
            cd g[4];
            double s[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { g[k] = gny[q * 4 + k]; s[k] = sny[q * 4 + k]; }
            update(H, g, s);
#pragma unroll
            for (int k = 0; k < 4; ++k) gny[q * 4 + k] = g[k];
        }
        if (!done && e2 > 0.0) atomicMax(errs + q, (unsigned long long)__double_as_longlong(e2));
        __syncthreads();
        if (!done) {
            ++n_it;
            if (__longlong_as_double((long long)errs[q]) < tol2) done = true;
        }
        const int running = __syncthreads_or(done ? 0 : 1);
        if (j == 0) errs[q] = 0ull;
        if (!running) break;
    }
    __syncthreads();
    double hs[4] = {0, 0, 0, 0};
    if (valid) {
        cd* Gp = a.Ghalf + p * 4 * F;
#pragma unroll
        for (int u = 0; u < WP_BINS; ++u) {
            const int f = j + M * u;
            const double w = f == 0 ? 1.0 : 2.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { Gp[k * F + f] = G[u][k]; hs[k] += w * G[u][k].x; }
        }
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const cd g = gny[q * 4 + k]; Gp[k * F + H] = g; hs[k] += g.x; }
        }
    }
    double* red = reinterpret_cast<double*>(z);                           // [WP_THREADS][4]
#pragma unroll
    for (int k = 0; k < 4; ++k) red[tid * 4 + k] = hs[k];
    __syncthreads();
    if (valid && j < 4) {
        double t = 0.0;
        for (int m = 0; m < M; ++m) t += red[(q * M + m) * 4 + j];            // fixed order
        a.h0[p * 4 + j] = t / (double)N;
    }
    if (valid && j == 0) {
        a.n_iter[p] = n_it;
        a.status[p] = done ? 1 : 0;
        atomicMax(a.summary, n_it);
        if (!done) atomicAdd(a.summary + 1, 1);
    }
}
template <int P, int M, int TB>
static int pair_launch_mixed(const PairArgs& a, hipStream_t st) {
    constexpr int N = P * M, PPW = TB / M, ZS = (P * wpm_row(M) > N ? P * wpm_row(M) : N), NHI = (N + 63) / 64;
    const size_t lds = ((size_t)PPW * 2 * ZS + 64 + NHI + PPW * 4) * sizeof(cd) + (size_t)PPW * 4 * 8 + (size_t)PPW * 8;
    static_assert(((size_t)PPW * 2 * ZS + 64 + NHI + PPW * 4) * sizeof(cd) + (size_t)PPW * 48 <= 160 * 1024, "the problems of a workgroup must fit LDS");
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)wilson_pair_mixed_kernel<P, M, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (a.P + PPW - 1) / PPW;
    SC_REQUIRE(blocks <= 0x7fffffffLL, "too many problems for one launch");
    hipLaunchKernelGGL((wilson_pair_mixed_kernel<P, M, TB>), dim3((unsigned)blocks), dim3(PPW * M), lds, st, a);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
// the window lengths of the mixed form: X(N, P) with M = N / P
#define WPM_LENGTHS(X) X(4000, 8, 512) X(3200, 8, 512) X(2400, 8, 512) X(2000, 8, 512) X(1600, 8, 512) X(1200, 8, 512) X(800, 8, 512) \
                       X(1000, 8, 256) X(600, 8, 256) X(400, 8, 256) X(200, 8, 256) X(500, 4, 256) X(300, 4, 256) X(250, 2, 256)
static bool pair_mixed_has(int64_t N) {
#define WPM_HAS(NN, PP, TT) if (N == NN) return true;
    WPM_LENGTHS(WPM_HAS)
#undef WPM_HAS
    return false;
}

// lam = 1e-12 * mean over (windows, entries) of H0^2 per pair, Hinv = (H0 + lam I)^-1, rot from Sigma = H0 H0^T
// (connectivity.py:1739-1742, :1847-1848; the arithmetic of sc_wilson.hip's k_pair_consts)
__global__ void pair_consts_kernel(const double* h0, double* hinv, double* rot, int64_t n_groups, int64_t n_pairs) {
    const int64_t pr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= n_pairs) return;
    double m = 0.0;
    int64_t n_ok = 0;
    for (int64_t g = 0; g < n_groups; ++g) {
        const double* hh = h0 + (g * n_pairs + pr) * 4;
        const double q = hh[0] * hh[0] + hh[1] * hh[1] + hh[2] * hh[2] + hh[3] * hh[3];
        if (isfinite(q)) { m += q; ++n_ok; }
    }
    const double lam = n_ok ? 1e-12 * m / (double)(4 * n_ok) : 0.0;
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t p = g * n_pairs + pr;
        const double a = h0[p * 4], b = h0[p * 4 + 1], c = h0[p * 4 + 2], d = h0[p * 4 + 3];
        const double ra = a + lam, rd = d + lam, det = ra * rd - b * c;
        hinv[p * 4] = rd / det; hinv[p * 4 + 1] = -b / det; hinv[p * 4 + 2] = -c / det; hinv[p * 4 + 3] = ra / det;
        const double s00 = a * a + b * b, s01 = a * c + b * d, s11 = c * c + d * d;
        rot[p * 4] = s00 - s00 * s00 / s00; rot[p * 4 + 1] = s11 - s01 * s01 / s00;
        rot[p * 4 + 2] = s00 - s01 * s01 / s11; rot[p * 4 + 3] = s11 - s11 * s11 / s11;
    }
}

__global__ void pair_fill_nan_kernel(double* out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = nan("");
}

// GP = log P - log(P - rot |H|^2), H = G Hinv (connectivity.py:1679-1779, :1825-1848) on the non-negative bins
__global__ void pair_granger_kernel(PairArgs a, const double* hinv, const double* rot, int64_t F, double* out) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = (int64_t)blockIdx.z * 65535 + blockIdx.y;
    if (f >= F || p >= a.P) return;
    const int64_t g = p / a.n_pairs, pr = p % a.n_pairs;
    const int ci = a.pairs[2 * pr], cj = a.pairs[2 * pr + 1];
    const int idx[2] = {ci, cj};
    double* o = out + ((g * F + f) * a.C) * a.C;
    const cd* Gp = a.Ghalf + p * 4 * F;
    const cd gg[4] = {Gp[f], Gp[F + f], Gp[2 * F + f], Gp[3 * F + f]};
    const double* hi = hinv + p * 4;
    cd Hm[4];
    Hm[0] = make_double2(gg[0].x * hi[0] + gg[1].x * hi[2], gg[0].y * hi[0] + gg[1].y * hi[2]);
    Hm[1] = make_double2(gg[0].x * hi[1] + gg[1].x * hi[3], gg[0].y * hi[1] + gg[1].y * hi[3]);
    Hm[2] = make_double2(gg[2].x * hi[0] + gg[3].x * hi[2], gg[2].y * hi[0] + gg[3].y * hi[2]);
    Hm[3] = make_double2(gg[2].x * hi[1] + gg[3].x * hi[3], gg[2].y * hi[1] + gg[3].y * hi[3]);
    double s[4];
    pair_read_S(a, g, ci, cj, F, (int)f, s);
    const double tp[2] = {s[0], s[1]};               // total power of the two channels
    for (int x = 0; x < 2; ++x)
        for (int y = 0; y < 2; ++y) {
            if (x == y) continue;                    // diagonal is NaN (connectivity.py:2337-2339)
            const cd hh = Hm[x * 2 + y];
            double intrinsic = tp[x] - rot[p * 4 + x * 2 + y] * (hh.x * hh.x + hh.y * hh.y);
            if (intrinsic == 0.0) intrinsic = 2.220446049250313e-16;
            double gp = log(tp[x]) - log(intrinsic);
            if (!(gp > 0.0)) gp = nan("");
            o[(int64_t)idx[x] * a.C + idx[y]] = gp;
        }
}

template <int LOG2N>
static int pair_launch(const PairArgs& a, hipStream_t st) {
    constexpr int WP_HALVES = LOG2N <= 8 ? 2 : 1;
    constexpr int N = 1 << LOG2N, TPF = N / 16, PPW = 256 / TPF, ZS = N + N / 16, NHI = N / 64;
    const size_t lds = ((size_t)PPW * 2 * ZS + 64 + NHI + PPW * 4) * sizeof(cd) + (size_t)PPW * 4 * 8 + (size_t)PPW * 8;
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)wilson_pair_kernel<LOG2N, WP_HALVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (a.P + PPW - 1) / PPW;
    SC_REQUIRE(blocks <= 0x7fffffffLL, "too many problems for one launch");
    hipLaunchKernelGGL((wilson_pair_kernel<LOG2N, WP_HALVES>), dim3((unsigned)blocks), dim3(256 * WP_HALVES), lds, st, a);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

// bytes of workspace the resident form needs (always within sc_granger_workspace_bytes of the same request)
static size_t pair_workspace_bytes(int64_t P, int64_t n_pairs, int64_t N) {
    return (size_t)P * (size_t)(N / 2 + 1) * 4 * 16 + (size_t)P * 4 * 8 * 4 + (size_t)n_pairs * 4 + 256;
}

bool sc_internal_granger_resident_applies(int64_t n_freq_accum, int64_t N) {
    const char* e = sc_switch(SC_SW_GRANGER_KERNEL);
    if (e && strcmp(e, "batched") == 0) return false;
    return n_freq_accum == N / 2 + 1 && (N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096 || pair_mixed_has(N));
}

// The resident form of sc_granger_pairwise_f64 (same arguments; called from there when it applies).
int sc_internal_granger_resident(const void* d_accum, int64_t n_groups, int64_t N, int64_t C, uint32_t planes, int64_t n_obs,
                                 const int32_t* d_pairs, int64_t n_pairs, double tol, int max_iter, void* d_work, size_t work_bytes,
                                 int keep_output, double* d_out, int32_t* d_n_iter, int32_t* d_status, int32_t* h_summary,
                                 hipStream_t st) {
    const int64_t P = n_groups * n_pairs, F = N / 2 + 1;
    SC_REQUIRE(work_bytes >= pair_workspace_bytes(P, n_pairs, N), "workspace too small");
    PairArgs a;
    a.accum = sc_rec(d_accum, planes);
    a.pairs = d_pairs;
    a.P = P; a.n_pairs = n_pairs; a.n_batch = n_pairs;
    a.C = (int)C; a.NB = sc_n_blocks(C); a.n_tiles = sc_n_tiles(a.NB);
    a.p_csm = sc_plane_offset(planes, SC_PLANE_CSM);
    a.floats_per_bin = (int64_t)sc_plane_count(planes) * a.n_tiles * SC_TILE_ELEMS;
    a.n_obs = (double)n_obs;
    char* w = (char*)d_work;
    a.Ghalf = (cd*)w; w += (size_t)P * F * 4 * 16;
    a.chol = (double*)w; w += (size_t)P * 32;
    a.h0 = (double*)w; w += (size_t)P * 32;
    double* hinv = (double*)w; w += (size_t)P * 32;
    double* rot = (double*)w; w += (size_t)P * 32;
    a.batch_bad = (int32_t*)w; w += (size_t)n_pairs * 4;
    a.summary = (int32_t*)w;
    a.n_iter = d_n_iter; a.status = d_status;
    a.tol = tol; a.max_iter = max_iter;
    SC_CHECK_HIP(hipMemsetAsync(a.batch_bad, 0, (size_t)n_pairs * 4 + 256, st));          // the flags and the summary behind them
    if (!keep_output)
        hipLaunchKernelGGL(pair_fill_nan_kernel, dim3((unsigned)((n_groups * F * C * C + 255) / 256)), dim3(256), 0, st, d_out,
                           n_groups * F * C * C);
    hipLaunchKernelGGL(pair_lag0_kernel, dim3((unsigned)P), dim3(256), 0, st, a, N);
    hipLaunchKernelGGL(pair_restart_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, a);
    int rc;
    const char* gk = sc_switch(SC_SW_GRANGER_KERNEL);
    if (gk && strcmp(gk, "p16") == 0 && (N == 4000 || N == 2000)) {        // (A/B: sixteen points per thread, one wave per SIMD)
        rc = N == 4000 ? pair_launch_mixed<16, 250, 256>(a, st) : pair_launch_mixed<16, 125, 256>(a, st);
    } else
    switch (N) {
#define WPM_CASE(NN, PP, TT) case NN: rc = pair_launch_mixed<PP, NN / PP, TT>(a, st); break;
    WPM_LENGTHS(WPM_CASE)
#undef WPM_CASE
    case 256: rc = pair_launch<8>(a, st); break;
    case 512: rc = pair_launch<9>(a, st); break;
    case 1024: rc = pair_launch<10>(a, st); break;
    case 2048: rc = pair_launch<11>(a, st); break;
    default: rc = pair_launch<12>(a, st); break;
    }
    if (rc != SC_OK) return rc;
    hipLaunchKernelGGL(pair_consts_kernel, dim3((unsigned)((n_pairs + 63) / 64)), dim3(64), 0, st, a.h0, hinv, rot, n_groups, n_pairs);
    const dim3 gridF((unsigned)((F + 255) / 256), (unsigned)(P < 65535 ? P : 65535), (unsigned)((P + 65534) / 65535));
    hipLaunchKernelGGL(pair_granger_kernel, gridF, dim3(256), 0, st, a, hinv, rot, F, d_out);
    int32_t sum[3] = {0, 0, 0};
    if (hipMemcpyAsync(sum, a.summary, sizeof sum, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
        hipGetLastError() != hipSuccess) {
        sc_set_error("pairwise Granger (resident form): %s", hipGetErrorString(hipGetLastError()));
        return SC_EHIP;
    }
    if (h_summary) { h_summary[0] = sum[0]; h_summary[1] = sum[1]; h_summary[2] = sum[2]; }
    return SC_OK;
}
