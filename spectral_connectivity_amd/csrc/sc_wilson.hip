// sc_wilson.hip -- batched 2x2 Wilson spectral factorisation + pairwise spectral Granger.
//
// The reference loops over channel pairs in Python and runs Wilson's algorithm on one
// (W, N, 2, 2) problem at a time (connectivity.py:2282-2340, minimum_phase_decomposition.py:
// 227-322; ~0.16 s per pair measured).  Here every (group, pair) problem is a row of one batch:
//   k_build    two-sided 2x2 cross-spectral matrices from the accumulator records (fp64)
//   k_init     G0 = chol(Re ifft_n(S)[lag 0])^H                 (minimum_phase...py:48-77)
//   loop <= max_iter, all problems at once:
//     k_predict  A = G^-1 (G^-1 S)^H + I, closed-form 2x2       (:184-224)
//     N = 256..4096: causal_fft_pair_kernel (sc_wilson_fft.hip)  A+ = fft(mask(ifft(A))) in one pass;
//       otherwise rocFFT ifft, k_causal (a[0] *= 1/2, strict lower of a[0] = 0, a[n >= (N+1)/2] = 0; :96-142), rocFFT fft
//     k_update   G <- G A+ unless the problem already converged; err = max |G - G_old| (:145-181, :301-315);
//                the next iteration's predict rides in the same pass
//   k_h0 / k_granger   H0 = Re ifft_n(G)[0]; H = G (H0 + lam I)^-1; Sigma = H0 H0^T;
//                      GP = log P - log(P - rot |H|^2)          (connectivity.py:1679-1779, :1825-1848)
// Everything is fp64: the reference's convergence test (max |dG| < 1e-8 absolute) is not
// reachable in fp32.  Each (group, pair) problem stops at its own convergence; the reference
// freezes a window once converged, which yields the same iterate.
// Layouts: S [P][4][N] doubles (s00, s11, Re s01, Im s01); G, A [P][4][N] complex128
// (entry e = 2*row + col), n fastest so the FFTs are unit-stride and pointwise kernels coalesce.
#include <rocfft/rocfft.h>
#include "sc_common.h"

typedef double2 cd;
__device__ inline cd cmul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline cd cconj(cd a) { return make_double2(a.x, -a.y); }
__device__ inline cd cadd(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline cd csub(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ inline cd cdivi(cd a, cd b) {
    const double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

// (group, pair) problem of a block: gridDim.y holds at most 65535 problems, gridDim.z the rest
#define WILSON_PMAX 65535
__device__ __forceinline__ int64_t wilson_problem() { return (int64_t)blockIdx.z * WILSON_PMAX + blockIdx.y; }
static inline dim3 wilson_grid(unsigned gx, int64_t P) {
    return dim3(gx, (unsigned)(P < WILSON_PMAX ? P : WILSON_PMAX), (unsigned)((P + WILSON_PMAX - 1) / WILSON_PMAX));
}

struct WilsonDims {
    int64_t P;       // problems = n_groups * n_pairs
    int64_t N;       // two-sided FFT length
    int64_t n_pairs;
    int64_t F;       // accumulated bins per group
    int C, NB, n_tiles;
    int p_csm;
    int two_sided;   // accumulators hold all N bins (uploaded coefficients) instead of N/2+1
    int64_t floats_per_bin;
    double n_obs;
};

__device__ inline double acc_read(ScRec rec, int plane, int n_tiles, int NB, int i, int j, bool* mirrored) {
    int ti = i >> 4, tj = j >> 4, ii = i & 15, jj = j & 15;
    const bool m = (ti > tj) || (ti == tj && ii > jj);
    if (m) { int t = ti; ti = tj; tj = t; t = ii; ii = jj; jj = t; }
    *mirrored = m;
    return rec[((int64_t)plane * n_tiles + sc_tile_index(ti, tj, NB)) * SC_TILE_ELEMS + ii * 16 + jj];
}

__global__ void k_build(ScRec accum, const int32_t* pairs, WilsonDims d, double* S) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = wilson_problem();
    if (n >= d.N || p >= d.P) return;
    const int64_t g = p / d.n_pairs, pr = p % d.n_pairs;
    const int i = pairs[2 * pr], j = pairs[2 * pr + 1];
    int64_t bin = n;
    bool conj = false;
    if (!d.two_sided && n > d.N / 2) { bin = d.N - n; conj = true; }   // real input: S(-f) = conj S(f)
    const ScRec rec = accum + (g * d.F + bin) * d.floats_per_bin;
    bool m, mm;
    const double s00 = (double)acc_read(rec, d.p_csm, d.n_tiles, d.NB, i, i, &mm) / d.n_obs;
    const double s11 = (double)acc_read(rec, d.p_csm, d.n_tiles, d.NB, j, j, &mm) / d.n_obs;
    const double re = (double)acc_read(rec, d.p_csm, d.n_tiles, d.NB, i, j, &m) / d.n_obs;
    double im = (double)acc_read(rec, d.p_csm + 1, d.n_tiles, d.NB, i, j, &m) / d.n_obs;
    if (m) im = -im;
    if (conj) im = -im;
    double* Sp = S + p * 4 * d.N;
    Sp[n] = s00; Sp[d.N + n] = s11; Sp[2 * d.N + n] = re; Sp[3 * d.N + n] = im;
}

// one block per problem: lag-0 covariance = mean_n Re S[n]; G0 = chol(R0)^H broadcast over n.
// Where the covariance is not positive definite the reference (minimum_phase_decomposition.py:78-93) logs a
// warning and starts from the Cholesky factor of a random Wishart matrix -- mean of 1000 products Z Z^T of standard
// normal matrices, i.e. c I plus O(3 %) noise drawn from the GLOBAL NumPy generator -- and it does so for the WHOLE
// batch its batched Cholesky was called on: every window of that channel pair, the healthy ones included.  That
// matters: at a finite FFT length the fixed point the iteration reaches depends on the start (measured on
// tests/golden/f12: 2e-3 of the prediction between the Cholesky start and the restart, 2e-5 between restarts of any
// scale or seed).  So a failing problem flags its batch (problems p with the same p % n_batch: the windows of one
// pair; n_batch = 1: everything handed to sc_wilson_factor_f64) and k_restart puts the expectation of the reference's
// draw, the identity, into every problem of a flagged batch; *n_fallback counts the problems restarted.
__global__ void __launch_bounds__(256) k_init(const double* S, cd* G, int32_t* status, int32_t* batch_bad, int64_t n_batch,
                                              int64_t N) {
    __shared__ double red[3][256];
    const int64_t p = blockIdx.x;
    const double* Sp = S + p * 4 * N;
    double a = 0, b = 0, c = 0;
    for (int64_t n = threadIdx.x; n < N; n += 256) { a += Sp[n]; b += Sp[N + n]; c += Sp[2 * N + n]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
        __syncthreads();
    }
    const double r00 = red[0][0] / (double)N, r11 = red[1][0] / (double)N, r01 = red[2][0] / (double)N;
    // lower Cholesky L of [[r00, r01],[r01, r11]]; G0 = L^T (upper triangular, real)
    double l00 = sqrt(r00), l10 = r01 / l00;
    const double t = r11 - l10 * l10;
    double l11 = sqrt(t);
    const bool bad = !(r00 > 0.0) || !(t > 0.0);
    if (bad) { l00 = 1.0; l10 = 0.0; l11 = 1.0; }
    if (threadIdx.x == 0) {
        status[p] = 0;
        if (bad) atomicOr(batch_bad + p % n_batch, 1);
    }
    cd* Gp = G + p * 4 * N;
    for (int64_t n = threadIdx.x; n < N; n += 256) {
        Gp[n] = make_double2(l00, 0); Gp[N + n] = make_double2(l10, 0);
        Gp[2 * N + n] = make_double2(0, 0); Gp[3 * N + n] = make_double2(l11, 0);
    }
}

__global__ void __launch_bounds__(256) k_restart(cd* G, const int32_t* batch_bad, int64_t n_batch, int32_t* n_fallback, int64_t N) {
    const int64_t p = blockIdx.x;
    if (!batch_bad[p % n_batch]) return;
    if (threadIdx.x == 0) atomicAdd(n_fallback, 1);
    cd* Gp = G + p * 4 * N;
    for (int64_t n = threadIdx.x; n < N; n += 256) {
        Gp[n] = make_double2(1.0, 0); Gp[N + n] = make_double2(0, 0);
        Gp[2 * N + n] = make_double2(0, 0); Gp[3 * N + n] = make_double2(1.0, 0);
    }
}

// A = G^-1 (G^-1 S)^H + I at one frequency, closed-form 2x2
__device__ __forceinline__ void predict2x2(cd g00, cd g01, cd g10, cd g11, const double* Sp, cd* Ap, int64_t n, int64_t N) {
    const cd s00 = make_double2(Sp[n], 0), s11 = make_double2(Sp[N + n], 0);
    const cd s01 = make_double2(Sp[2 * N + n], Sp[3 * N + n]), s10 = cconj(s01);
    const cd det = csub(cmul(g00, g11), cmul(g01, g10));
    // Ginv = 1/det [[g11, -g01], [-g10, g00]]
    const cd i00 = cdivi(g11, det), i01 = cdivi(make_double2(-g01.x, -g01.y), det);
    const cd i10 = cdivi(make_double2(-g10.x, -g10.y), det), i11 = cdivi(g00, det);
    // X = Ginv S
    const cd x00 = cadd(cmul(i00, s00), cmul(i01, s10)), x01 = cadd(cmul(i00, s01), cmul(i01, s11));
    const cd x10 = cadd(cmul(i10, s00), cmul(i11, s10)), x11 = cadd(cmul(i10, s01), cmul(i11, s11));
    // Y = Ginv X^H ; A = Y + I
    const cd h00 = cconj(x00), h01 = cconj(x10), h10 = cconj(x01), h11 = cconj(x11);
    cd a00 = cadd(cmul(i00, h00), cmul(i01, h10)); a00.x += 1.0;
    cd a11 = cadd(cmul(i10, h01), cmul(i11, h11)); a11.x += 1.0;
    Ap[n] = a00;
    Ap[N + n] = cadd(cmul(i00, h01), cmul(i01, h11));
    Ap[2 * N + n] = cadd(cmul(i10, h00), cmul(i11, h10));
    Ap[3 * N + n] = a11;
}

__global__ void k_predict(const double* S, const cd* G, const int32_t* status, cd* A, int64_t N, int64_t P) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = wilson_problem();
    if (n >= N || p >= P || status[p] != 0) return;
    const cd* Gp = G + p * 4 * N;
    predict2x2(Gp[n], Gp[N + n], Gp[2 * N + n], Gp[3 * N + n], S + p * 4 * N, A + p * 4 * N, n, N);
}

// after the (unnormalised) inverse FFT: 1/N, halve lag 0, zero strict lower triangle at lag 0,
// zero the non-causal half
__global__ void k_causal(cd* A, int64_t N, int64_t P) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = wilson_problem();
    if (n >= N || p >= P) return;
    cd* Ap = A + p * 4 * N;
    const double invN = 1.0 / (double)N;
    const bool keep = n < (N + 1) / 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cd v = Ap[e * N + n];
        double sc = keep ? invN : 0.0;
        if (n == 0) { sc *= 0.5; if (e == 2) sc = 0.0; }
        Ap[e * N + n] = make_double2(v.x * sc, v.y * sc);
    }
}

__device__ inline void atomic_max_nonneg(double* addr, double v) {
    // order of non-negative doubles == order of their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

// G <- G A+ and err = max |G - G_old|; the next iteration's A = predict(G_new) overwrites A+ in the same pass
// (a problem that turns out to have converged leaves an A nobody reads).
__global__ void __launch_bounds__(256) k_update(cd* G, cd* Aplus, const double* S, const int32_t* status, double* err,
                                                int64_t N, int64_t P) {
    __shared__ double red[256];
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p = wilson_problem();
    if (p >= P) return;                                // the whole block shares p
    double e = 0.0;
    if (n < N && status[p] == 0) {
        cd* Gp = G + p * 4 * N;
        cd* Ap = Aplus + p * 4 * N;
        const cd g00 = Gp[n], g01 = Gp[N + n], g10 = Gp[2 * N + n], g11 = Gp[3 * N + n];
        const cd a00 = Ap[n], a01 = Ap[N + n], a10 = Ap[2 * N + n], a11 = Ap[3 * N + n];
        const cd n00 = cadd(cmul(g00, a00), cmul(g01, a10)), n01 = cadd(cmul(g00, a01), cmul(g01, a11));
        const cd n10 = cadd(cmul(g10, a00), cmul(g11, a10)), n11 = cadd(cmul(g10, a01), cmul(g11, a11));
        cd d;
        d = csub(n00, g00); e = fmax(e, hypot(d.x, d.y));
        d = csub(n01, g01); e = fmax(e, hypot(d.x, d.y));
        d = csub(n10, g10); e = fmax(e, hypot(d.x, d.y));
        d = csub(n11, g11); e = fmax(e, hypot(d.x, d.y));
        Gp[n] = n00; Gp[N + n] = n01; Gp[2 * N + n] = n10; Gp[3 * N + n] = n11;
        predict2x2(n00, n01, n10, n11, S + p * 4 * N, Ap, n, N);
    }
    red[threadIdx.x] = e;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] > 0.0) atomic_max_nonneg(err + p, red[0]);
}

// status: 0 running -> 1 converged (err < tol); counts iterations; clears err; *n_running = #still 0 (one slot per
// iteration: the host reads the slots of a whole batch of iterations at once)
__global__ void k_flags(int32_t* status, int32_t* n_iter, double* err, double tol, int64_t P, int32_t* n_running) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (status[p] == 0) {
        n_iter[p] += 1;
        if (err[p] < tol) status[p] = 1;
        else atomicAdd(n_running, 1);
    }
    err[p] = 0.0;
}

// H0 = Re ifft_n(G)[lag 0] = mean_n Re G[n]  -> h0[p][4]
__global__ void __launch_bounds__(256) k_h0(const cd* G, double* h0, int64_t N) {
    __shared__ double red[4][256];
    const int64_t p = blockIdx.x;
    const cd* Gp = G + p * 4 * N;
    double s[4] = {0, 0, 0, 0};
    for (int64_t n = threadIdx.x; n < N; n += 256)
        for (int e = 0; e < 4; ++e) s[e] += Gp[e * N + n].x;
    for (int e = 0; e < 4; ++e) red[e][threadIdx.x] = s[e];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st)
            for (int e = 0; e < 4; ++e) red[e][threadIdx.x] += red[e][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 4) h0[p * 4 + threadIdx.x] = red[threadIdx.x][0] / (double)N;
}

// per pair: lam = 1e-12 * mean over (groups, entries) of H0^2 (connectivity.py:1739-1742 takes the
// mean over the whole (W,1,2,2) array); Hinv = (H0 + lam I)^-1; rot from Sigma = H0 H0^T
__global__ void k_pair_consts(const double* h0, double* hinv, double* rot, int64_t n_groups, int64_t n_pairs) {
    const int64_t pr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= n_pairs) return;
    // groups whose factor came out non-finite (a degenerate window) are left out of the mean, so that only they
    // are NaN in the output
    double m = 0.0;
    int64_t n_ok = 0;
    for (int64_t g = 0; g < n_groups; ++g) {
        const double* h = h0 + (g * n_pairs + pr) * 4;
        const double q = h[0] * h[0] + h[1] * h[1] + h[2] * h[2] + h[3] * h[3];
        if (isfinite(q)) { m += q; ++n_ok; }
    }
    const double lam = n_ok ? 1e-12 * m / (double)(4 * n_ok) : 0.0;
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t p = g * n_pairs + pr;
        const double a = h0[p * 4], b = h0[p * 4 + 1], c = h0[p * 4 + 2], d = h0[p * 4 + 3];
        const double ra = a + lam, rd = d + lam, det = ra * rd - b * c;
        hinv[p * 4] = rd / det; hinv[p * 4 + 1] = -b / det; hinv[p * 4 + 2] = -c / det; hinv[p * 4 + 3] = ra / det;
        // Sigma = H0 H0^T
        const double s00 = a * a + b * b, s01 = a * c + b * d, s11 = c * c + d * d;
        // rot[x][y] = var[y] - Sigma[x][y]^2 / var[x]   (connectivity.py:1847-1848)
        rot[p * 4] = s00 - s00 * s00 / s00; rot[p * 4 + 1] = s11 - s01 * s01 / s00;
        rot[p * 4 + 2] = s00 - s01 * s01 / s11; rot[p * 4 + 3] = s11 - s11 * s11 / s11;
    }
}

__global__ void k_fill_nan(double* out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = nan("");
}

__global__ void k_granger(const cd* G, const double* S, const double* hinv, const double* rot,
                          const int32_t* status, const int32_t* pairs, WilsonDims d, int64_t Fout, double* out) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = wilson_problem();
    if (f >= Fout || p >= d.P) return;
    const int64_t g = p / d.n_pairs, pr = p % d.n_pairs;
    const int i = pairs[2 * pr], j = pairs[2 * pr + 1];
    const int idx[2] = {i, j};
    const int64_t N = d.N;
    double* o = out + ((g * Fout + f) * d.C) * d.C;
    const cd* Gp = G + p * 4 * N;
    const double* Sp = S + p * 4 * N;
    const cd gg[4] = {Gp[f], Gp[N + f], Gp[2 * N + f], Gp[3 * N + f]};
    const double* hi = hinv + p * 4;
    // H = G Hinv (Hinv real)
    cd H[4];
    H[0] = make_double2(gg[0].x * hi[0] + gg[1].x * hi[2], gg[0].y * hi[0] + gg[1].y * hi[2]);
    H[1] = make_double2(gg[0].x * hi[1] + gg[1].x * hi[3], gg[0].y * hi[1] + gg[1].y * hi[3]);
    H[2] = make_double2(gg[2].x * hi[0] + gg[3].x * hi[2], gg[2].y * hi[0] + gg[3].y * hi[2]);
    H[3] = make_double2(gg[2].x * hi[1] + gg[3].x * hi[3], gg[2].y * hi[1] + gg[3].y * hi[3]);
    const double tp[2] = {Sp[f], Sp[N + f]};         // total power of the two channels
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            if (a == b) continue;                    // diagonal is NaN (connectivity.py:2337-2339)
            const cd h = H[a * 2 + b];
            double intrinsic = tp[a] - rot[p * 4 + a * 2 + b] * (h.x * h.x + h.y * h.y);
            if (intrinsic == 0.0) intrinsic = 2.220446049250313e-16;
            double gp = log(tp[a]) - log(intrinsic);
            if (!(gp > 0.0)) gp = nan("");
            o[(int64_t)idx[a] * d.C + idx[b]] = gp;
        }
}

// ------------------------------------------------------------------------------- host side
struct WilsonPlan {
    rocfft_plan fwd, inv;
    rocfft_execution_info info_f, info_i;
    void* work;
    size_t work_bytes;
};

#define WILSON_HIST 1024     // iterations whose "still running" counts the workspace can log (max_iterations <= this)
#define WILSON_POLL 4        // iterations queued between two looks at the counts
extern "C" int sc_granger_workspace_bytes(int64_t n_groups, int64_t n_pairs, int64_t N, size_t* bytes) {
    SC_REQUIRE(bytes && n_groups >= 1 && n_pairs >= 1 && N >= 2, "bad workspace query");
    const size_t P = (size_t)n_groups * n_pairs;
    // S (4 doubles) + G (4 complex) + A (4 complex) per (problem, bin) + per-problem scalars
    *bytes = P * (size_t)N * (4 * 8 + 4 * 16 + 4 * 16) + P * (8 + 4 + 4 + 4 * 8 * 3) + 256 + WILSON_HIST * 4;
    return SC_OK;
}

#define SC_CHECK_FFT2(expr)                                                                      \
    do {                                                                                         \
        rocfft_status s_ = (expr);                                                               \
        if (s_ != rocfft_status_success) {                                                       \
            sc_set_error("%s failed: rocfft_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
            rc = SC_EFFT; goto done;                                                             \
        }                                                                                        \
    } while (0)

struct WilsonWork {
    double* S; cd* G; cd* A; double* err; double* h0; double* hinv; double* rot;
    int32_t* n_fallback;     // problems started from the identity (lag-0 covariance not positive definite)
    int32_t* n_running;      // [WILSON_HIST] problems still running after iteration i
};

static WilsonWork wilson_carve(void* d_work, int64_t P, int64_t N) {
    WilsonWork k;
    char* w = (char*)d_work;
    k.S = (double*)w; w += (size_t)P * N * 4 * 8;
    k.G = (cd*)w; w += (size_t)P * N * 4 * 16;
    k.A = (cd*)w; w += (size_t)P * N * 4 * 16;
    k.err = (double*)w; w += (size_t)P * 8;
    k.h0 = (double*)w; w += (size_t)P * 32;
    k.hinv = (double*)w; w += (size_t)P * 32;
    k.rot = (double*)w; w += (size_t)P * 32;
    k.n_fallback = (int32_t*)w; w += 256;
    k.n_running = (int32_t*)w;
    return k;
}

// k_init + the Wilson iteration on work.S -> work.G.  The stream is synchronised once per WILSON_POLL iterations:
// every iteration logs how many problems are still running into its own slot, converged problems are skipped by
// every kernel, so queueing a few iterations past the last convergence changes nothing but costs empty launches.
static int wilson_iterate(const WilsonWork& k, int64_t P, int64_t n_batch, int64_t N, double tol, int max_iter,
                          int32_t* d_n_iter, int32_t* d_status, int* iters_out, int* running_out, int* fallback_out,
                          hipStream_t st) {
    int rc = SC_OK;
    rocfft_plan fwd = nullptr, inv = nullptr;
    bool fwd_cached = false, inv_cached = false;
    rocfft_execution_info info = nullptr;
    void* fft_work = nullptr;
    size_t ws_f = 0, ws_i = 0;
    static int rocfft_ready = 0;
    if (!rocfft_ready) { rocfft_setup(); rocfft_ready = 1; }
    const dim3 gridN = wilson_grid((unsigned)((N + 255) / 256), P);
    int iters = 0, running = (int)P, queued = 0;
    int32_t hist[WILSON_POLL];
    const bool fused = sc_internal_causal_fft_supported(N);
    if (max_iter > WILSON_HIST) {
        sc_set_error("max_iterations = %d exceeds the %d iterations the workspace can log", max_iter, WILSON_HIST);
        return SC_EINVAL;
    }

    if (!fused) {
        if ((rc = sc_internal_z2z_plan(&fwd, 1, N, 4 * P, &fwd_cached)) != SC_OK) goto done;
        if ((rc = sc_internal_z2z_plan(&inv, 0, N, 4 * P, &inv_cached)) != SC_OK) goto done;
        SC_CHECK_FFT2(rocfft_plan_get_work_buffer_size(fwd, &ws_f));
        SC_CHECK_FFT2(rocfft_plan_get_work_buffer_size(inv, &ws_i));
        SC_CHECK_FFT2(rocfft_execution_info_create(&info));
        if (ws_f < ws_i) ws_f = ws_i;
        if (ws_f) {
            if (hipMallocAsync(&fft_work, ws_f, st) != hipSuccess) { sc_set_error("rocFFT work buffer alloc failed"); rc = SC_ENOMEM; goto done; }
            SC_CHECK_FFT2(rocfft_execution_info_set_work_buffer(info, fft_work, ws_f));
        }
        SC_CHECK_FFT2(rocfft_execution_info_set_stream(info, st));
    }
    (void)hipMemsetAsync(k.err, 0, (size_t)P * 8, st);
    (void)hipMemsetAsync(d_n_iter, 0, (size_t)P * 4, st);
    (void)hipMemsetAsync(k.n_fallback, 0, 256 + (size_t)WILSON_HIST * 4, st);      // fallback count + the slots
    // (the per-batch flags borrow the start of the error array: P doubles >= n_batch ints, cleared again below)
    hipLaunchKernelGGL(k_init, dim3((unsigned)P), dim3(256), 0, st, k.S, k.G, d_status, (int32_t*)k.err, n_batch, N);
    hipLaunchKernelGGL(k_restart, dim3((unsigned)P), dim3(256), 0, st, k.G, (const int32_t*)k.err, n_batch, k.n_fallback, N);
    (void)hipMemsetAsync(k.err, 0, (size_t)P * 8, st);
    hipLaunchKernelGGL(k_predict, gridN, dim3(256), 0, st, k.S, k.G, d_status, k.A, N, P);
    while (queued < max_iter && running > 0) {
        const int first = queued;
        for (int b = 0; b < WILSON_POLL && queued < max_iter; ++b, ++queued) {
            if (fused) {        // one kernel for ifft -> causal mask -> fft
                if ((rc = sc_internal_causal_fft_pair(k.A, d_status, P, 2, N, st)) != SC_OK) goto done;
            } else {
                void* bufs[1] = {k.A};
                SC_CHECK_FFT2(rocfft_execute(inv, bufs, nullptr, info));
                hipLaunchKernelGGL(k_causal, gridN, dim3(256), 0, st, k.A, N, P);
                SC_CHECK_FFT2(rocfft_execute(fwd, bufs, nullptr, info));
            }
            // G <- G A+ with the next iteration's A = predict(G) in the same pass
            hipLaunchKernelGGL(k_update, gridN, dim3(256), 0, st, k.G, k.A, k.S, d_status, k.err, N, P);
            hipLaunchKernelGGL(k_flags, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, d_status, d_n_iter, k.err, tol, P,
                               k.n_running + queued);
        }
        if (hipMemcpyAsync(hist, k.n_running + first, (size_t)(queued - first) * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            sc_set_error("Wilson iterations %d..%d: %s", first, queued, hipGetErrorString(hipGetLastError()));
            rc = SC_EHIP; goto done;
        }
        for (int b = 0; b < queued - first; ++b) {
            running = hist[b];
            iters = first + b + 1;
            if (running == 0) break;
        }
    }
    if (hipMemcpyAsync(fallback_out, k.n_fallback, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        sc_set_error("Wilson: %s", hipGetErrorString(hipGetLastError()));
        rc = SC_EHIP; goto done;
    }
    *iters_out = iters;
    *running_out = running;
done:
    if (info) rocfft_execution_info_destroy(info);
    if (fwd && !fwd_cached) rocfft_plan_destroy(fwd);       // (cached plans live as long as the process: sc_internal_z2z_plan)
    if (inv && !inv_cached) rocfft_plan_destroy(inv);
    if (fft_work) (void)hipFreeAsync(fft_work, st);
    return rc;
}

extern "C" int sc_granger_pairwise_f64(const void* d_accum, int64_t n_groups, int64_t n_freq_accum,
                                       int64_t N, int64_t C, uint32_t planes, int64_t n_obs,
                                       const int32_t* d_pairs, int64_t n_pairs, double tol, int max_iter,
                                       void* d_work, size_t work_bytes, int flags, double* d_out, int32_t* d_n_iter,
                                       int32_t* d_status, int32_t* h_summary, void* stream) {
    ScTimed timed_("granger_pairwise", stream);
    SC_REQUIRE(d_accum && d_pairs && d_work && d_out && d_n_iter && d_status, "NULL argument");
    SC_REQUIRE(planes & SC_PLANE_CSM, "accumulator record must contain SC_PLANE_CSM");
    SC_REQUIRE(n_freq_accum == N || n_freq_accum == N / 2 + 1, "accumulators must hold N or N/2+1 bins");
    SC_REQUIRE(n_groups >= 1 && n_pairs >= 1 && n_groups * n_pairs <= (int64_t)WILSON_PMAX * WILSON_PMAX, "bad problem count");
    size_t need = 0;
    sc_granger_workspace_bytes(n_groups, n_pairs, N, &need);
    SC_REQUIRE(work_bytes >= need, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (sc_internal_granger_resident_applies(n_freq_accum, N)) {
        // records of real series, N = 256 ... 4096: the whole iteration of a pair on one compute unit (sc_wilson_pair.hip)
        SC_REQUIRE(max_iter >= 1, "max_iterations must be positive");
        return sc_internal_granger_resident(d_accum, n_groups, N, C, planes, n_obs, d_pairs, n_pairs, tol, max_iter, d_work, work_bytes,
                                            (flags & SC_GRANGER_KEEP_OUTPUT) ? 1 : 0, d_out, d_n_iter, d_status, h_summary, st);
    }
    WilsonDims d;
    d.P = n_groups * n_pairs; d.N = N; d.n_pairs = n_pairs; d.F = n_freq_accum; d.C = (int)C;
    d.NB = sc_n_blocks(C); d.n_tiles = sc_n_tiles(d.NB);
    d.p_csm = sc_plane_offset(planes, SC_PLANE_CSM);
    d.two_sided = (n_freq_accum == N && N > 1) ? 1 : 0;
    d.floats_per_bin = (int64_t)sc_plane_count(planes) * d.n_tiles * SC_TILE_ELEMS;
    d.n_obs = (double)n_obs;
    const int64_t P = d.P, Fout = N / 2 + 1;
    const WilsonWork k = wilson_carve(d_work, P, N);
    const dim3 gridN = wilson_grid((unsigned)((N + 255) / 256), P), gridF = wilson_grid((unsigned)((Fout + 255) / 256), P);
    if (!(flags & SC_GRANGER_KEEP_OUTPUT))
        hipLaunchKernelGGL(k_fill_nan, dim3((unsigned)((n_groups * Fout * C * C + 255) / 256)), dim3(256), 0, st, d_out,
                           n_groups * Fout * C * C);
    hipLaunchKernelGGL(k_build, gridN, dim3(256), 0, st, sc_rec(d_accum, planes), d_pairs, d, k.S);
    int iters = 0, running = 0, fallback = 0;
    const int rc = wilson_iterate(k, P, n_pairs, N, tol, max_iter, d_n_iter, d_status, &iters, &running, &fallback, st);
    if (rc != SC_OK) return rc;
    hipLaunchKernelGGL(k_h0, dim3((unsigned)P), dim3(256), 0, st, k.G, k.h0, N);
    hipLaunchKernelGGL(k_pair_consts, dim3((unsigned)((n_pairs + 63) / 64)), dim3(64), 0, st, k.h0, k.hinv, k.rot,
                       n_groups, n_pairs);
    hipLaunchKernelGGL(k_granger, gridF, dim3(256), 0, st, k.G, k.S, k.hinv, k.rot, d_status, d_pairs, d, Fout, d_out);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        sc_set_error("Granger epilogue failed: %s", hipGetErrorString(hipGetLastError()));
        return SC_EHIP;
    }
    if (h_summary) { h_summary[0] = iters; h_summary[1] = running; h_summary[2] = fallback; }
    return SC_OK;
}

// Minimum-phase factor only (minimum_phase_decomposition.py:227-322) for callers that hand in their
// own two-sided 2x2 Hermitian spectra: d_S [P][4][N] doubles (s00, s11, Re s01, Im s01);
// d_G out [P][4][N] complex128 (entries g00, g01, g10, g11), with S = G G^H.
extern "C" int sc_wilson_factor_f64(const double* d_S, int64_t n_problems, int64_t N, double tol, int max_iter,
                                    void* d_work, size_t work_bytes, void* d_G, int32_t* d_n_iter,
                                    int32_t* d_status, int32_t* h_summary, void* stream) {
    ScTimed timed_("wilson_factor", stream);
    SC_REQUIRE(d_S && d_work && d_G && d_n_iter && d_status, "NULL argument");
    SC_REQUIRE(n_problems >= 1 && n_problems <= (int64_t)WILSON_PMAX * WILSON_PMAX && N >= 2, "bad problem size");
    size_t need = 0;
    sc_granger_workspace_bytes(1, n_problems, N, &need);
    SC_REQUIRE(work_bytes >= need, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const WilsonWork k = wilson_carve(d_work, n_problems, N);
    SC_CHECK_HIP(hipMemcpyAsync(k.S, d_S, (size_t)n_problems * N * 4 * 8, hipMemcpyDeviceToDevice, st));
    int iters = 0, running = 0, fallback = 0;
    const int rc = wilson_iterate(k, n_problems, 1, N, tol, max_iter, d_n_iter, d_status, &iters, &running, &fallback, st);
    if (rc != SC_OK) return rc;
    SC_CHECK_HIP(hipMemcpyAsync(d_G, k.G, (size_t)n_problems * N * 4 * 16, hipMemcpyDeviceToDevice, st));
    SC_CHECK_HIP(hipStreamSynchronize(st));
    if (h_summary) { h_summary[0] = iters; h_summary[1] = running; h_summary[2] = fallback; }
    return SC_OK;
}
