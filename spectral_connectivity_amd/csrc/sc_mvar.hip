// sc_mvar.hip -- full C x C Wilson spectral factorisation, batched over time windows, and the
// directed measures of the implied MVAR model.
//
// Reference path: Connectivity._minimum_phase_factor / _transfer_function / _noise_covariance /
// _MVAR_Fourier_coefficients (connectivity.py:567-589) -> minimum_phase_decomposition
// (minimum_phase_decomposition.py:227-322) and directed_transfer_function, directed_coherence,
// partial_directed_coherence, generalized_partial_directed_coherence,
// direct_directed_transfer_function (connectivity.py:1237-1426, helpers :1679-1748, :1873-1950).
//
//   m_build / m_upload   two-sided Hermitian spectra S[p][e][n] (e = i C + j, n fastest)
//   m_init               G0 = chol(Re ifft_n(S)[lag 0])^H, broadcast over n
//   loop <= max_iter (all windows at once, converged windows frozen):
//     m_predict          A = G^-1 (G^-1 S)^H + I          one workgroup per (window, bin): LU with
//                                                          partial pivoting in LDS, two solves
//     rocFFT Z2Z         a = ifft_n(A)                     C^2 unit-stride series per window
//     m_causal           1/N, a[0] *= 1/2, strict lower triangle of a[0] = 0, a[n >= (N+1)/2] = 0
//     rocFFT Z2Z         A+ = fft_n(a)
//     m_update           G <- G A+, err = max |G - G_old|  one workgroup per (window, bin)
//   measures             H0 = Re mean_n G; H = G (H0 + lam I)^-1 on the non-negative bins;
//                        A_mvar = (H + lam' I)^-1; Sigma = H0 H0^T; DTF / DC / PDC / gPDC / dDTF.
// Everything is fp64 (the reference's convergence test max |dG| < 1e-8 is out of fp32's reach).
// One matrix pair lives in LDS: C <= 64 (2 x 64 KB).  Larger systems are rejected, not emulated.
#include <rocfft/rocfft.h>
#include "sc_common.h"

typedef double2 cd;
__device__ inline cd m_mul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline cd m_sub(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ inline cd m_conj(cd a) { return make_double2(a.x, -a.y); }
__device__ inline cd m_div(cd a, cd b) {
    const double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

#define MV_CMAX 64

// ---- dense complex linear algebra on LDS-resident matrices (one workgroup, any block size) ------
// LU with partial pivoting of M (C x C, row-major), in place: unit-lower multipliers below the diagonal,
// U on and above.  Every row operation is also applied to R (C x NR, row-major), so on return
// R = L^-1 P R.  piv[k] = row swapped with k.  All threads of the block must call.
__device__ void mv_lu_forward(cd* M, cd* R, int C, int NR, int* piv) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = 0; k < C; ++k) {
        if (tid < 64) {              // wave 0: arg max_i>=k |M[i][k]|  (C <= 64: one candidate per lane)
            const int i = k + tid;
            double best = -1.0;
            int bi = k;
            if (i < C) { const cd v = M[i * C + k]; best = v.x * v.x + v.y * v.y; bi = i; }
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) piv[k] = bi;
        }
        __syncthreads();
        const int pv = piv[k];
        if (pv != k) {
            for (int j = tid; j < C + NR; j += nt) {
                cd* a = j < C ? &M[k * C + j] : &R[k * NR + (j - C)];
                cd* b = j < C ? &M[pv * C + j] : &R[pv * NR + (j - C)];
                const cd t = *a; *a = *b; *b = t;
            }
            __syncthreads();
        }
        const cd d = M[k * C + k];
        for (int i = k + 1 + tid; i < C; i += nt) M[i * C + k] = m_div(M[i * C + k], d);
        __syncthreads();
        // trailing update of M and R: thread (ty, tx) walks rows ty, ty + nty, ... and columns tx, tx + 32, ...
        // (no integer division in the O(C^3) loop)
        const int tx = tid & 31, ty = tid >> 5, nty = nt >> 5 ? nt >> 5 : 1, ntx = nt < 32 ? nt : 32;
        const int cm = C - k - 1;
        for (int i = k + 1 + ty; i < C; i += nty) {
            const cd l = M[i * C + k];
            for (int jj = (nt < 32 ? tid : tx); jj < cm + NR; jj += ntx) {
                if (jj < cm) {
                    const int j = k + 1 + jj;
                    M[i * C + j] = m_sub(M[i * C + j], m_mul(l, M[k * C + j]));
                } else {
                    const int j = jj - cm;
                    R[i * NR + j] = m_sub(R[i * NR + j], m_mul(l, R[k * NR + j]));
                }
            }
        }
        __syncthreads();
    }
}

// R <- L^-1 P R for a NEW right-hand side, with the factors and pivots left by mv_lu_forward.  The
// stored multipliers went through every later row swap (LAPACK layout, P M = L U), so ALL swaps are
// applied to R first and the unit-lower solve follows.
__device__ void mv_apply_forward(const cd* M, cd* R, int C, int NR, const int* piv) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int j = tid; j < NR; j += nt)          // a thread owns whole columns: the swaps need no barrier
        for (int k = 0; k < C; ++k) {
            const int pv = piv[k];
            if (pv != k) { const cd t = R[k * NR + j]; R[k * NR + j] = R[pv * NR + j]; R[pv * NR + j] = t; }
        }
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        const int tx = tid & 31, ty = tid >> 5, nty = nt >> 5 ? nt >> 5 : 1, ntx = nt < 32 ? nt : 32;
        for (int i = k + 1 + ty; i < C; i += nty) {
            const cd l = M[i * C + k];
            for (int j = (nt < 32 ? tid : tx); j < NR; j += ntx) R[i * NR + j] = m_sub(R[i * NR + j], m_mul(l, R[k * NR + j]));
        }
        __syncthreads();
    }
}

// R <- U^-1 R
__device__ void mv_back_subst(const cd* M, cd* R, int C, int NR) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = C - 1; k >= 0; --k) {
        const cd d = M[k * C + k];
        for (int j = tid; j < NR; j += nt) R[k * NR + j] = m_div(R[k * NR + j], d);
        __syncthreads();
        const int tx = tid & 31, ty = tid >> 5, nty = nt >> 5 ? nt >> 5 : 1, ntx = nt < 32 ? nt : 32;
        for (int i = ty; i < k; i += nty) {
            const cd l = M[i * C + k];
            for (int j = (nt < 32 ? tid : tx); j < NR; j += ntx) R[i * NR + j] = m_sub(R[i * NR + j], m_mul(l, R[k * NR + j]));
        }
        __syncthreads();
    }
}

// in-place conjugate transpose of a C x C matrix
__device__ void mv_ctranspose(cd* X, int C) {
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) {
        const int i = idx / C, j = idx % C;
        if (i < j) {
            const cd a = X[i * C + j], b = X[j * C + i];
            X[i * C + j] = m_conj(b);
            X[j * C + i] = m_conj(a);
        } else if (i == j) {
            X[idx] = m_conj(X[idx]);
        }
    }
    __syncthreads();
}

// ---- spectra in, factor out ---------------------------------------------------------------------
struct MvDims {
    int64_t P, N, F;     // windows, two-sided FFT length, accumulated bins per window
    int C, NB, n_tiles, p_csm, two_sided;
    int64_t floats_per_bin;
    double n_obs;
};

// S[p][e][n] from the accumulator records (upper-triangular 16x16 tiles, un-normalised sums)
__global__ void m_build(ScRec accum, MvDims d, cd* S) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    const int64_t p = blockIdx.z;
    if (n >= d.N) return;
    const int i = e / d.C, j = e % d.C;
    int64_t bin = n;
    bool conj = false;
    if (!d.two_sided && n > d.N / 2) { bin = d.N - n; conj = true; }   // real input: S(-f) = conj S(f)
    const ScRec rec = accum + (p * d.F + bin) * d.floats_per_bin;
    int ti = i >> 4, tj = j >> 4, ii = i & 15, jj = j & 15;
    const bool m = (ti > tj) || (ti == tj && ii > jj);
    if (m) { int t = ti; ti = tj; tj = t; t = ii; ii = jj; jj = t; }
    const int64_t off = (int64_t)sc_tile_index(ti, tj, d.NB) * SC_TILE_ELEMS + ii * 16 + jj;
    const double re = (double)rec[(int64_t)d.p_csm * d.n_tiles * SC_TILE_ELEMS + off] / d.n_obs;
    double im = (double)rec[(int64_t)(d.p_csm + 1) * d.n_tiles * SC_TILE_ELEMS + off] / d.n_obs;
    if (m) im = -im;
    if (conj) im = -im;
    if (i == j) im = 0.0;
    S[(p * d.C * d.C + e) * d.N + n] = make_double2(re, im);
}

// natural [p][n][e] <-> series [p][e][n]
__global__ void m_to_series(const cd* nat, cd* ser, int64_t N, int E) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    const int64_t p = blockIdx.z;
    if (n < N) ser[(p * E + e) * N + n] = nat[(p * N + n) * E + e];
}
__global__ void m_to_natural(const cd* ser, cd* nat, int64_t N, int E) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    const int64_t p = blockIdx.z;
    if (n < N) nat[(p * N + n) * E + e] = ser[(p * E + e) * N + n];
}

// one block per window: R0 = Re mean_n S[n]; lower Cholesky; G0 = L^T for every n; status -1 if not PD
__global__ void __launch_bounds__(256) m_init(const cd* S, cd* G, int32_t* status, int64_t N, int C) {
    extern __shared__ double r0[];        // [C][C]
    __shared__ int bad;
    const int64_t p = blockIdx.x;
    const int E = C * C;
    if (threadIdx.x == 0) bad = 0;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const cd* s = S + (p * E + e) * N;
        double a = 0.0;
        for (int64_t n = 0; n < N; ++n) a += s[n].x;
        r0[e] = a / (double)N;
    }
    __syncthreads();
    for (int k = 0; k < C; ++k) {
        if (threadIdx.x == 0) {
            const double v = r0[k * C + k];
            if (!(v > 0.0)) bad = 1;
            r0[k * C + k] = sqrt(v);
        }
        __syncthreads();
        const double dk = r0[k * C + k];
        for (int i = k + 1 + threadIdx.x; i < C; i += blockDim.x) r0[i * C + k] /= dk;
        __syncthreads();
        const int rows = C - k - 1;
        for (int idx = threadIdx.x; idx < rows * rows; idx += blockDim.x) {
            const int i = k + 1 + idx / rows, j = k + 1 + idx % rows;
            if (j <= i) r0[i * C + j] -= r0[i * C + k] * r0[j * C + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) status[p] = bad ? -1 : 0;       // -1: not positive definite (LinAlgError)
    for (int e = 0; e < E; ++e) {
        const int i = e / C, j = e % C;
        const double v = (j >= i) ? r0[j * C + i] : 0.0;   // upper triangular L^T
        cd* g = G + (p * E + e) * N;
        for (int64_t n = threadIdx.x; n < N; n += blockDim.x) g[n] = make_double2(v, 0.0);
    }
}

__global__ void m_predict(const cd* S, const cd* G, const int32_t* status, cd* A, int64_t N, int C) {
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* Gl = reinterpret_cast<cd*>(mv_smem);
    cd* X = Gl + C * C;
    int* piv = reinterpret_cast<int*>(X + C * C);
    const int64_t n = blockIdx.x, p = blockIdx.y;
    if (status[p] != 0) return;
    const int E = C * C;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        Gl[e] = G[(p * E + e) * N + n];
        X[e] = S[(p * E + e) * N + n];
    }
    __syncthreads();
    mv_lu_forward(Gl, X, C, C, piv);       // X = L^-1 P S
    mv_back_subst(Gl, X, C, C);            // X = G^-1 S
    mv_ctranspose(X, C);                   // X = (G^-1 S)^H
    mv_apply_forward(Gl, X, C, C, piv);
    mv_back_subst(Gl, X, C, C);            // X = G^-1 (G^-1 S)^H
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        cd v = X[e];
        if (e / C == e % C) v.x += 1.0;
        A[(p * E + e) * N + n] = v;
    }
}

__global__ void m_causal(cd* A, int64_t N, int C) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    const int64_t p = blockIdx.z;
    if (n >= N) return;
    const int i = e / C, j = e % C;
    double sc = (n < (N + 1) / 2) ? 1.0 / (double)N : 0.0;
    if (n == 0) { sc *= 0.5; if (i > j) sc = 0.0; }
    cd* a = A + (p * C * C + e) * N + n;
    *a = make_double2(a->x * sc, a->y * sc);
}

__device__ inline void mv_atomic_max_nonneg(double* addr, double v) {
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

__global__ void m_update(cd* G, const cd* Aplus, const int32_t* status, double* err, int64_t N, int C) {
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* Gl = reinterpret_cast<cd*>(mv_smem);
    cd* Al = Gl + C * C;
    __shared__ double red[256];
    const int64_t n = blockIdx.x, p = blockIdx.y;
    if (status[p] != 0) return;
    const int E = C * C;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        Gl[e] = G[(p * E + e) * N + n];
        Al[e] = Aplus[(p * E + e) * N + n];
    }
    __syncthreads();
    double emax = 0.0;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int i = e / C, j = e % C;
        cd acc = make_double2(0.0, 0.0);
        for (int k = 0; k < C; ++k) {
            const cd t = m_mul(Gl[i * C + k], Al[k * C + j]);
            acc.x += t.x; acc.y += t.y;
        }
        const cd dlt = m_sub(acc, Gl[e]);
        emax = fmax(emax, hypot(dlt.x, dlt.y));
        G[(p * E + e) * N + n] = acc;
    }
    red[threadIdx.x] = emax;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] > 0.0) mv_atomic_max_nonneg(err + p, red[0]);
}

__global__ void m_flags(int32_t* status, int32_t* n_iter, double* err, double tol, int64_t P, int32_t* n_running) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (status[p] == 0) {
        n_iter[p] += 1;
        if (err[p] < tol) status[p] = 1;
        else atomicAdd(n_running, 1);
    }
    err[p] = 0.0;
}

// ---- measures -----------------------------------------------------------------------------------
// one block per window: H0 = Re mean_n G[n] (natural layout [p][n][e]); partial sum of H0^2
__global__ void __launch_bounds__(256) m_h0(const cd* G, double* h0, double* sq, int64_t N, int E) {
    __shared__ double red[256];
    const int64_t p = blockIdx.x;
    double s2 = 0.0;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        double a = 0.0;
        for (int64_t n = 0; n < N; ++n) a += G[(p * N + n) * E + e].x;
        a /= (double)N;
        h0[p * E + e] = a;
        s2 += a * a;
    }
    red[threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sq[p] = red[0];
}

// serial, fixed-order sum of a short array (per-window / per-bin partials): out[0] = scale * sum
__global__ void m_sum(const double* v, int64_t n, double scale, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0.0;
        for (int64_t i = 0; i < n; ++i) a += v[i];
        out[0] = a * scale;
    }
}

// one block per window: Hinv = (H0 + lam I)^-1 (connectivity.py:1739-1746), Sigma = H0 H0^T (:1703-1708)
__global__ void m_h0_inverse(const double* h0, const double* lam, double* hinv, double* sigma, int C) {
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* M = reinterpret_cast<cd*>(mv_smem);
    cd* R = M + C * C;
    int* piv = reinterpret_cast<int*>(R + C * C);
    const int64_t p = blockIdx.x;
    const int E = C * C;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int i = e / C, j = e % C;
        M[e] = make_double2(h0[p * E + e] + (i == j ? lam[0] : 0.0), 0.0);
        R[e] = make_double2(i == j ? 1.0 : 0.0, 0.0);
        double s = 0.0;
        for (int k = 0; k < C; ++k) s += h0[p * E + i * C + k] * h0[p * E + j * C + k];
        sigma[p * E + e] = s;
    }
    __syncthreads();
    mv_lu_forward(M, R, C, C, piv);
    mv_back_subst(M, R, C, C);
    for (int e = threadIdx.x; e < E; e += blockDim.x) hinv[p * E + e] = R[e].x;
}

// H[p][f] = G[p][f] Hinv[p] on the non-negative bins; partial sums of |H|^2 per (p, f)
__global__ void m_transfer(const cd* G, const double* hinv, cd* H, double* sq, int64_t N, int64_t F, int C) {
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* Gl = reinterpret_cast<cd*>(mv_smem);
    double* hi = reinterpret_cast<double*>(Gl + C * C);
    __shared__ double red[256];
    const int64_t f = blockIdx.x, p = blockIdx.y;
    const int E = C * C;
    for (int e = threadIdx.x; e < E; e += blockDim.x) { Gl[e] = G[(p * N + f) * E + e]; hi[e] = hinv[p * E + e]; }
    __syncthreads();
    double s2 = 0.0;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int i = e / C, j = e % C;
        cd acc = make_double2(0.0, 0.0);
        for (int k = 0; k < C; ++k) { const double w = hi[k * C + j]; acc.x += Gl[i * C + k].x * w; acc.y += Gl[i * C + k].y * w; }
        H[(p * F + f) * E + e] = acc;
        s2 += acc.x * acc.x + acc.y * acc.y;
    }
    red[threadIdx.x] = s2;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sq[p * F + f] = red[0];
}

// A_mvar[p][f] = (H + lam' I)^-1   (connectivity.py:581-589)
__global__ void m_mvar_inverse(const cd* H, const double* lam, cd* Amv, int C) {
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* M = reinterpret_cast<cd*>(mv_smem);
    cd* R = M + C * C;
    int* piv = reinterpret_cast<int*>(R + C * C);
    const int64_t b = blockIdx.x;                 // p * F + f
    const int E = C * C;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int i = e / C, j = e % C;
        cd v = H[b * E + e];
        if (i == j) v.x += lam[0];
        M[e] = v;
        R[e] = make_double2(i == j ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    mv_lu_forward(M, R, C, C, piv);
    mv_back_subst(M, R, C, C);
    for (int e = threadIdx.x; e < E; e += blockDim.x) Amv[b * E + e] = R[e];
}

// inflow over frequencies and sources: tot[p][i] = sum_f sum_j |H_ij|^2 (dDTF, connectivity.py:1420-1422)
__global__ void m_inflow_all(const cd* H, double* tot, int64_t F, int C) {
    const int64_t p = blockIdx.x;
    const int E = C * C;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        double a = 0.0;
        for (int64_t f = 0; f < F; ++f)
            for (int j = 0; j < C; ++j) { const cd v = H[(p * F + f) * E + i * C + j]; a += v.x * v.x + v.y * v.y; }
        tot[p * C + i] = a;
    }
}

// one block per (p, f): out[p][f][i][j]
__global__ void m_measure(const cd* H, const cd* Amv, const double* sigma, const double* tot, int which,
                          double* out, int64_t F, int C) {
    extern __shared__ __align__(16) unsigned char mv_smem[];
    double* pw = reinterpret_cast<double*>(mv_smem);      // |H_ij|^2 or |A_ij|^2
    double* nrm = pw + C * C;                               // per-row (inflow) or per-column (outflow) sums
    const int64_t b = blockIdx.x, p = b / F;
    const int E = C * C;
    const bool use_a = which == SC_MVAR_PDC || which == SC_MVAR_GPDC;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const cd v = use_a ? Amv[b * E + e] : H[b * E + e];
        pw[e] = v.x * v.x + v.y * v.y;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < C; r += blockDim.x) {
        double a = 0.0;
        if (which == SC_MVAR_DTF) {
            for (int j = 0; j < C; ++j) a += pw[r * C + j];                       // inflow into i = r
        } else if (which == SC_MVAR_DC) {
            const double nv = sigma[p * E + r * C + r];
            for (int j = 0; j < C; ++j) a += nv * pw[r * C + j];
        } else if (which == SC_MVAR_PDC || which == SC_MVAR_DDTF) {
            if (which == SC_MVAR_DDTF) {                                          // PDC needs |A|^2 column sums
                for (int i = 0; i < C; ++i) { const cd v = Amv[b * E + i * C + r]; a += v.x * v.x + v.y * v.y; }
            } else {
                for (int i = 0; i < C; ++i) a += pw[i * C + r];                   // outflow from j = r
            }
        } else {                                                                  // gPDC
            for (int i = 0; i < C; ++i) a += pw[i * C + r] / sigma[p * E + i * C + i];
        }
        nrm[r] = a;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int i = e / C, j = e % C;
        double v;
        if (which == SC_MVAR_DTF) {
            // |H_ij / sqrt(sum_j |H_ij|^2)|^2
            const double s = sqrt(nrm[i]);
            const cd h = H[b * E + e];
            const double re = h.x / s, im = h.y / s;
            v = re * re + im * im;
        } else if (which == SC_MVAR_DC) {
            v = sqrt(sigma[p * E + i * C + i]) * pw[e] / sqrt(nrm[i]);
        } else if (which == SC_MVAR_PDC) {
            const double s = sqrt(nrm[j]);
            const cd a = Amv[b * E + e];
            const double re = a.x / s, im = a.y / s;
            v = re * re + im * im;
        } else if (which == SC_MVAR_GPDC) {
            const double sn = sqrt(sigma[p * E + i * C + i]), s = sqrt(nrm[j]);
            const cd a = Amv[b * E + e];
            const double re = a.x / sn / s, im = a.y / sn / s;
            v = re * re + im * im;
        } else {                                                                  // dDTF
            const double s = sqrt(tot[p * C + i]), sp = sqrt(nrm[j]);
            const cd h = H[b * E + e], a = Amv[b * E + e];
            const double hr = h.x / s, hi = h.y / s, ar = a.x / sp, ai = a.y / sp;
            v = sqrt(hr * hr + hi * hi) * sqrt(ar * ar + ai * ai);
        }
        out[b * E + e] = v;
    }
}

// ---- host side ------------------------------------------------------------------------------------
static int mv_make_z2z(rocfft_plan* plan, rocfft_transform_type type, size_t N, size_t batch) {
    size_t lengths[1] = {N};
    rocfft_status s = rocfft_plan_create(plan, rocfft_placement_inplace, type, rocfft_precision_double, 1,
                                         lengths, batch, nullptr);
    if (s != rocfft_status_success) {
        sc_set_error("rocfft_plan_create(Z2Z N=%zu batch=%zu) failed: %d", N, batch, (int)s);
        return SC_EFFT;
    }
    return SC_OK;
}

#define MV_CHECK_FFT(expr)                                                                       \
    do {                                                                                         \
        rocfft_status s_ = (expr);                                                               \
        if (s_ != rocfft_status_success) {                                                       \
            sc_set_error("%s failed: rocfft_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
            rc = SC_EFFT; goto done;                                                             \
        }                                                                                        \
    } while (0)

static int mv_threads(int C) {
    const int e = C * C;
    return e >= 256 ? 256 : (e > 128 ? 256 : (e > 64 ? 128 : 64));
}

static size_t mv_pair_lds(int C) { return (size_t)2 * C * C * sizeof(cd) + (size_t)C * sizeof(int) + 16; }

extern "C" int sc_mvar_max_signals(void) { return MV_CMAX; }

extern "C" int sc_mvar_workspace_bytes(int64_t n_groups, int64_t C, int64_t N, size_t* bytes) {
    SC_REQUIRE(bytes && n_groups >= 1 && C >= 1 && N >= 2, "bad workspace query");
    const size_t E = (size_t)C * C, P = (size_t)n_groups, F = (size_t)N / 2 + 1;
    // factor: S, G, A series; measures: H, A_mvar natural + small per-window arrays
    const size_t factor = 3 * P * E * (size_t)N * sizeof(cd) + P * 16 + 64;
    const size_t meas = 2 * P * F * E * sizeof(cd) + P * E * 8 * 3 + P * F * 8 + P * (size_t)C * 8 + P * 8 + 256;
    *bytes = (factor > meas ? factor : meas) + 256;
    return SC_OK;
}

// d_accum (accumulator records) or d_S ([P][N][C][C] complex128, two-sided) -> d_G [P][N][C][C] complex128
extern "C" int sc_mvar_factor_f64(const void* d_accum, const void* d_S, int64_t n_groups, int64_t n_freq_accum,
                                  int64_t N, int64_t C, uint32_t planes, int64_t n_obs, double tol, int max_iter,
                                  void* d_work, size_t work_bytes, void* d_G, int32_t* d_n_iter, int32_t* d_status,
                                  int32_t* h_summary, void* stream) {
    ScTimed timed_("mvar_factor", stream);
    SC_REQUIRE((d_accum != nullptr) != (d_S != nullptr), "pass exactly one of d_accum and d_S");
    SC_REQUIRE(d_work && d_G && d_n_iter && d_status, "NULL argument");
    SC_REQUIRE(n_groups >= 1 && n_groups <= 65535 && N >= 2 && N <= 1 << 24, "bad problem size");
    if (C < 1 || C > MV_CMAX) {
        sc_set_error("full Wilson factorisation keeps a C x C matrix pair in LDS: n_signals <= %d (got %lld)", MV_CMAX,
                     (long long)C);
        return SC_EUNSUPPORTED;
    }
    size_t need = 0;
    sc_mvar_workspace_bytes(n_groups, C, N, &need);
    SC_REQUIRE(work_bytes >= need, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = n_groups;
    const int E = (int)(C * C);
    char* w = (char*)d_work;
    cd* S = (cd*)w; w += (size_t)P * E * N * sizeof(cd);
    cd* G = (cd*)w; w += (size_t)P * E * N * sizeof(cd);
    cd* A = (cd*)w; w += (size_t)P * E * N * sizeof(cd);
    double* err = (double*)w; w += (size_t)P * 8;
    int32_t* n_running = (int32_t*)w;
    const dim3 gridE((unsigned)((N + 255) / 256), (unsigned)E, (unsigned)P);
    if (d_accum) {
        SC_REQUIRE(planes & SC_PLANE_CSM, "accumulator record must contain SC_PLANE_CSM");
        SC_REQUIRE(n_freq_accum == N || n_freq_accum == N / 2 + 1, "accumulators must hold N or N/2+1 bins");
        MvDims d;
        d.P = P; d.N = N; d.F = n_freq_accum; d.C = (int)C;
        d.NB = sc_n_blocks(C); d.n_tiles = sc_n_tiles(d.NB);
        d.p_csm = sc_plane_offset(planes, SC_PLANE_CSM);
        d.two_sided = (n_freq_accum == N && N > 1) ? 1 : 0;
        d.floats_per_bin = (int64_t)sc_plane_count(planes) * d.n_tiles * SC_TILE_ELEMS;
        d.n_obs = (double)n_obs;
        hipLaunchKernelGGL(m_build, gridE, dim3(256), 0, st, sc_rec(d_accum, planes), d, S);
    } else {
        hipLaunchKernelGGL(m_to_series, gridE, dim3(256), 0, st, (const cd*)d_S, S, N, E);
    }
    int rc = SC_OK;
    rocfft_plan fwd = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    void* fft_work = nullptr;
    size_t ws_f = 0, ws_i = 0;
    static int rocfft_ready = 0;
    if (!rocfft_ready) { rocfft_setup(); rocfft_ready = 1; }
    const int nt = mv_threads((int)C);
    const size_t lds = mv_pair_lds((int)C);
    const dim3 gridB((unsigned)N, (unsigned)P);
    int iters = 0, running = (int)P;
    const bool fused = sc_internal_causal_fft_supported(N);
    (void)hipFuncSetAttribute((const void*)m_predict, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)m_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (!fused) {
        if ((rc = mv_make_z2z(&fwd, rocfft_transform_type_complex_forward, (size_t)N, (size_t)E * P)) != SC_OK) goto done;
        if ((rc = mv_make_z2z(&inv, rocfft_transform_type_complex_inverse, (size_t)N, (size_t)E * P)) != SC_OK) goto done;
        MV_CHECK_FFT(rocfft_plan_get_work_buffer_size(fwd, &ws_f));
        MV_CHECK_FFT(rocfft_plan_get_work_buffer_size(inv, &ws_i));
        MV_CHECK_FFT(rocfft_execution_info_create(&info));
        if (ws_f < ws_i) ws_f = ws_i;
        if (ws_f) {
            if (hipMalloc(&fft_work, ws_f) != hipSuccess) { sc_set_error("rocFFT work buffer alloc failed"); rc = SC_ENOMEM; goto done; }
            MV_CHECK_FFT(rocfft_execution_info_set_work_buffer(info, fft_work, ws_f));
        }
        MV_CHECK_FFT(rocfft_execution_info_set_stream(info, st));
    }
    (void)hipMemsetAsync(err, 0, (size_t)P * 8, st);
    (void)hipMemsetAsync(d_n_iter, 0, (size_t)P * 4, st);
    hipLaunchKernelGGL(m_init, dim3((unsigned)P), dim3(256), (size_t)E * 8, st, S, G, d_status, N, (int)C);
    for (iters = 0; iters < max_iter; ++iters) {
        void* bufs[1] = {A};
        hipLaunchKernelGGL(m_predict, gridB, dim3(nt), lds, st, S, G, d_status, A, N, (int)C);
        if (fused) {        // ifft -> causal mask -> fft in one kernel (sc_wilson_fft.hip)
            if ((rc = sc_internal_causal_fft_pair(A, d_status, P, (int)C, N, st)) != SC_OK) goto done;
        } else {
            MV_CHECK_FFT(rocfft_execute(inv, bufs, nullptr, info));
            hipLaunchKernelGGL(m_causal, gridE, dim3(256), 0, st, A, N, (int)C);
            MV_CHECK_FFT(rocfft_execute(fwd, bufs, nullptr, info));
        }
        hipLaunchKernelGGL(m_update, gridB, dim3(nt), lds, st, G, A, d_status, err, N, (int)C);
        (void)hipMemsetAsync(n_running, 0, 4, st);
        hipLaunchKernelGGL(m_flags, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, d_status, d_n_iter, err, tol, P,
                           n_running);
        if (hipMemcpyAsync(&running, n_running, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            sc_set_error("Wilson iteration %d: %s", iters, hipGetErrorString(hipGetLastError()));
            rc = SC_EHIP; goto done;
        }
        if (running == 0) { ++iters; break; }
    }
    hipLaunchKernelGGL(m_to_natural, gridE, dim3(256), 0, st, G, (cd*)d_G, N, E);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        sc_set_error("Wilson factor copy-out failed: %s", hipGetErrorString(hipGetLastError()));
        rc = SC_EHIP; goto done;
    }
    if (h_summary) { h_summary[0] = iters; h_summary[1] = running; }
done:
    if (info) rocfft_execution_info_destroy(info);
    if (fwd) rocfft_plan_destroy(fwd);
    if (inv) rocfft_plan_destroy(inv);
    if (fft_work) (void)hipFree(fft_work);
    return rc;
}

// d_G [P][N][C][C] complex128 -> the requested quantity.
//   SC_MVAR_DTF .. SC_MVAR_DDTF : double  [P][N/2+1][C][C]
//   SC_MVAR_TRANSFER, SC_MVAR_COEFFICIENTS : complex128 [P][N/2+1][C][C]
//   SC_MVAR_NOISE_COVARIANCE : double [P][C][C]
extern "C" int sc_mvar_measure_f64(const void* d_G, int64_t n_groups, int64_t N, int64_t C, int which, void* d_out,
                                   void* d_work, size_t work_bytes, void* stream) {
    ScTimed timed_("mvar_measure", stream);
    SC_REQUIRE(d_G && d_out && d_work, "NULL argument");
    SC_REQUIRE(which >= SC_MVAR_DTF && which <= SC_MVAR_NOISE_COVARIANCE, "unknown MVAR quantity");
    if (C < 1 || C > MV_CMAX) {
        sc_set_error("MVAR measures keep a C x C matrix pair in LDS: n_signals <= %d (got %lld)", MV_CMAX, (long long)C);
        return SC_EUNSUPPORTED;
    }
    size_t need = 0;
    sc_mvar_workspace_bytes(n_groups, C, N, &need);
    SC_REQUIRE(work_bytes >= need, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = n_groups, F = N / 2 + 1;
    const int E = (int)(C * C);
    char* w = (char*)d_work;
    cd* H = (cd*)w; w += (size_t)P * F * E * sizeof(cd);
    cd* Amv = (cd*)w; w += (size_t)P * F * E * sizeof(cd);
    double* h0 = (double*)w; w += (size_t)P * E * 8;
    double* hinv = (double*)w; w += (size_t)P * E * 8;
    double* sigma = (double*)w; w += (size_t)P * E * 8;
    double* sq = (double*)w; w += (size_t)P * F * 8;
    double* tot = (double*)w; w += (size_t)P * C * 8;
    double* lam = (double*)w;                         // [0] lam of H0, [1] lam' of H
    const int nt = mv_threads((int)C);
    const size_t lds = mv_pair_lds((int)C);
    (void)hipFuncSetAttribute((const void*)m_h0_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)m_transfer, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)m_mvar_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const cd* G = (const cd*)d_G;
    hipLaunchKernelGGL(m_h0, dim3((unsigned)P), dim3(256), 0, st, G, h0, sq, N, E);
    hipLaunchKernelGGL(m_sum, dim3(1), dim3(64), 0, st, sq, P, 1e-12 / (double)(P * E), lam);
    hipLaunchKernelGGL(m_h0_inverse, dim3((unsigned)P), dim3(nt), lds, st, h0, lam, hinv, sigma, (int)C);
    if (which == SC_MVAR_NOISE_COVARIANCE) {
        SC_CHECK_HIP(hipMemcpyAsync(d_out, sigma, (size_t)P * E * 8, hipMemcpyDeviceToDevice, st));
        SC_CHECK_HIP(hipStreamSynchronize(st));
        return SC_OK;
    }
    hipLaunchKernelGGL(m_transfer, dim3((unsigned)F, (unsigned)P), dim3(nt), lds, st, G, hinv, H, sq, N, F, (int)C);
    if (which == SC_MVAR_TRANSFER) {
        SC_CHECK_HIP(hipMemcpyAsync(d_out, H, (size_t)P * F * E * sizeof(cd), hipMemcpyDeviceToDevice, st));
        SC_CHECK_HIP(hipStreamSynchronize(st));
        return SC_OK;
    }
    hipLaunchKernelGGL(m_sum, dim3(1), dim3(64), 0, st, sq, P * F, 1e-12 / (double)(P * F * E), lam + 1);
    hipLaunchKernelGGL(m_mvar_inverse, dim3((unsigned)(P * F)), dim3(nt), lds, st, H, lam + 1, Amv, (int)C);
    if (which == SC_MVAR_COEFFICIENTS) {
        SC_CHECK_HIP(hipMemcpyAsync(d_out, Amv, (size_t)P * F * E * sizeof(cd), hipMemcpyDeviceToDevice, st));
        SC_CHECK_HIP(hipStreamSynchronize(st));
        return SC_OK;
    }
    if (which == SC_MVAR_DDTF) hipLaunchKernelGGL(m_inflow_all, dim3((unsigned)P), dim3(64), 0, st, H, tot, F, (int)C);
    hipLaunchKernelGGL(m_measure, dim3((unsigned)(P * F)), dim3(nt), (size_t)(E + C) * 8, st, H, Amv, sigma, tot, which,
                       (double*)d_out, F, (int)C);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        sc_set_error("MVAR measure failed: %s", hipGetErrorString(hipGetLastError()));
        return SC_EHIP;
    }
    return SC_OK;
}
