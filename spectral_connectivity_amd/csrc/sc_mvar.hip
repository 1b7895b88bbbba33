// sc_mvar.hip -- full C x C Wilson spectral factorisation, batched over time windows, and the
// directed measures of the implied MVAR model.
//
// Reference path: Connectivity._minimum_phase_factor / _transfer_function / _noise_covariance /
// _MVAR_Fourier_coefficients (connectivity.py:567-589) -> minimum_phase_decomposition
// (minimum_phase_decomposition.py:227-322) and directed_transfer_function, directed_coherence,
// partial_directed_coherence, generalized_partial_directed_coherence,
// direct_directed_transfer_function (connectivity.py:1237-1426, helpers :1679-1748, :1873-1950).
//
//   m_build / m_to_series   two-sided Hermitian spectra S[p][e][n] (e = i C + j, n fastest) from the records / from d_S
//   m_lag0, m_chol, m_fill  G0 = chol(Re ifft_n(S)[lag 0])^H, broadcast over n (identity where the Cholesky fails)
//   loop <= max_iter (all windows at once, converged windows frozen, convergence polled every 4 iterations):
//     m_predict_gj          A = G^-1 S G^-H + I: one workgroup per (window, bin), [G | S] in registers, Gauss-Jordan
//                           with partial pivoting, the second factor as column operations from the logged multipliers
//                           (65 ... 128 signals: m_inverse_inplace + two m_gemm_mfma products)
//     causal transform pair a = ifft_n(A), 1/N, a[0] *= 1/2, strict lower triangle of a[0] = 0, a[n >= (N+1)/2] = 0,
//                           A+ = fft_n(a): one kernel (sc_wilson_fft.hip) for N = 256 ... 4096, else rocFFT Z2Z + m_causal
//     m_update_mfma         G <- G A+, err = max |G - G_old| on the fp64 matrix cores (m_gemm_mfma beyond 64 signals)
//   measures                H0 = Re mean_n G; H = G (H0 + lam I)^-1 on the non-negative bins;
//                           A_mvar = (H + lam' I)^-1; Sigma = H0 H0^T; DTF / DC / PDC / gPDC / dDTF.
// Everything is fp64 (the reference's convergence test max |dG| < 1e-8 is out of fp32's reach).  Up to 128 signals the
// C x C factor of one (window, bin) lives in the registers of one workgroup; 129 ... 512 signals (round 6; 256 before: the most an accumulator
// record holds) run the same iteration with the inverse as a panel-blocked Gauss-Jordan on the matrix in global memory
// (m_inverse_global) and the products as 128 x 128 output blocks of m_gemm_mfma.
#include <stdlib.h>
#include <string.h>
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

#define MV_CMAX 512          // <= 64: register-resident [G | S] elimination; 65 ... 128: explicit in-register inverse + matrix-
#define MV_CSMALL 64         // core products; 129 ... 512: panel-blocked inverse in global memory + blocked products
#define MV_CMID 128
#define MV_GRID_Y 32768      // grid.y of the per-element kernels (C^2 = 65536 elements at 256 signals exceed the limit of 65535)

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

// ---- spectra in, factor out ---------------------------------------------------------------------
struct MvDims {
    int64_t P, N, F;     // windows, two-sided FFT length, accumulated bins per window
    int C, NB, n_tiles, p_csm, two_sided;
    int64_t floats_per_bin;
    double n_obs;
};

// S[p][e][n] from the accumulator records (upper-triangular 16x16 tiles, un-normalised sums)
__global__ void m_build(ScRec accum, MvDims d, cd* S, int64_t sn, int64_t se) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.z;
    if (n >= d.N) return;
    for (int e = blockIdx.y; e < d.C * d.C; e += gridDim.y) {
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
    S[p * d.C * d.C * d.N + n * sn + e * se] = make_double2(re, im);      // series: sn = 1, se = N; natural: sn = C^2, se = 1
    }
}

// natural [p][n][e] <-> series [p][e][n]
__global__ void m_to_series(const cd* nat, cd* ser, int64_t N, int E) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.z;
    if (n < N)
        for (int e = blockIdx.y; e < E; e += gridDim.y) ser[(p * E + e) * N + n] = nat[(p * N + n) * E + e];
}
__global__ void m_to_natural(const cd* ser, cd* nat, int64_t N, int E) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.z;
    if (n < N)
        for (int e = blockIdx.y; e < E; e += gridDim.y) nat[(p * N + n) * E + e] = ser[(p * E + e) * N + n];
}

// R0[p][e] = Re mean_n S[p][e][n]: one wave per series, lanes along n (unit stride)
__global__ void __launch_bounds__(256) m_lag0(const cd* __restrict__ S, double* __restrict__ r0, int64_t N, int64_t n_series) {
    const int64_t series = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (series >= n_series) return;
    const int lane = threadIdx.x & 63;
    const cd* s = S + series * N;
    double a = 0.0;
    for (int64_t n = lane; n < N; n += 64) a += s[n].x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) r0[series] = a / (double)N;
}

// the same for the natural layout [p][n][e] (beyond 64 signals): threads along e
__global__ void __launch_bounds__(256) m_lag0_nat(const cd* __restrict__ S, double* __restrict__ r0, int64_t N, int E) {
    const int64_t p = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    double a = 0.0;
    for (int64_t n = 0; n < N; ++n) a += S[(p * N + n) * E + e].x;
    r0[p * E + e] = a / (double)N;
}
__global__ void m_fill_nat(const double* __restrict__ g0, cd* __restrict__ G, int64_t N, int E) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int64_t n = blockIdx.y, p = blockIdx.z;
    if (e < E) G[(p * N + n) * E + e] = make_double2(g0[p * E + e], 0.0);
}

// one block per window: lower Cholesky of R0 in LDS, G0 = L^T written back over r0 (upper triangular, real); a lag-0
// covariance that is not positive definite leaves the identity there (the expectation of the reference's random
// Wishart start, minimum_phase_decomposition.py:78-93) and is counted in *n_fallback
// (in_place: beyond 128 signals the matrix does not fit LDS -- the factor is built where it lies, in global memory)
__global__ void __launch_bounds__(256) m_chol(double* r0g, int32_t* status, int32_t* n_fallback, int C, int in_place) {
    extern __shared__ double r0_lds[];        // [C][C]
    __shared__ int bad;
    const int64_t p = blockIdx.x;
    const int E = C * C;
    double* r0 = in_place ? r0g + p * E : r0_lds;
    if (threadIdx.x == 0) bad = 0;
    if (!in_place)
        for (int e = threadIdx.x; e < E; e += blockDim.x) r0[e] = r0g[p * E + e];
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
    if (threadIdx.x == 0) {
        status[p] = 0;
        if (bad) atomicAdd(n_fallback, 1);
    }
    if (in_place) {         // transpose the lower triangle into the upper one pair by pair
        for (int e = threadIdx.x; e < E; e += blockDim.x) {
            const int i = e / C, j = e % C;
            if (j > i) { r0[i * C + j] = bad ? 0.0 : r0[j * C + i]; r0[j * C + i] = 0.0; }
            else if (j == i && bad) r0[e] = 1.0;
        }
        return;
    }
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int i = e / C, j = e % C;
        r0g[p * E + e] = bad ? (i == j ? 1.0 : 0.0) : ((j >= i) ? r0[j * C + i] : 0.0);      // upper triangular L^T
    }
}

// The reference's batched Cholesky fails as a whole (minimum_phase_decomposition.py:78-93): one window without a factor
// restarts EVERY window from its random draw (expectation: a multiple of the identity), and at a finite FFT length the
// fixed point depends on the start -- so one failing window puts the identity into all of them (see k_restart in
// sc_wilson.hip); *n_fallback then counts the windows restarted.
__global__ void __launch_bounds__(256) m_restart_all(double* __restrict__ r0g, int32_t* n_fallback, int64_t P, int C) {
    if (*n_fallback == 0) return;
    const int64_t total = P * C * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int e = (int)(i % ((int64_t)C * C));
        r0g[i] = (e / C == e % C) ? 1.0 : 0.0;
    }
}
__global__ void m_restart_count(int32_t* n_fallback, int32_t P) {
    if (*n_fallback > 0) *n_fallback = P;
}

// G[p][e][n] = G0[p][e] for every n
__global__ void m_fill(const double* __restrict__ g0, cd* __restrict__ G, int64_t N, int E) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.z;
    if (n < N)
        for (int e = blockIdx.y; e < E; e += gridDim.y) G[(p * E + e) * N + n] = make_double2(g0[p * E + e], 0.0);
}

__global__ void m_causal(cd* A, int64_t N, int C) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.z;
    if (n >= N) return;
    for (int e = blockIdx.y; e < C * C; e += gridDim.y) {
        const int i = e / C, j = e % C;
        double sc = (n < (N + 1) / 2) ? 1.0 / (double)N : 0.0;
        if (n == 0) { sc *= 0.5; if (i > j) sc = 0.0; }
        cd* a = A + (p * C * C + e) * N + n;
        *a = make_double2(a->x * sc, a->y * sc);
    }
}

__device__ inline void mv_atomic_max_nonneg(double* addr, double v) {
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

// ---- predict / update of the Wilson iteration, second generation ------------------------------------------------
// The first version (LU with partial pivoting on an LDS-resident matrix pair, three LDS accesses per complex FMA) was 3.2 ms
// per launch for 1792 problems of 64 x 64: LDS-bandwidth bound at 3 % of the fp64 rate.  m_predict_gj keeps the WHOLE
// augmented matrix [G | S] in registers: the 256 threads of a workgroup form a 16 x 16 grid, thread (ty, tx) owns rows
// ty + 16 a and columns tx + 16 b (a, b < Q, C <= 16 Q), i.e. Q^2 complex elements of G and of S.  Gauss-Jordan with
// partial pivoting, one column per step: the owners of column k publish it (and |.|^2 of the rows that have not been
// pivots yet) in LDS, every wave finds the pivot row with a butterfly of shuffles, the owners of that row publish it
// scaled by 1 / pivot, and every thread updates its own block from registers: four LDS reads of a column value and
// eight of a row value per step against 2 Q^2 complex FMAs -- the kernel runs at the fp64 VALU rate (which on MI355X
// IS the fp64 matrix rate: 78.6 TFLOP/s both, so a rank-4 v_mfma_f64 formulation of the elimination has nothing to
// gain).  No row is ever moved: after C steps G has become a permutation matrix, row pr_k of the right-hand side holds
// row k of Y = G^-1 S.  The second factor of A = G^-1 S G^-H + I re-uses the elimination instead of a second solve:
// with E = E_C ... E_1 the recorded row operations (E G = P), A - I = P^T (Y' E^H) P, and Y' E^H is the same sequence
// applied as COLUMN operations (column pr_k scaled by conj(1 / pivot), column r reduced by conj(m_r) times it) to the
// block the thread already holds -- half the work of the first pass, multipliers read back from LDS (C^2 complex).
// 2 syncs per step in the first pass, 1 in the second; 70 KB of LDS and ~180 registers: two workgroups per CU.
// maximum of one unsigned per lane over the wave, as a scalar
__device__ __forceinline__ unsigned mv_wave_max_u32(unsigned v) {
    // inside each row of 16 lanes: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_ror:4, row_ror:8
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false));
    const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}

// Pass 1 of the Gauss-Jordan elimination on the register-resident [G | S] (see m_predict_gj): row operations with
// partial pivoting, multipliers / pivots / pivot rows logged in LDS.  All 256 threads of the workgroup must call.
template <int Q>
__device__ __forceinline__ void mv_gj_eliminate(cd (&g)[Q][Q], cd (&s)[Q][Q], cd* mult, cd* colbuf, cd* rowbuf, cd* pinv,
                                                unsigned* key, int* prow, int* pos, int C) {
    constexpr int CP = 16 * Q;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, lane = tid & 63;
    unsigned used = 0;                                     // bit a: row ty + 16 a has been a pivot
    if (tid < CP) pos[tid] = tid < C ? tid : 0;            // (only a NaN input leaves an entry at this default)
#pragma unroll
    for (int kb = 0; kb < Q; ++kb) {
        for (int kx = 0; kx < 16; ++kx) {
            const int k = 16 * kb + kx;
            if (k >= C) break;
            cd* cb = colbuf + (k & 1) * CP;
            if (tx == kx) {
#pragma unroll
                for (int a = 0; a < Q; ++a) {
                    const int r = ty + 16 * a;
                    const cd v = g[a][kb];
                    cb[r] = v;
                    // key = leading 26 bits of |v|^2 (sign, exponent, 14 mantissa bits of the double: order preserving)
                    // over 63 - row: the maximum is the largest candidate, the lowest row among near-ties
                    const unsigned hi = (unsigned)(__double_as_longlong(v.x * v.x + v.y * v.y) >> 32);
                    key[r] = (((used >> a) & 1u) || r >= C) ? 0u : (((hi & ~63u) | (unsigned)(63 - r)) | 0x40u);
                }
            }
            __syncthreads();
            // pivot row: maximum key over the candidate rows, found by every wave for itself (CP <= 64: one key per
            // lane): four DPP steps inside each row of 16 lanes, then the four row maxima through scalar registers
            const int pr = 63 - (int)(mv_wave_max_u32(lane < CP ? key[lane] : 0u) & 63u);
            const cd piv = cb[pr];
            const double pden = piv.x * piv.x + piv.y * piv.y;
            const cd inv = make_double2(piv.x / pden, -piv.y / pden);
            if (ty == (pr & 15)) {         // one wave in four: the owners scale the pivot row in place and publish it
#pragma unroll
                for (int a = 0; a < Q; ++a)
                    if (ty + 16 * a == pr) {
#pragma unroll
                        for (int b = 0; b < Q; ++b) {
                            g[a][b] = m_mul(g[a][b], inv);
                            s[a][b] = m_mul(s[a][b], inv);
                            rowbuf[tx + 16 * b] = g[a][b];
                            rowbuf[CP + tx + 16 * b] = s[a][b];
                        }
                        used |= 1u << a;
                    }
            }
            if (tid == 0) { prow[k] = pr; pinv[k] = inv; pos[pr] = k; }
            __syncthreads();
            cd wg[Q], ws[Q];
#pragma unroll
            for (int b = 0; b < Q; ++b) { wg[b] = rowbuf[tx + 16 * b]; ws[b] = rowbuf[CP + tx + 16 * b]; }
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                const int r = ty + 16 * a;
                cd m = cb[r];
                if (r == pr) m = make_double2(0.0, 0.0);
                if (tx == 0) mult[k * CP + r] = m;
#pragma unroll
                for (int b = 0; b < Q; ++b) {
                    if (b >= kb) {      // columns tx + 16 b with b < kb were eliminated in earlier blocks: the pivot row is 0 there
                        g[a][b].x = fma(-m.x, wg[b].x, fma(m.y, wg[b].y, g[a][b].x));
                        g[a][b].y = fma(-m.x, wg[b].y, fma(-m.y, wg[b].x, g[a][b].y));
                    }
                    s[a][b].x = fma(-m.x, ws[b].x, fma(m.y, ws[b].y, s[a][b].x));
                    s[a][b].y = fma(-m.x, ws[b].y, fma(-m.y, ws[b].x, s[a][b].y));
                }
            }
        }
    }
    __syncthreads();
}

template <int Q>
__global__ void __launch_bounds__(256, 2) m_predict_gj(const cd* __restrict__ S, const cd* __restrict__ G,
                                                       const int32_t* __restrict__ status, cd* __restrict__ A,
                                                       int64_t N, int C) {
    constexpr int CP = 16 * Q;
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* mult = reinterpret_cast<cd*>(mv_smem);             // [C][CP] multipliers m_r of step k (0 for the pivot row)
    cd* colbuf = mult + CP * CP;                           // [2][CP]  column k (double buffered)
    cd* rowbuf = colbuf + 2 * CP;                          // [2 CP]   scaled pivot row of G, then of S
    cd* pinv = rowbuf + 2 * CP;                            // [CP]     1 / pivot of step k
    unsigned* key = reinterpret_cast<unsigned*>(pinv + CP);   // [CP]  pivot-search keys of column k (0: row not a candidate)
    int* prow = reinterpret_cast<int*>(key + 2 * CP);      // [CP]     pivot row of step k
    int* pos = prow + CP;                                  // [CP]     step at which row r was the pivot
    const int64_t n = blockIdx.x, p = blockIdx.y;
    if (status[p] != 0) return;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, lane = tid & 63;
    const int E = C * C;
    cd g[Q][Q], s[Q][Q];
#pragma unroll
    for (int a = 0; a < Q; ++a)
#pragma unroll
        for (int b = 0; b < Q; ++b) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            if (r < C && c < C) {
                g[a][b] = G[((int64_t)p * E + r * C + c) * N + n];
                s[a][b] = S[((int64_t)p * E + r * C + c) * N + n];
            } else {
                g[a][b] = make_double2(r == c ? 1.0 : 0.0, 0.0);
                s[a][b] = make_double2(0.0, 0.0);
            }
        }
    mv_gj_eliminate<Q>(g, s, mult, colbuf, rowbuf, pinv, key, prow, pos, C);
    // ---- pass 2: the same elimination as column operations on Y' (held in s) ----
    for (int k = 0; k < C; ++k) {
        const int pr = prow[k];
        const cd cinv = m_conj(pinv[k]);
        cd* cb = colbuf + (k & 1) * CP;
        if (tx == (pr & 15)) {
#pragma unroll
            for (int b = 0; b < Q; ++b)
                if (tx + 16 * b == pr) {
#pragma unroll
                    for (int a = 0; a < Q; ++a) {
                        s[a][b] = m_mul(s[a][b], cinv);
                        cb[ty + 16 * a] = s[a][b];
                    }
                }
        }
        __syncthreads();
        cd mc[Q];
#pragma unroll
        for (int b = 0; b < Q; ++b) mc[b] = mult[k * CP + tx + 16 * b];        // 0 for the pivot column itself
#pragma unroll
        for (int a = 0; a < Q; ++a) {
            const cd v = cb[ty + 16 * a];
#pragma unroll
            for (int b = 0; b < Q; ++b) {
                // s -= conj(m) v
                s[a][b].x = fma(-mc[b].x, v.x, fma(-mc[b].y, v.y, s[a][b].x));
                s[a][b].y = fma(-mc[b].x, v.y, fma(mc[b].y, v.x, s[a][b].y));
            }
        }
    }
    // A[i][j] = W[pr_i][pr_j] + delta_ij: element (r, c) of the block belongs to i = pos[r], j = pos[c]
#pragma unroll
    for (int a = 0; a < Q; ++a)
#pragma unroll
        for (int b = 0; b < Q; ++b) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            if (r < C && c < C) {
                const int i = pos[r], j = pos[c];
                cd v = s[a][b];
                if (i == j) v.x += 1.0;
                A[((int64_t)p * E + i * C + j) * N + n] = v;
            }
        }
}

// out[b] = (M[b] + lam I)^-1 for natural-layout C x C matrices M[b][e] (A_mvar = (H + lam' I)^-1, connectivity.py:581-589):
// the same register-resident elimination with the identity as right-hand side; row pr_k of it holds row k of the inverse
template <int Q>
__global__ void __launch_bounds__(256, 2) m_inverse_gj(const cd* __restrict__ M, const double* __restrict__ lam,
                                                       cd* __restrict__ out, int C) {
    constexpr int CP = 16 * Q;
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* mult = reinterpret_cast<cd*>(mv_smem);
    cd* colbuf = mult + CP * CP;
    cd* rowbuf = colbuf + 2 * CP;
    cd* pinv = rowbuf + 2 * CP;
    unsigned* key = reinterpret_cast<unsigned*>(pinv + CP);
    int* prow = reinterpret_cast<int*>(key + 2 * CP);
    int* pos = prow + CP;
    const int64_t bidx = blockIdx.x;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int E = C * C;
    const double l0 = lam[0];
    cd g[Q][Q], s[Q][Q];
#pragma unroll
    for (int a = 0; a < Q; ++a)
#pragma unroll
        for (int b = 0; b < Q; ++b) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            if (r < C && c < C) {
                g[a][b] = M[bidx * E + r * C + c];
                if (r == c) g[a][b].x += l0;
            } else {
                g[a][b] = make_double2(r == c ? 1.0 : 0.0, 0.0);
            }
            s[a][b] = make_double2(r == c ? 1.0 : 0.0, 0.0);
        }
    mv_gj_eliminate<Q>(g, s, mult, colbuf, rowbuf, pinv, key, prow, pos, C);
#pragma unroll
    for (int a = 0; a < Q; ++a)
#pragma unroll
        for (int b = 0; b < Q; ++b) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            if (r < C && c < C) out[bidx * E + pos[r] * C + c] = s[a][b];
        }
}

// G <- G A+ and err = max |G_new - G| on the fp64 matrix cores: complex C x C x C product per (window, bin) as four real
// v_mfma_f64_16x16x4_f64 per 16 x 16 tile and 4-deep slice of the inner dimension; both operands staged in LDS (row
// stride CP + 1 complex: the A-operand reads walk 16 rows at one column).  Wave w owns tile rows w, w + 4, ...
typedef double mv_f64x4 __attribute__((ext_vector_type(4)));
template <int Q>
__global__ void __launch_bounds__(256) m_update_mfma(cd* __restrict__ G, const cd* __restrict__ Aplus,
                                                     const int32_t* __restrict__ status, double* __restrict__ err,
                                                     int64_t N, int C) {
    constexpr int CP = 16 * Q, LS = CP + 1;
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* Gl = reinterpret_cast<cd*>(mv_smem);
    cd* Al = Gl + CP * LS;
    __shared__ double red[4];
    const int64_t n = blockIdx.x, p = blockIdx.y;
    if (status[p] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = C * C;
    for (int e = tid; e < CP * CP; e += 256) {
        const int i = e / CP, j = e - i * CP;
        const bool in = i < C && j < C;
        Gl[i * LS + j] = in ? G[((int64_t)p * E + i * C + j) * N + n] : make_double2(0.0, 0.0);
        Al[i * LS + j] = in ? Aplus[((int64_t)p * E + i * C + j) * N + n] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const int li = lane & 15, lk = lane >> 4;
    double emax = 0.0;
    for (int ti = wave; ti < Q; ti += 4) {
        mv_f64x4 re[Q], im[Q];
#pragma unroll
        for (int tj = 0; tj < Q; ++tj) { re[tj] = (mv_f64x4){0.0, 0.0, 0.0, 0.0}; im[tj] = re[tj]; }
        for (int kk = 0; kk < CP / 4; ++kk) {
            const cd a = Gl[(16 * ti + li) * LS + 4 * kk + lk];         // A operand: row li, inner index lk
#pragma unroll
            for (int tj = 0; tj < Q; ++tj) {
                const cd b = Al[(4 * kk + lk) * LS + 16 * tj + li];     // B operand: inner index lk, column li
                re[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, re[tj], 0, 0, 0);
                re[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, b.y, re[tj], 0, 0, 0);
                im[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.y, im[tj], 0, 0, 0);
                im[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, im[tj], 0, 0, 0);
            }
        }
        // C/D of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int tj = 0; tj < Q; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + lk + 4 * r, j = 16 * tj + li;
                if (i < C && j < C) {
                    const cd old = Gl[i * LS + j];
                    emax = fmax(emax, hypot(re[tj][r] - old.x, im[tj][r] - old.y));
                    G[((int64_t)p * E + i * C + j) * N + n] = make_double2(re[tj][r], im[tj][r]);
                }
            }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) emax = fmax(emax, __shfl_xor(emax, off));
    if (lane == 0) red[wave] = emax;
    __syncthreads();
    if (tid == 0) {
        const double e4 = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        if (e4 > 0.0) mv_atomic_max_nonneg(err + p, e4);
    }
}

// ---- systems of 65 ... 128 signals ---------------------------------------------------------------------------------
// [G | S] of 128 channels is 512 KB -- the whole register file of a CU -- so the augmented elimination of m_predict_gj
// stops at 64.  Beyond, the prediction step is written as A = G^-1 S G^-H + I (S is Hermitian, so (G^-1 S)^H = S G^-H)
// with an EXPLICIT inverse: G alone (256 KB at 128 channels: half the register file, one workgroup per CU) is inverted
// in place in registers by the same pivoted Gauss-Jordan, and the three C x C x C products of an iteration
// (G^-1 S, (.) G^-H, G A+) run on the fp64 matrix cores from K-slices staged in LDS (m_gemm_mfma).  The Wilson
// iteration is a fixed-point iteration -- it corrects the rounding of the explicit inverse like any other perturbation.
struct MvMat {
    cd* p;
    int64_t sp, sn, se;      // element e = i C + j of problem (window p, bin n) lives at p * sp + n * sn + e * se
};
__device__ __forceinline__ int64_t mv_at(const MvMat& m, int64_t p, int64_t n) { return p * m.sp + n * m.sn; }
// Bin of workgroup blockIdx.x.  In the series layout the eight bins 8 m ... 8 m + 7 of an element share one 128-byte line;
// workgroups are dealt to the eight XCDs round-robin, so with n = blockIdx.x every XCD would pull every line into its
// own L2 for 16 of its bytes.  XCD x takes the contiguous bins [x N / 8, (x + 1) N / 8) instead.
__device__ __forceinline__ int64_t mv_bin_of_block() {
    const unsigned bx = blockIdx.x, N = gridDim.x;
    return (N & 63u) == 0 ? (int64_t)(bx & 7u) * (N >> 3) + (bx >> 3) : (int64_t)bx;
}

// Out = (M + lam I)^-1, in place in registers: 512 threads form a 32 x 16 grid, thread (ty, tx) owns rows ty + 32 a,
// columns tx + 16 b (16 complex elements of a 128 x 128 matrix: 128 registers, two waves per SIMD).  Step k: column k is published and the pivot row chosen as in mv_gj_eliminate (no row ever moves); the
// pivot row is scaled by 1 / pivot, its column-k slot takes 1 / pivot itself and the column-k slots of the other rows
// are cleared before the rank-1 update, so that slot k ends up holding column pr_k of the accumulated row operations E
// (E M = P, P[pr_k][k] = 1): M^-1[k][pr_j] = slot[pr_k][j].
template <int Q>
__global__ void __launch_bounds__(512) m_inverse_inplace(MvMat M, const double* __restrict__ lam, MvMat Out,
                                                         const int32_t* __restrict__ status, int C) {
    constexpr int CP = 16 * Q, RB = CP <= 64 ? 64 : 128, TY = 32, RA = (CP + TY - 1) / TY;
    __shared__ cd colbuf[2][RA * TY];
    __shared__ cd rowbuf[CP];
    __shared__ unsigned key[RB];
    __shared__ int prow[CP], pos[RA * TY];
    const int64_t n = mv_bin_of_block(), p = blockIdx.y;
    if (status && status[p] != 0) return;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, lane = tid & 63;
    const cd* src = M.p + mv_at(M, p, n);
    const double l0 = lam ? lam[0] : 0.0;
    cd g[RA][Q];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < Q; ++b) {
            const int r = ty + TY * a, c = tx + 16 * b;
            if (r < C && c < C) {
                g[a][b] = src[(int64_t)(r * C + c) * M.se];
                if (r == c) g[a][b].x += l0;
            } else {
                g[a][b] = make_double2(r == c ? 1.0 : 0.0, 0.0);
            }
        }
    unsigned used = 0;
    if (tid < CP) { pos[tid] = tid < C ? tid : 0; prow[tid] = tid < C ? tid : 0; }
    if (tid < RB) key[tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < Q; ++kb) {
        for (int kx = 0; kx < 16; ++kx) {
            const int k = 16 * kb + kx;
            if (k >= C) break;
            cd* cb = colbuf[k & 1];
            if (tx == kx) {
#pragma unroll
                for (int a = 0; a < RA; ++a) {
                    const int r = ty + TY * a;
                    const cd v = g[a][kb];
                    cb[r] = v;
                    const unsigned hi = (unsigned)(__double_as_longlong(v.x * v.x + v.y * v.y) >> 32);
                    if (r < RB)
                        key[r] = (((used >> a) & 1u) || r >= C) ? 0u
                                 : ((hi & ~(unsigned)(2 * RB - 1)) | (unsigned)RB | (unsigned)(RB - 1 - r));
                }
            }
            __syncthreads();
            unsigned kv = key[lane];
            if constexpr (RB > 64) kv = max(kv, key[lane + 64]);
            const int pr = RB - 1 - (int)(mv_wave_max_u32(kv) & (unsigned)(RB - 1));
            if (ty == (pr & (TY - 1))) {       // (the reciprocal is the owners' business: one wave in eight pays for it)
                const cd piv = cb[pr];
                const double pden = piv.x * piv.x + piv.y * piv.y;
                const cd inv = make_double2(piv.x / pden, -piv.y / pden);
#pragma unroll
                for (int a = 0; a < RA; ++a)
                    if (ty + TY * a == pr) {
#pragma unroll
                        for (int b = 0; b < Q; ++b) {
                            g[a][b] = (b == kb && tx == kx) ? inv : m_mul(g[a][b], inv);
                            rowbuf[tx + 16 * b] = g[a][b];
                        }
                        used |= 1u << a;
                    }
            }
            if (tid == 0) { prow[k] = pr; pos[pr] = k; }
            __syncthreads();
            cd w[Q];
#pragma unroll
            for (int b = 0; b < Q; ++b) w[b] = rowbuf[tx + 16 * b];
#pragma unroll
            for (int a = 0; a < RA; ++a) {
                const int r = ty + TY * a;
                cd m = cb[r];
                if (r == pr) m = make_double2(0.0, 0.0);
                else if (tx == kx) g[a][kb] = make_double2(0.0, 0.0);
#pragma unroll
                for (int b = 0; b < Q; ++b) {
                    g[a][b].x = fma(-m.x, w[b].x, fma(m.y, w[b].y, g[a][b].x));
                    g[a][b].y = fma(-m.x, w[b].y, fma(-m.y, w[b].x, g[a][b].y));
                }
            }
        }
    }
    __syncthreads();
    cd* dst = Out.p + mv_at(Out, p, n);
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < Q; ++b) {
            const int r = ty + TY * a, c = tx + 16 * b;
            if (r < C && c < C) dst[(int64_t)(pos[r] * C + prow[c]) * Out.se] = g[a][b];
        }
}

// Out = (M + lam I)^-1 for 65 ... 128 signals, third generation (round 6): the same pivoted Gauss-Jordan, sixteen pivots at a
// time.  m_inverse_inplace pays two workgroup barriers, a pivot search and a division for every rank-1 update of the whole
// matrix: 128 dependent steps of ~3 800 cycles where the arithmetic of a step is 1 000 (0.27 of the fp64 rate).  Here the
// matrix sits in the registers of the workgroup as matrix-core accumulator tiles (wave w: tile row w, every tile column,
// (re, im) x 4 doubles per tile and lane -- the layout m_gemm_mfma produces), and one PANEL of sixteen columns at a time
// goes through the dependent steps:
//   1. tile column kb of every wave -> LDS; thread (row r, column quad g) keeps four panel entries of its row in registers;
//   2. sixteen pivot steps on the 16 Q x 16 panel alone, ONE barrier each: after its update every thread publishes its
//      four entries (and the search key of the next column), so that after the barrier any thread finds the pivot row
//      (search over the rows not used yet, no row ever moves), reads the pivot, its own multiplier and the four pivot-row
//      entries it needs, and takes the reciprocal itself (v_rcp_f64 + two Newton steps: a division would be a third of
//      the dependent chain) -- the in-place trick of m_inverse_inplace: the pivot's slot takes 1 / pivot, the slot of
//      row r takes -m_r / pivot; afterwards column j of the panel is column pr_j of the product E of the sixteen row operations;
//   3. the sixteen pivot rows (old values, all columns) leave the accumulators for LDS and are cleared there;
//   4. W <- W + E[:, pivots] R on the matrix cores: 4 x (Q - 1) x 4 v_mfma_f64_16x16x4 per wave, A operand = the panel,
//      B operand = the pivot rows, both from LDS; tile column kb takes the panel itself.
// 19 barriers per panel instead of 32 per sixteen whole-matrix updates, and the O(C^3) work without a dependent step.
// M^-1[k][pr_j] = W[pr_k][j] at the end, as before.  64 Q threads, ~105 KB of LDS, one workgroup per CU.
__device__ __forceinline__ double mv_rcp(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    return fma(fma(-d, x, 1.0), x, x);
}
template <int Q>
__global__ void __launch_bounds__(64 * Q) m_inverse_mfma(MvMat M, const double* __restrict__ lam, MvMat Out,
                                                         const int32_t* __restrict__ status, int C, int dbg) {
    constexpr int CP = 16 * Q, NT = 64 * Q, LSA = 17, LSR = CP + 1;
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* Pn = reinterpret_cast<cd*>(mv_smem);          // [2][CP][LSA]  the panel before an even / odd pivot step (the last one: the A operand)
    cd* Rr = Pn + 2 * CP * LSA;                       // [16][LSR]     the pivot rows of the panel (values before the update)
    __shared__ unsigned key[2][128];
    __shared__ int prow[CP], pos[CP], ppos[CP];
    const int64_t n = mv_bin_of_block(), p = blockIdx.y;
    if (status && status[p] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr_row = tid % CP, pg = tid / CP;       // panel role: row, column quad
    const cd* src = M.p + mv_at(M, p, n);
    const double l0 = lam ? lam[0] : 0.0;
    mv_f64x4 re[Q], im[Q];
#pragma unroll
    for (int tj = 0; tj < Q; ++tj)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int r = 16 * wave + lk + 4 * r4, c = 16 * tj + li;
            cd v = make_double2(r == c ? 1.0 : 0.0, 0.0);
            if (r < C && c < C) {
                v = src[(int64_t)(r * C + c) * M.se];
                if (r == c) v.x += l0;
            }
            re[tj][r4] = v.x; im[tj][r4] = v.y;
        }
    if (tid < CP) { pos[tid] = tid < C ? tid : 0; prow[tid] = tid < C ? tid : 0; }
    if (tid < 128) { key[0][tid] = 0u; key[1][tid] = 0u; }
    for (int idx = tid; idx < 16 * LSR; idx += NT) Rr[idx] = make_double2(0.0, 0.0);
    bool used = false;                                 // row pr_row has been a pivot (the four threads of a row agree)
    const int Cr = (C + 15) & ~15;
    auto keyof = [&](cd v) -> unsigned {
        const unsigned hi = (unsigned)(__double_as_longlong(v.x * v.x + v.y * v.y) >> 32);
        return (used || pr_row >= C) ? 0u : ((hi & ~255u) | 128u | (unsigned)(127 - pr_row));
    };
#pragma unroll 1
    for (int kb = 0; 16 * kb < C; ++kb) {
        const int nb = C - 16 * kb < 16 ? C - 16 * kb : 16;
        __syncthreads();                               // (the panel and the pivot rows of the last block are consumed)
#pragma unroll
        for (int tj = 0; tj < Q; ++tj)
            if (tj == kb) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    Pn[(16 * wave + lk + 4 * r4) * LSA + li] = make_double2(re[tj][r4], im[tj][r4]);
            }
        if (tid < CP) ppos[tid] = -1;
        __syncthreads();
        cd pe[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) pe[c] = Pn[pr_row * LSA + 4 * pg + c];
        if (pg == 0) key[0][pr_row] = keyof(pe[0]);
        // (step 0 reads the panel from buffer 0 as the extraction left it; the keys need one more barrier)
#pragma unroll 1
        for (int jg = 0; jg < 4; ++jg) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * jg + jj;
                if (j < nb && !(dbg & 2)) {
                    const cd* cur = Pn + (j & 1) * CP * LSA;
                    cd* nxt = Pn + ((j + 1) & 1) * CP * LSA;
                    __syncthreads();
                    const unsigned* kc = key[j & 1];
                    const int pr = 127 - (int)(mv_wave_max_u32(max(kc[lane], kc[lane + 64])) & 127u);
                    const cd piv = cur[pr * LSA + j];
                    const cd m = cur[pr_row * LSA + j];
                    cd w[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[c] = cur[pr * LSA + 4 * pg + c];
                    const double d = mv_rcp(piv.x * piv.x + piv.y * piv.y);
                    const cd inv = make_double2(piv.x * d, -piv.y * d);
                    if (pr_row == pr) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) pe[c] = (pg == jg && c == jj) ? inv : m_mul(w[c], inv);
                        used = true;
                        if (pg == 0) { prow[16 * kb + j] = pr; pos[pr] = 16 * kb + j; ppos[pr] = j; }
                    } else {
                        const cd l = m_mul(m, inv);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (pg == jg && c == jj) { pe[c] = make_double2(-l.x, -l.y); continue; }
                            pe[c].x = fma(-l.x, w[c].x, fma(l.y, w[c].y, pe[c].x));
                            pe[c].y = fma(-l.x, w[c].y, fma(-l.y, w[c].x, pe[c].y));
                        }
                    }
                    // publish: the panel before step j + 1 (after the last step: the A operand, columns past a short panel zero)
                    const bool last = j + 1 == nb;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        nxt[pr_row * LSA + 4 * pg + c] = (last && 4 * pg + c >= nb) ? make_double2(0.0, 0.0) : pe[c];
                    if (jj < 3) { if (pg == jg) key[(j + 1) & 1][pr_row] = keyof(pe[(jj + 1) & 3]); }
                    else if (pg == jg + 1) key[(j + 1) & 1][pr_row] = keyof(pe[0]);
                }
            }
        }
        __syncthreads();                               // the panel is final (buffer nb & 1), the pivots are known
        const cd* Pa = Pn + (nb & 1) * CP * LSA;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int jp = ppos[16 * wave + lk + 4 * r4];
            if (jp >= 0) {
#pragma unroll
                for (int tj = 0; tj < Q; ++tj) {
                    Rr[jp * LSR + 16 * tj + li] = make_double2(re[tj][r4], im[tj][r4]);
                    re[tj][r4] = 0.0; im[tj][r4] = 0.0;
                }
            }
        }
        __syncthreads();
        if (16 * wave < Cr && !(dbg & 1)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const cd av = Pa[(16 * wave + li) * LSA + 4 * kk + lk];
#pragma unroll
                for (int tj = 0; tj < Q; ++tj) {
                    if (tj == kb || 16 * tj >= Cr) continue;
                    const cd b = Rr[(4 * kk + lk) * LSR + 16 * tj + li];
                    re[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b.x, re[tj], 0, 0, 0);
                    re[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av.y, b.y, re[tj], 0, 0, 0);
                    im[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b.y, im[tj], 0, 0, 0);
                    im[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, b.x, im[tj], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int tj = 0; tj < Q; ++tj)
            if (tj == kb) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const cd v = Pa[(16 * wave + lk + 4 * r4) * LSA + li];
                    re[tj][r4] = v.x; im[tj][r4] = v.y;
                }
            }
    }
    __syncthreads();
    cd* dst = Out.p + mv_at(Out, p, n);
    // (tried: rows through an LDS patch so that they leave as whole lines -- 1.098 against 1.087 ms, the store is not what the kernel
    //  waits for; the pivot steps on one wave per SIMD with eight entries a thread -- 0.76 against 0.57 ms for the steps, the same
    //  1.10 in total: the steps are bound by the LDS traffic of republishing the panel, not by instruction issue;
    //  profiles/r06_mvar_inverse_ab.txt)
#pragma unroll
    for (int tj = 0; tj < Q; ++tj)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int r = 16 * wave + lk + 4 * r4, c = 16 * tj + li;
            if (r < C && c < C) dst[(int64_t)(pos[r] * C + prow[c]) * Out.se] = make_double2(re[tj][r4], im[tj][r4]);
        }
}

// Out = (M + lam I)^-1 for 129 ... 256 signals: the matrix (1 MB at 256) lives in a global scratch W (one C x C row-major
// matrix per (window, bin)) and is inverted in place by Gauss-Jordan with partial pivoting, sixteen pivots at a time so that
// the matrix crosses the memory system once per PANEL instead of once per pivot:
//   1. the panel's sixteen columns go to LDS and are eliminated there (pivot search over the rows not used yet, no row ever
//      moves; the in-place trick of m_inverse_inplace: the pivot's column slot takes 1 / pivot, the other rows' slots are
//      cleared before the update) -- afterwards column j of the panel holds column pr_j of the product E of the sixteen
//      row operations;
//   2. the sixteen pivot rows R = W[pr_j][:] (old values) go to LDS;
//   3. every other column c:  W[r][c] <- (r is one of the pivot rows ? 0 : W[r][c]) + sum_j E[r][j] R[j][c]
//      (E M = M + (E - I)[:, pivot rows] M[pivot rows, :]): round 6: 16 x 16 tiles on the fp64 matrix cores;
//   4. the panel is written back.
// M^-1[k][pr_j] = W[pr_k][j] at the end.  1024 threads, 133 KB of LDS: one workgroup per CU.
// RMAX = 256 with panels of PB = 16 columns (144 KB of LDS), RMAX = 512 with panels of 8 (156 KB + 7 KB static): the limit of the full
// factorisation (sc_mvar_max_signals).
template <int PB, int RMAX>
__global__ void __launch_bounds__(1024) m_inverse_global(MvMat M, const double* __restrict__ lam, MvMat Out, cd* Work,
                                                         const int32_t* __restrict__ status, int C) {
    extern __shared__ __align__(16) unsigned char mv_smem[];
    constexpr int PBS = PB + 1, RS = RMAX + 1;        // (odd strides: the matrix-core operand reads walk rows of Pn and columns of Rr)
    cd* Pn = reinterpret_cast<cd*>(mv_smem);          // [RMAX][PBS]    the panel
    cd* Rr = Pn + RMAX * PBS;                         // [PB][RS]       the pivot rows
    cd* colbuf = Rr + PB * RS;                        // [2][RMAX]      column j of the panel before step j (even / odd j)
    cd* rowbuf = colbuf + 2 * RMAX;                   // [PB]           the scaled pivot row of step j
    __shared__ unsigned key[RMAX];
    __shared__ int prow[RMAX], pos[RMAX];
    __shared__ unsigned char used[RMAX];
    const int64_t n = mv_bin_of_block(), p = blockIdx.y;
    if (status && status[p] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int E = C * C;
    const cd* src = M.p + mv_at(M, p, n);
    cd* W = Work + ((int64_t)p * gridDim.x + n) * E;
    const double l0 = lam ? lam[0] : 0.0;
    for (int e = tid; e < E; e += 1024) {
        cd v = src[(int64_t)e * M.se];
        if (e / C == e % C) v.x += l0;
        W[e] = v;
    }
    if (tid < RMAX) { used[tid] = 0; prow[tid] = 0; pos[tid] = 0; key[tid] = 0u; }
    __syncthreads();
    for (int k0 = 0; k0 < C; k0 += PB) {
        const int nb = C - k0 < PB ? C - k0 : PB;
        const int Cr = (C + 15) & ~15;
        // The panel in registers: thread (row, column quad) keeps four entries (RMAX rows x PB / 4 quads = the 1024 threads); a pivot
        // step publishes column j, finds the pivot row, has its four owners publish the scaled row, and updates from registers: two
        // barriers and ~0.7 KB of LDS traffic per wave where the LDS-resident form had three and 16 accesses per thread (round 6).
        constexpr int NQ = PB / 4;
        static_assert(RMAX * NQ == 1024, "one thread per (row, column quad)");
        const int prw = tid % RMAX, pq = tid / RMAX;
        cd pe[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            pe[c] = (prw < C && 4 * pq + c < nb) ? W[prw * C + k0 + 4 * pq + c] : make_double2(0.0, 0.0);
        bool row_used = prw < C && used[prw];
#pragma unroll 1
        for (int jq = 0; jq < NQ; ++jq) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = 4 * jq + jj;
                if (j < nb) {
                    cd* cb = colbuf + (j & 1) * RMAX;
                    if (pq == jq) {
                        const cd v = pe[jj];
                        cb[prw] = v;
                        const unsigned hi = (unsigned)(__double_as_longlong(v.x * v.x + v.y * v.y) >> 32);
                        key[prw] = (prw >= C || row_used) ? 0u : ((hi & ~(unsigned)(2 * RMAX - 1)) | (unsigned)RMAX | (unsigned)(RMAX - 1 - prw));
                    }
                    __syncthreads();
                    unsigned kv = 0u;
#pragma unroll
                    for (int q = 0; q < RMAX / 64; ++q) kv = max(kv, key[lane + 64 * q]);
                    const int pr = RMAX - 1 - (int)(mv_wave_max_u32(kv) & (unsigned)(RMAX - 1));
                    if (prw == pr) {
                        const cd piv = cb[pr];
                        const double pden = piv.x * piv.x + piv.y * piv.y;
                        const cd inv = make_double2(piv.x / pden, -piv.y / pden);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            pe[c] = (pq == jq && c == jj) ? inv : m_mul(pe[c], inv);
                            rowbuf[4 * pq + c] = pe[c];
                        }
                        row_used = true;
                        if (pq == 0) { prow[k0 + j] = pr; pos[pr] = k0 + j; used[pr] = 1; }
                    }
                    __syncthreads();
                    if (prw != pr) {
                        const cd m = cb[prw];
                        if (pq == jq) pe[jj] = make_double2(0.0, 0.0);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const cd w = rowbuf[4 * pq + c];
                            pe[c].x = fma(-m.x, w.x, fma(m.y, w.y, pe[c].x));
                            pe[c].y = fma(-m.x, w.y, fma(-m.y, w.x, pe[c].y));
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) Pn[prw * PBS + 4 * pq + c] = pe[c];          // (rows >= C and columns >= nb: zeros)
        __syncthreads();
        for (int idx = tid; idx < PB * Cr; idx += 1024) {
            const int j = idx / Cr, c = idx - j * Cr;
            Rr[j * RS + c] = (j < nb && c < C) ? W[prow[k0 + j] * C + c] : make_double2(0.0, 0.0);
        }
        __syncthreads();
        {   // W <- (pivot row of this panel ? 0 : W) + E R on the matrix cores (round 6; the VALU form read LDS six times for eight complex
            // FMAs and ran at a quarter of the fp64 rate): 16 x 16 tiles dealt over the 16 waves, two in flight per wave, A operand = the
            // panel, B operand = the pivot rows; a panel of 16 columns is a whole tile column and is skipped, one of 8 is masked on the store
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
            const int nt = Cr >> 4, ntt = nt * nt;
            const int tskip = PB == 16 ? (k0 >> 4) : -1;
            for (int t0 = wave; t0 < ntt; t0 += 32) {
                mv_f64x4 re[2], im[2];
                int ti[2], tj[2];
                bool on[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + 16 * u;
                    ti[u] = t / nt; tj[u] = t - ti[u] * nt;
                    on[u] = t < ntt && tj[u] != tskip;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int r = 16 * ti[u] + lk + 4 * r4, c = 16 * tj[u] + li;
                        cd v = make_double2(0.0, 0.0);
                        if (on[u] && r < C && c < C && !(used[r] && pos[r] >= k0)) v = W[r * C + c];
                        re[u][r4] = v.x; im[u][r4] = v.y;
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (!on[u]) continue;
#pragma unroll
                    for (int kk = 0; kk < PB / 4; ++kk) {
                        const cd av = Pn[(16 * ti[u] + li) * PBS + 4 * kk + lk];
                        const cd bv = Rr[(4 * kk + lk) * RS + 16 * tj[u] + li];
                        re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.x, re[u], 0, 0, 0);
                        re[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av.y, bv.y, re[u], 0, 0, 0);
                        im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, bv.y, im[u], 0, 0, 0);
                        im[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, bv.x, im[u], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (!on[u]) continue;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int r = 16 * ti[u] + lk + 4 * r4, c = 16 * tj[u] + li;
                        if (r < C && c < C && !(c >= k0 && c < k0 + nb)) W[r * C + c] = make_double2(re[u][r4], im[u][r4]);
                    }
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < C * PB; idx += 1024) {
            const int r = idx / PB, j = idx % PB;
            if (j < nb) W[r * C + k0 + j] = Pn[r * PBS + j];
        }
        __syncthreads();
    }
    cd* dst = Out.p + mv_at(Out, p, n);
    for (int e = tid; e < E; e += 1024) {
        const int r = e / C, c = e - r * C;
        dst[(int64_t)(pos[r] * C + prow[c]) * Out.se] = W[e];
    }
}

// O = X Y (BH: X Y^H; ADD_I: + I; ERR: err[p] = max |O - X| elementwise, O may alias X) per (window, bin), C <= 16 Q <=
// 128, on the fp64 matrix cores.  K-slices of 16: X[:, k0 : k0 + 16] and Y[k0 : k0 + 16, :] wait in LDS (68 KB) while the
// next slice travels HBM / L2 -> registers; 512 threads, wave w owns tile row w and every tile column of it: 8 tiles x
// (re, im) x 4 = 64 accumulator doubles per lane at 128 channels, two waves per SIMD -- 32 independent MFMAs per 4-deep
// step and wave.
// Beyond 128 signals (Q = 8) the output is cut into 128 x 128 blocks, one workgroup each (blockIdx.z = 2 block row + block
// column; every block runs the whole K range): O must then not alias X, and a frozen window is copied across so that the
// caller can swap the two buffers.
template <int Q, bool BH, bool ADD_I, bool ERR>
__global__ void __launch_bounds__(512) m_gemm_mfma(MvMat X, MvMat Y, MvMat O, const int32_t* __restrict__ status,
                                                   double* __restrict__ err, int C) {
    static_assert(Q % 2 == 0 && Q <= 8, "even tile count, one tile row per wave");
    constexpr int CP = 16 * Q, KS = 16, LSX = KS + 1, LSY = CP + 1, NU = Q / 2;
    extern __shared__ __align__(16) unsigned char mv_smem[];
    cd* Xs = reinterpret_cast<cd*>(mv_smem);      // [CP][LSX]
    cd* Ys = Xs + CP * LSX;                       // [KS][LSY]
    __shared__ double red[8];
    const int64_t n = mv_bin_of_block(), p = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const cd* xb = X.p + mv_at(X, p, n);
    const cd* yb = Y.p + mv_at(Y, p, n);
    const int nbk = gridDim.z == 1 ? 1 : (C + CP - 1) / CP;                          // the output cut into nbk x nbk blocks
    const int i0 = (int)(blockIdx.z / nbk) * CP, j0 = (int)(blockIdx.z % nbk) * CP;   // output block (0, 0 up to 128 signals)
    // X Y^H + I is the prediction step's G^-1 S G^-H + I: Hermitian.  Tiles below the diagonal are not computed but written as
    // the mirror images of the tiles above (the lower-left block of a cut output: by the workgroup of the upper-right one), and
    // the tile rows are dealt so that the two waves of a SIMD (w, w + 4) share 9 of the 36 tiles: rows w and Q + 3 - w.
    constexpr bool HERM = BH && ADD_I;
    if (HERM && i0 > j0) return;
    const bool diag = i0 == j0;
    const int wrow = (HERM && wave >= 4) ? Q + 3 - wave : wave;
    if (status && status[p] != 0) {
        if (ERR && gridDim.z > 1) {              // frozen window, blocked launch: O <- X for this block
            cd* ob = O.p + mv_at(O, p, n);
            for (int idx = tid; idx < CP * CP; idx += 512) {
                const int i = i0 + idx / CP, j = j0 + idx % CP;
                if (i < C && j < C) ob[(int64_t)(i * C + j) * O.se] = xb[(int64_t)(i * C + j) * X.se];
            }
        }
        return;
    }
    cd rx[NU], ry[NU];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = tid + 512 * u;
            {
                const int i = i0 + (idx >> 4), k = k0 + (idx & 15);
                rx[u] = (i < C && k < C) ? xb[(int64_t)(i * C + k) * X.se] : make_double2(0.0, 0.0);
            }
            if constexpr (BH) {
                const int j = j0 + (idx >> 4), k = k0 + (idx & 15);
                ry[u] = (j < C && k < C) ? yb[(int64_t)(j * C + k) * Y.se] : make_double2(0.0, 0.0);
            } else {
                const int kr = idx / CP, j = j0 + idx - kr * CP, k = k0 + kr;
                ry[u] = (k < C && j < C) ? yb[(int64_t)(k * C + j) * Y.se] : make_double2(0.0, 0.0);
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = tid + 512 * u;
            Xs[(idx >> 4) * LSX + (idx & 15)] = rx[u];
            if constexpr (BH) Ys[(idx & 15) * LSY + (idx >> 4)] = m_conj(ry[u]);
            else { const int kr = idx / CP; Ys[kr * LSY + (idx - kr * CP)] = ry[u]; }
        }
    };
    mv_f64x4 re[Q], im[Q];
#pragma unroll
    for (int tj = 0; tj < Q; ++tj) { re[tj] = (mv_f64x4){0.0, 0.0, 0.0, 0.0}; im[tj] = re[tj]; }
    const int li = lane & 15, lk = lane >> 4;
    const int Cr = (C + 15) & ~15;
    // Tiles that hold signals only: a block of the cut output (and any system that is not a multiple of 128 wide) multiplies the
    // tile rows / columns below C and nothing else -- at 130 signals the four 128 x 128 blocks were 4 x the arithmetic of 128 signals
    // for 3 % more entries (round 5: 2.0-2.3 ms per product against 0.5-0.7; profiles/r06_mvar_kernels.txt).  Wave-uniform tests.
    const int ntj = (C - j0 + 15) / 16 < Q ? (C - j0 + 15) / 16 : Q;
    const bool mine = wave < Q && i0 + 16 * wrow < C;
    fetch(0);
    park();
    __syncthreads();
#pragma unroll 1
    for (int k0 = 0; k0 < Cr; k0 += KS) {
        const bool more = k0 + KS < Cr;
        if (more) fetch(k0 + KS);
        if (mine) {
#pragma unroll
            for (int kk = 0; kk < KS / 4; ++kk) {
                const cd av = Xs[(16 * wrow + li) * LSX + 4 * kk + lk];
#pragma unroll
                for (int tj = 0; tj < Q; ++tj) {
                    if (tj >= ntj || (HERM && diag && tj < wrow)) continue;
                    const cd b = Ys[(4 * kk + lk) * LSY + 16 * tj + li];
                    re[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b.x, re[tj], 0, 0, 0);
                    re[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av.y, b.y, re[tj], 0, 0, 0);
                    im[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, b.y, im[tj], 0, 0, 0);
                    im[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, b.x, im[tj], 0, 0, 0);
                }
            }
        }
        __syncthreads();                 // slice consumed
        if (more) { park(); __syncthreads(); }
    }
    cd* ob = O.p + mv_at(O, p, n);
    double emax = 0.0;
    if (mine) {
#pragma unroll
        for (int tj = 0; tj < Q; ++tj) {
            if (HERM && diag && tj < wrow) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 16 * wrow + lk + 4 * r, j = j0 + 16 * tj + li;
                if (i < C && j < C) {
                    cd v = make_double2(re[tj][r], im[tj][r]);
                    if (ADD_I && i == j) v.x += 1.0;
                    if constexpr (ERR) {
                        const cd old = xb[(int64_t)(i * C + j) * X.se];
                        emax = fmax(emax, hypot(v.x - old.x, v.y - old.y));
                    }
                    ob[(int64_t)(i * C + j) * O.se] = v;
                    if (HERM && (!diag || tj > wrow)) ob[(int64_t)(j * C + i) * O.se] = m_conj(v);
                }
            }
        }
    }
    if constexpr (ERR) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) emax = fmax(emax, __shfl_xor(emax, off));
        if (lane == 0) red[wave] = emax;
        __syncthreads();
        if (tid == 0) {
            double e8 = red[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) e8 = fmax(e8, red[q]);
            if (e8 > 0.0) mv_atomic_max_nonneg(err + p, e8);
        }
    }
}

// small helpers of the measures beyond 64 signals
__global__ void m_real_to_cd(const double* __restrict__ a, cd* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_double2(a[i], 0.0);
}
__global__ void m_cd_to_real(const cd* __restrict__ a, double* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i].x;
}
// Sigma = H0 H0^T (connectivity.py:1703-1708): one thread per element
__global__ void m_sigma(const double* __restrict__ h0, double* __restrict__ sigma, int C) {
    const int64_t p = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x, E = C * C;
    if (e >= E) return;
    const int i = e / C, j = e % C;
    double s = 0.0;
    for (int k = 0; k < C; ++k) s += h0[p * E + i * C + k] * h0[p * E + j * C + k];
    sigma[p * E + e] = s;
}
// sq[b] = sum_e |H[b][e]|^2 in a fixed order: one block per (window, bin)
__global__ void __launch_bounds__(256) m_sumsq(const cd* __restrict__ H, double* __restrict__ sq, int E) {
    __shared__ double red[256];
    const int64_t b = blockIdx.x;
    double s2 = 0.0;
    for (int e = threadIdx.x; e < E; e += 256) { const cd v = H[b * E + e]; s2 += v.x * v.x + v.y * v.y; }
    red[threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sq[b] = red[0];
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
// H0 = Re mean_n G[n] (natural layout [p][n][e]: threads along e read unit stride); partial sums of H0^2 per block:
// grid (P, n_chunks), chunk c covers elements [256 c, 256 c + 256)
__global__ void __launch_bounds__(256) m_h0(const cd* G, double* h0, double* sq, int64_t N, int E) {
    __shared__ double red[256];
    const int64_t p = blockIdx.x;
    const int e = blockIdx.y * 256 + threadIdx.x;
    double s2 = 0.0;
    if (e < E) {
        double a = 0.0;
        for (int64_t n = 0; n < N; ++n) a += G[(p * N + n) * E + e].x;
        a /= (double)N;
        h0[p * E + e] = a;
        s2 = a * a;
    }
    red[threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sq[p * gridDim.y + blockIdx.y] = red[0];
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
// (cache = 0, beyond 128 signals: the squared moduli do not fit LDS and are parked in `pw_global` [P F][C C] instead)
__global__ void m_measure(const cd* H, const cd* Amv, const double* sigma, const double* tot, int which,
                          double* out, int64_t F, int C, int cache, double* pw_global) {
    extern __shared__ __align__(16) unsigned char mv_smem[];
    const int64_t b = blockIdx.x, p = b / F;
    const int E = C * C;
    double* nrm = reinterpret_cast<double*>(mv_smem);                    // per-row (inflow) or per-column (outflow) sums
    double* pw = cache ? nrm + C : pw_global + b * E;                    // |H_ij|^2 or |A_ij|^2
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
#define MV_CHECK_FFT(expr)                                                                       \
    do {                                                                                         \
        rocfft_status s_ = (expr);                                                               \
        if (s_ != rocfft_status_success) {                                                       \
            sc_set_error("%s failed: rocfft_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
            rc = SC_EFFT; goto done;                                                             \
        }                                                                                        \
    } while (0)

#define MV_HIST 1024          // iterations whose "still running" counts the workspace logs (max_iterations <= this)
#define MV_POLL 4             // iterations queued between two looks at the counts

static size_t mv_gj_lds(int Q) {
    const size_t CP = 16 * (size_t)Q;
    return (CP * CP + 2 * CP + 2 * CP + CP) * sizeof(cd) + 2 * CP * sizeof(unsigned) + 2 * CP * sizeof(int) + 64;
}
static int mv_launch_predict(int Q, dim3 grid, hipStream_t st, const cd* S, const cd* G, const int32_t* status, cd* A,
                             int64_t N, int C) {
    const size_t lds = mv_gj_lds(Q);
#define MV_PRED(QQ)                                                                                          \
    case QQ:                                                                                                 \
        (void)hipFuncSetAttribute((const void*)m_predict_gj<QQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(m_predict_gj<QQ>, grid, dim3(256), lds, st, S, G, status, A, N, C);              \
        break;
    switch (Q) { MV_PRED(1) MV_PRED(2) MV_PRED(3) default: MV_PRED(4) }
#undef MV_PRED
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
static int mv_launch_update(int Q, dim3 grid, hipStream_t st, cd* G, const cd* Aplus, const int32_t* status, double* err,
                            int64_t N, int C) {
    const size_t CP = 16 * (size_t)Q;
    const size_t lds = 2 * CP * (CP + 1) * sizeof(cd);
#define MV_UPD(QQ)                                                                                           \
    case QQ:                                                                                                 \
        (void)hipFuncSetAttribute((const void*)m_update_mfma<QQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(m_update_mfma<QQ>, grid, dim3(256), lds, st, G, Aplus, status, err, N, C);       \
        break;
    switch (Q) { MV_UPD(1) MV_UPD(2) MV_UPD(3) default: MV_UPD(4) }
#undef MV_UPD
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

static MvMat mv_series(cd* b, int64_t N, int E) { return MvMat{b, (int64_t)E * N, 1, N}; }
static MvMat mv_natural(cd* b, int64_t N, int E) { return MvMat{b, N * (int64_t)E, (int64_t)E, 1}; }
static int mv_big_q(int64_t C) { return C <= 64 ? 4 : (C <= 96 ? 6 : 8); }
// Largest system of the register-resident [G | S] elimination (m_predict_gj / m_update_mfma).  Round 6: 48 -- at 49 ... 64 signals the
// explicit inverse + matrix-core products are faster (7 windows x 256 bins of 64 signals: DTF 24.7 -> 20.3 ms; at 48 signals the two
// paths tie, 13.3 ms, tools/cliff_sweep.py mvar); SC_MVAR_INVERSE=small64 keeps the round-3 boundary (A/B, tests).
static int mv_small_max(void) {
    const char* sel = sc_switch(SC_SW_MVAR_INVERSE);
    if (sel && strncmp(sel, "small", 5) == 0) {               // small64 (rounds 3-5), small32, ...: the boundary itself
        const int v = atoi(sel + 5);
        return v >= 16 && v <= MV_CSMALL ? v : MV_CSMALL;
    }
    return 48;
}

// (scratch: one C x C matrix per problem of the grid, beyond 128 signals only)
static int mv_launch_inverse_big(int64_t C, dim3 grid, hipStream_t st, MvMat M, const double* lam, MvMat Out,
                                 const int32_t* status, cd* scratch) {
    if (C > MV_CMID) {
        SC_REQUIRE(scratch, "inverse beyond 128 signals needs a scratch");
        if (C <= 256) {
            const size_t lds = (size_t)(256 * 17 + 16 * 257 + 2 * 256 + 16) * sizeof(cd);
            SC_CHECK_HIP(hipFuncSetAttribute((const void*)m_inverse_global<16, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((m_inverse_global<16, 256>), grid, dim3(1024), lds, st, M, lam, Out, scratch, status, (int)C);
        } else {
            const size_t lds = (size_t)(512 * 9 + 8 * 513 + 2 * 512 + 8) * sizeof(cd);
            SC_CHECK_HIP(hipFuncSetAttribute((const void*)m_inverse_global<8, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((m_inverse_global<8, 512>), grid, dim3(1024), lds, st, M, lam, Out, scratch, status, (int)C);
        }
        SC_CHECK_HIP(hipGetLastError());
        return SC_OK;
    }
    const char* sel = sc_switch(SC_SW_MVAR_INVERSE);      // "registers": the round-3 kernel (one whole-matrix update per pivot)
    if (sel && sel[0] == 'r' && mv_big_q(C) >= 6) {
        if (mv_big_q(C) == 6) hipLaunchKernelGGL(m_inverse_inplace<6>, grid, dim3(512), 0, st, M, lam, Out, status, (int)C);
        else hipLaunchKernelGGL(m_inverse_inplace<8>, grid, dim3(512), 0, st, M, lam, Out, status, (int)C);
    } else if (mv_big_q(C) == 4) {
        const size_t lds = (size_t)(2 * 64 * 17 + 16 * 65) * sizeof(cd);          // 51 KB: three workgroups of 256 threads per compute unit
        SC_CHECK_HIP(hipFuncSetAttribute((const void*)m_inverse_mfma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(m_inverse_mfma<4>, grid, dim3(256), lds, st, M, lam, Out, status, (int)C, 0);
    } else if (mv_big_q(C) == 6) {
        const int dbg = sel ? atoi(sel) : 0;             // (timing ablations: 1 no matrix-core updates, 2 no pivot steps)
        const size_t lds = (size_t)(2 * 96 * 17 + 16 * 97) * sizeof(cd);
        SC_CHECK_HIP(hipFuncSetAttribute((const void*)m_inverse_mfma<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(m_inverse_mfma<6>, grid, dim3(384), lds, st, M, lam, Out, status, (int)C, dbg);
    } else {
        const int dbg = sel ? atoi(sel) : 0;
        const size_t lds = (size_t)(2 * 128 * 17 + 16 * 129) * sizeof(cd);
        SC_CHECK_HIP(hipFuncSetAttribute((const void*)m_inverse_mfma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(m_inverse_mfma<8>, grid, dim3(512), lds, st, M, lam, Out, status, (int)C, dbg);
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

enum { MV_GEMM_PLAIN = 0, MV_GEMM_BH_I = 1, MV_GEMM_ERR = 2 };
template <int Q>
static int mv_launch_gemm_q(int mode, dim3 grid, hipStream_t st, MvMat X, MvMat Y, MvMat O, const int32_t* status,
                            double* err, int C) {
    constexpr size_t CP = 16 * Q;
    const size_t lds = (CP * 17 + 16 * (CP + 1)) * sizeof(cd);
#define MV_GEMM(BH, ADDI, ERRF)                                                                                     \
    do {                                                                                                            \
        auto k = m_gemm_mfma<Q, BH, ADDI, ERRF>;                                                                    \
        SC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
        hipLaunchKernelGGL(k, grid, dim3(512), lds, st, X, Y, O, status, err, C);                                   \
    } while (0)
    if (mode == MV_GEMM_PLAIN) MV_GEMM(false, false, false);
    else if (mode == MV_GEMM_BH_I) MV_GEMM(true, true, false);
    else MV_GEMM(false, false, true);
#undef MV_GEMM
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
static int mv_launch_gemm(int64_t C, int mode, dim3 grid, hipStream_t st, MvMat X, MvMat Y, MvMat O,
                          const int32_t* status, double* err) {
    if (C > MV_CMID) {          // 128 x 128 output blocks, one workgroup each; O must not alias X or Y
        SC_REQUIRE(O.p != X.p && O.p != Y.p, "blocked product in place");
        const unsigned nbk = (unsigned)((C + 127) / 128);
        grid.z = nbk * nbk;
    }
    // (<= 64 signals: the 96-wide instantiation with its tile rows / columns beyond the signals skipped -- a 64-wide one on 512 threads
    //  compiles to an accumulator shuffle of 800 moves and 420 bytes of scratch a lane, on 256 threads it is slower: DTF at 64 signals x 7
    //  windows 29.3 ms against 20.3)
    return mv_big_q(C) <= 6 ? mv_launch_gemm_q<6>(mode, grid, st, X, Y, O, status, err, (int)C)
                            : mv_launch_gemm_q<8>(mode, grid, st, X, Y, O, status, err, (int)C);
}

static int mv_threads(int C) {
    const int e = C * C;
    return e >= 256 ? 256 : (e > 128 ? 256 : (e > 64 ? 128 : 64));
}

static size_t mv_pair_lds(int C) { return (size_t)2 * C * C * sizeof(cd) + (size_t)C * sizeof(int) + 16; }

extern "C" int sc_mvar_max_signals(void) { return MV_CMAX; }

extern "C" int sc_mvar_workspace_bytes(int64_t n_groups, int64_t C, int64_t N, size_t* bytes) {
    SC_REQUIRE(bytes && n_groups >= 1 && C >= 1 && N >= 2, "bad workspace query");
    const size_t E = (size_t)C * C, P = (size_t)n_groups, F = (size_t)N / 2 + 1;
    // factor: S, G, A series (beyond 64 signals also G^-1 and G^-1 S); measures: H, A_mvar natural + small per-window arrays
    const size_t n_big = C > mv_small_max() ? 5 : 3;
    const size_t factor = n_big * P * E * (size_t)N * sizeof(cd) + P * 16 + P * E * 8 + 128 + (size_t)MV_HIST * 4;
    // (beyond 128 signals: + the scratch of the blocked inverse, which also parks |H|^2 / |A|^2 for m_measure)
    // (sq: one partial sum per (window, bin) and, before that, per (window, 256-element chunk of the matrix))
    const size_t n_sq = (F > 16 ? F : 16) > (E + 255) / 256 ? (F > 16 ? F : 16) : (E + 255) / 256;
    const size_t meas = (C > MV_CMID ? 3 : 2) * P * F * E * sizeof(cd) + P * E * 8 * 3 + P * n_sq * 8 +
                        P * (size_t)C * 8 + P * 8 + 256;
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
        sc_set_error("full Wilson factorisation: n_signals <= %d (got %lld)",
                     MV_CMAX, (long long)C);
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
    const bool big = C > mv_small_max();
    cd* Ginv = nullptr;
    cd* T = nullptr;
    if (big) {
        Ginv = (cd*)w; w += (size_t)P * E * N * sizeof(cd);
        T = (cd*)w; w += (size_t)P * E * N * sizeof(cd);
    }
    double* err = (double*)w; w += (size_t)P * 8;
    double* g0 = (double*)w; w += (size_t)P * E * 8;
    int32_t* n_fallback = (int32_t*)w; w += 64;
    int32_t* n_running = (int32_t*)w;
    const dim3 gridE((unsigned)((N + 255) / 256), (unsigned)(E < MV_GRID_Y ? E : MV_GRID_Y), (unsigned)P);
    const bool huge = C > MV_CMID;
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
        hipLaunchKernelGGL(m_build, gridE, dim3(256), 0, st, sc_rec(d_accum, planes), d, S, big ? (int64_t)E : (int64_t)1,
                           big ? (int64_t)1 : N);
    } else if (big) {
        SC_CHECK_HIP(hipMemcpyAsync(S, d_S, (size_t)P * E * N * sizeof(cd), hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(m_to_series, gridE, dim3(256), 0, st, (const cd*)d_S, S, N, E);
    }
    int rc = SC_OK;
    rocfft_plan fwd = nullptr, inv = nullptr;
    bool fwd_cached = false, inv_cached = false;
    rocfft_execution_info info = nullptr;
    void* fft_work = nullptr;
    size_t ws_f = 0, ws_i = 0;
    static int rocfft_ready = 0;
    if (!rocfft_ready) { rocfft_setup(); rocfft_ready = 1; }
    const dim3 gridB((unsigned)N, (unsigned)P);
    int iters = 0, running = (int)P, queued = 0;
    int32_t hist[MV_POLL];
    const bool fused = sc_internal_causal_fft_supported(N);
    // Beyond 64 signals A crosses the transform in the layout of the products around it where the transform has the loads for it
    const bool natA = big && fused && sc_internal_causal_fft_natural_supported(N);
    const MvMat Adesc = natA ? mv_natural(A, N, E) : mv_series(A, N, E);
    const int Q = (int)((C + 15) / 16);
    if (max_iter > MV_HIST) {
        sc_set_error("max_iterations = %d exceeds the %d iterations the workspace can log", max_iter, MV_HIST);
        return SC_EINVAL;
    }
    if (!fused) {
        if ((rc = sc_internal_z2z_plan(&fwd, 1, (size_t)N, (size_t)E * P, &fwd_cached)) != SC_OK) goto done;
        if ((rc = sc_internal_z2z_plan(&inv, 0, (size_t)N, (size_t)E * P, &inv_cached)) != SC_OK) goto done;
        MV_CHECK_FFT(rocfft_plan_get_work_buffer_size(fwd, &ws_f));
        MV_CHECK_FFT(rocfft_plan_get_work_buffer_size(inv, &ws_i));
        MV_CHECK_FFT(rocfft_execution_info_create(&info));
        if (ws_f < ws_i) ws_f = ws_i;
        if (ws_f) {
            if (hipMallocAsync(&fft_work, ws_f, st) != hipSuccess) { sc_set_error("rocFFT work buffer alloc failed"); rc = SC_ENOMEM; goto done; }
            MV_CHECK_FFT(rocfft_execution_info_set_work_buffer(info, fft_work, ws_f));
        }
        MV_CHECK_FFT(rocfft_execution_info_set_stream(info, st));
    }
    (void)hipMemsetAsync(err, 0, (size_t)P * 8, st);
    (void)hipMemsetAsync(d_n_iter, 0, (size_t)P * 4, st);
    (void)hipMemsetAsync(n_fallback, 0, 64 + (size_t)MV_HIST * 4, st);
    // G0 = chol(Re ifft_n(S)[lag 0])^H broadcast over the bins (minimum_phase_decomposition.py:48-77)
    if (big) hipLaunchKernelGGL(m_lag0_nat, dim3((unsigned)((E + 255) / 256), (unsigned)P), dim3(256), 0, st, S, g0, N, E);
    else hipLaunchKernelGGL(m_lag0, dim3((unsigned)(((int64_t)P * E + 3) / 4)), dim3(256), 0, st, S, g0, N, (int64_t)P * E);
    if (!huge) SC_CHECK_HIP(hipFuncSetAttribute((const void*)m_chol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)E * 8)));
    hipLaunchKernelGGL(m_chol, dim3((unsigned)P), dim3(256), huge ? (size_t)0 : (size_t)E * 8, st, g0, d_status, n_fallback, (int)C,
                       huge ? 1 : 0);
    hipLaunchKernelGGL(m_restart_all, dim3(64), dim3(256), 0, st, g0, n_fallback, (int64_t)P, (int)C);
    hipLaunchKernelGGL(m_restart_count, dim3(1), dim3(1), 0, st, n_fallback, (int32_t)P);
    if (big) hipLaunchKernelGGL(m_fill_nat, dim3((unsigned)((E + 255) / 256), (unsigned)N, (unsigned)P), dim3(256), 0, st, g0, G, N, E);
    else hipLaunchKernelGGL(m_fill, gridE, dim3(256), 0, st, g0, G, N, E);
    // The stream is synchronised once per MV_POLL iterations: every iteration logs how many windows are still running
    // into its own slot; converged windows are skipped by every kernel, so the iterations queued past the last
    // convergence are empty launches.
    while (queued < max_iter && running > 0) {
        const int first = queued;
        for (int b = 0; b < MV_POLL && queued < max_iter; ++b, ++queued) {
            void* bufs[1] = {A};
            if (big) {          // A = G^-1 S G^-H + I: explicit inverse, two matrix-core products
                // (G, S, G^-1 and G^-1 S in the natural layout [p][n][e]: coalesced; only A crosses the transform as series)
                // (beyond 128 signals T doubles as the scratch of the inverse: it is written only by the product after it)
                if ((rc = mv_launch_inverse_big(C, gridB, st, mv_natural(G, N, E), nullptr, mv_natural(Ginv, N, E), d_status, T)) != SC_OK) goto done;
                if ((rc = mv_launch_gemm(C, MV_GEMM_PLAIN, gridB, st, mv_natural(Ginv, N, E), mv_natural(S, N, E),
                                         mv_natural(T, N, E), d_status, nullptr)) != SC_OK) goto done;
                if ((rc = mv_launch_gemm(C, MV_GEMM_BH_I, gridB, st, mv_natural(T, N, E), mv_natural(Ginv, N, E),
                                         Adesc, d_status, nullptr)) != SC_OK) goto done;
            } else if ((rc = mv_launch_predict(Q, gridB, st, S, G, d_status, A, N, (int)C)) != SC_OK) goto done;
            if (fused) {        // ifft -> causal mask -> fft in one kernel (sc_wilson_fft.hip)
                if ((rc = natA ? sc_internal_causal_fft_pair_natural(A, d_status, P, (int)C, N, st)
                               : sc_internal_causal_fft_pair(A, d_status, P, (int)C, N, st)) != SC_OK) goto done;
            } else {
                MV_CHECK_FFT(rocfft_execute(inv, bufs, nullptr, info));
                hipLaunchKernelGGL(m_causal, gridE, dim3(256), 0, st, A, N, (int)C);
                MV_CHECK_FFT(rocfft_execute(fwd, bufs, nullptr, info));
            }
            if (huge) {         // the blocked product cannot run in place: G A+ into T (frozen windows copied), then swap
                if ((rc = mv_launch_gemm(C, MV_GEMM_ERR, gridB, st, mv_natural(G, N, E), Adesc, mv_natural(T, N, E),
                                         d_status, err)) != SC_OK) goto done;
                cd* t = G; G = T; T = t;
            } else if (big) {
                if ((rc = mv_launch_gemm(C, MV_GEMM_ERR, gridB, st, mv_natural(G, N, E), Adesc, mv_natural(G, N, E),
                                         d_status, err)) != SC_OK) goto done;
            } else if ((rc = mv_launch_update(Q, gridB, st, G, A, d_status, err, N, (int)C)) != SC_OK) goto done;
            hipLaunchKernelGGL(m_flags, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, d_status, d_n_iter, err, tol, P,
                               n_running + queued);
        }
        if (hipMemcpyAsync(hist, n_running + first, (size_t)(queued - first) * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
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
    if (big) (void)hipMemcpyAsync(d_G, G, (size_t)P * E * N * sizeof(cd), hipMemcpyDeviceToDevice, st);
    else hipLaunchKernelGGL(m_to_natural, gridE, dim3(256), 0, st, G, (cd*)d_G, N, E);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        sc_set_error("Wilson factor copy-out failed: %s", hipGetErrorString(hipGetLastError()));
        rc = SC_EHIP; goto done;
    }
    if (h_summary) {
        int fb = 0;
        (void)hipMemcpy(&fb, n_fallback, 4, hipMemcpyDeviceToHost);      // the stream was synchronised just above
        h_summary[0] = iters; h_summary[1] = running; h_summary[2] = fb;
    }
done:
    if (info) rocfft_execution_info_destroy(info);
    if (fwd && !fwd_cached) rocfft_plan_destroy(fwd);       // (cached plans live as long as the process: sc_internal_z2z_plan)
    if (inv && !inv_cached) rocfft_plan_destroy(inv);
    if (fft_work) (void)hipFreeAsync(fft_work, st);
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
        sc_set_error("MVAR measures: n_signals <= %d (got %lld)", MV_CMAX, (long long)C);
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
    const size_t n_sq = (size_t)(F > 16 ? F : 16) > ((size_t)E + 255) / 256 ? (size_t)(F > 16 ? F : 16) : ((size_t)E + 255) / 256;
    double* sq = (double*)w; w += (size_t)P * n_sq * 8;      // per (window, bin) or per (window, 256-element chunk)
    double* tot = (double*)w; w += (size_t)P * C * 8;
    double* lam = (double*)w; w += 64;                // [0] lam of H0, [1] lam' of H
    const bool huge = C > MV_CMID;
    w += (16 - ((uintptr_t)w & 15)) & 15;
    cd* scratch = huge ? (cd*)w : nullptr;            // [P][F][E]: the blocked inverse's matrices, then m_measure's squared moduli
    const int nt = mv_threads((int)C);
    const bool big = C > mv_small_max();
    const size_t lds = big ? 0 : mv_pair_lds((int)C);
    if (!big) {
        (void)hipFuncSetAttribute((const void*)m_h0_inverse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)m_transfer, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const cd* G = (const cd*)d_G;
    const int h0_chunks = (E + 255) / 256;
    hipLaunchKernelGGL(m_h0, dim3((unsigned)P, (unsigned)h0_chunks), dim3(256), 0, st, G, h0, sq, N, E);
    hipLaunchKernelGGL(m_sum, dim3(1), dim3(64), 0, st, sq, P * h0_chunks, 1e-12 / (double)(P * E), lam);
    // beyond 64 signals the LDS-resident LU kernels give way to the in-register inverse and the matrix-core product;
    // (H0 + lam I)^-1 as a complex matrix with zero imaginary part, parked in the (still unused) A_mvar buffer
    cd* h0c = Amv;
    cd* hinvc = Amv + (size_t)P * E;
    if (big) {
        const unsigned nb = (unsigned)(((int64_t)P * E + 255) / 256);
        hipLaunchKernelGGL(m_real_to_cd, dim3(nb), dim3(256), 0, st, h0, h0c, (int64_t)P * E);
        int rcb = mv_launch_inverse_big(C, dim3(1, (unsigned)P), st, MvMat{h0c, (int64_t)E, 0, 1}, lam,
                                        MvMat{hinvc, (int64_t)E, 0, 1}, nullptr, scratch);
        if (rcb != SC_OK) return rcb;
        hipLaunchKernelGGL(m_cd_to_real, dim3(nb), dim3(256), 0, st, hinvc, hinv, (int64_t)P * E);
        hipLaunchKernelGGL(m_sigma, dim3((unsigned)((E + 255) / 256), (unsigned)P), dim3(256), 0, st, h0, sigma, (int)C);
    } else {
        hipLaunchKernelGGL(m_h0_inverse, dim3((unsigned)P), dim3(nt), lds, st, h0, lam, hinv, sigma, (int)C);
    }
    if (which == SC_MVAR_NOISE_COVARIANCE) {
        SC_CHECK_HIP(hipMemcpyAsync(d_out, sigma, (size_t)P * E * 8, hipMemcpyDeviceToDevice, st));
        SC_CHECK_HIP(hipStreamSynchronize(st));
        return SC_OK;
    }
    if (big) {          // H = G (H0 + lam I)^-1 on the non-negative bins, then the per-(window, bin) sums of |H|^2
        int rcb = mv_launch_gemm(C, MV_GEMM_PLAIN, dim3((unsigned)F, (unsigned)P), st,
                                 MvMat{const_cast<cd*>(G), N * (int64_t)E, (int64_t)E, 1}, MvMat{hinvc, (int64_t)E, 0, 1},
                                 MvMat{H, F * (int64_t)E, (int64_t)E, 1}, nullptr, nullptr);
        if (rcb != SC_OK) return rcb;
        hipLaunchKernelGGL(m_sumsq, dim3((unsigned)(P * F)), dim3(256), 0, st, H, sq, E);
    } else {
        hipLaunchKernelGGL(m_transfer, dim3((unsigned)F, (unsigned)P), dim3(nt), lds, st, G, hinv, H, sq, N, F, (int)C);
    }
    if (which == SC_MVAR_TRANSFER) {
        SC_CHECK_HIP(hipMemcpyAsync(d_out, H, (size_t)P * F * E * sizeof(cd), hipMemcpyDeviceToDevice, st));
        SC_CHECK_HIP(hipStreamSynchronize(st));
        return SC_OK;
    }
    hipLaunchKernelGGL(m_sum, dim3(1), dim3(64), 0, st, sq, P * F, 1e-12 / (double)(P * F * E), lam + 1);
    if (big) {          // A_mvar = (H + lam' I)^-1 per (window, bin); the launch overwrites the parked H0 matrices last
        int rcb = mv_launch_inverse_big(C, dim3((unsigned)F, (unsigned)P), st, MvMat{H, F * (int64_t)E, (int64_t)E, 1}, lam + 1,
                                        MvMat{Amv, F * (int64_t)E, (int64_t)E, 1}, nullptr, scratch);
        if (rcb != SC_OK) return rcb;
    } else {
        const int Q = (int)((C + 15) / 16);
        const size_t glds = mv_gj_lds(Q);
#define MV_INV(QQ)                                                                                              \
    case QQ:                                                                                                    \
        (void)hipFuncSetAttribute((const void*)m_inverse_gj<QQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds); \
        hipLaunchKernelGGL(m_inverse_gj<QQ>, dim3((unsigned)(P * F)), dim3(256), glds, st, H, lam + 1, Amv, (int)C); \
        break;
        switch (Q) { MV_INV(1) MV_INV(2) MV_INV(3) default: MV_INV(4) }
#undef MV_INV
    }
    if (which == SC_MVAR_COEFFICIENTS) {
        SC_CHECK_HIP(hipMemcpyAsync(d_out, Amv, (size_t)P * F * E * sizeof(cd), hipMemcpyDeviceToDevice, st));
        SC_CHECK_HIP(hipStreamSynchronize(st));
        return SC_OK;
    }
    if (which == SC_MVAR_DDTF) hipLaunchKernelGGL(m_inflow_all, dim3((unsigned)P), dim3(64), 0, st, H, tot, F, (int)C);
    const size_t mlds = huge ? (size_t)C * 8 : (size_t)(E + C) * 8;
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)m_measure, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
    hipLaunchKernelGGL(m_measure, dim3((unsigned)(P * F)), dim3(nt), mlds, st, H, Amv, sigma, tot, which,
                       (double*)d_out, F, (int)C, huge ? 0 : 1, (double*)scratch);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        sc_set_error("MVAR measure failed: %s", hipGetErrorString(hipGetLastError()));
        return SC_EHIP;
    }
    return SC_OK;
}
