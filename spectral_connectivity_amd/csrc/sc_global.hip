// sc_global.hip -- global coherence.
//
// Reference: Connectivity.global_coherence / _estimate_global_coherence (connectivity.py:822-895,
// :2245-2279) take, per (time window, frequency bin -- all N two-sided bins), the thin SVD of the
// n_signals x (n_trials n_tapers) coefficient matrix X and return the leading squared singular values
// / n_estimates and the left singular vectors.  X X^H / n_estimates IS the cross-spectral matrix the
// engine already holds on the device, so the same numbers are the leading eigenpairs of that C x C
// Hermitian matrix: one workgroup per (window, bin) runs a parallel cyclic Jacobi in fp64 on the
// LDS-resident matrix (C / 2 disjoint rotations per step, round-robin pairing), accumulating the
// eigenvectors.  Matrix + eigenvectors in LDS: C <= 64; 64 < C <= 128: see global_coherence_big_kernel.
#include <cstdlib>
#include <cstring>
#include "sc_common.h"
#include "sc_jacobi.h"

typedef double2 cd;

#define GC_CMAX 64
#define GC_EIGH_MIN 32       // beyond: Householder + bisection + inverse iteration (global_coherence_eigh_kernel)

struct GcArgs {
    ScRec accum;
    double* values;        // [P][N][max_rank]
    cd* vectors;           // [P][N][C][max_rank]
    int64_t N, F, floats_per_bin;
    int C, NB, n_tiles, p_csm, two_sided, max_rank, ascending;
    double n_obs;
};

__global__ void __launch_bounds__(256) global_coherence_kernel(GcArgs a) {
    extern __shared__ __align__(16) unsigned char gc_smem[];
    const int C = a.C, M = C + (C & 1);                 // players of the round-robin (dummy if C is odd)
    cd* A = reinterpret_cast<cd*>(gc_smem);             // [C][C]
    cd* V = A + C * C;                                  // [C][C]
    double* rc = reinterpret_cast<double*>(V + C * C);  // [M/2] cos
    cd* rs = reinterpret_cast<cd*>(rc + M / 2 + (M / 2 & 1));   // [M/2] sin * e^{i phi}
    int* rp = reinterpret_cast<int*>(rs + M / 2);       // [M/2][2] pair indices (-1: idle)
    double* ev = reinterpret_cast<double*>(rp + M + (M & 1));   // [C] eigenvalues
    int* order = reinterpret_cast<int*>(ev + C);        // [C]
    __shared__ double red[2][256];
    __shared__ int done;
    const int tid = threadIdx.x;
    const int64_t n = blockIdx.x, p = blockIdx.y;
    int64_t bin = n;
    bool conj = false;
    if (!a.two_sided && n > a.N / 2) { bin = a.N - n; conj = true; }   // real input: S(-f) = conj S(f)
    const ScRec rec = a.accum + (p * a.F + bin) * a.floats_per_bin;
    for (int e = tid; e < C * C; e += 256) {
        const int i = e / C, j = e % C;
        int ti = i >> 4, tj = j >> 4, ii = i & 15, jj = j & 15;
        const bool m = (ti > tj) || (ti == tj && ii > jj);
        if (m) { int t = ti; ti = tj; tj = t; t = ii; ii = jj; jj = t; }
        const int64_t off = (int64_t)sc_tile_index(ti, tj, a.NB) * SC_TILE_ELEMS + ii * 16 + jj;
        const double re = (double)rec[(int64_t)a.p_csm * a.n_tiles * SC_TILE_ELEMS + off] / a.n_obs;
        double im = (double)rec[(int64_t)(a.p_csm + 1) * a.n_tiles * SC_TILE_ELEMS + off] / a.n_obs;
        if (m) im = -im;
        if (conj) im = -im;
        if (i == j) im = 0.0;
        A[e] = make_double2(re, im);
        V[e] = make_double2(i == j ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    for (int sweep = 0; sweep < 16; ++sweep) {
        // convergence: off-diagonal mass against the diagonal
        double off = 0.0, dia = 0.0;
        for (int e = tid; e < C * C; e += 256) {
            const int i = e / C, j = e % C;
            const double v = A[e].x * A[e].x + A[e].y * A[e].y;
            if (i == j) dia += v; else off += v;
        }
        red[0][tid] = off; red[1][tid] = dia;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
            __syncthreads();
        }
        if (tid == 0) done = (red[0][0] <= 1e-28 * red[1][0] || red[0][0] == 0.0) ? 1 : 0;
        __syncthreads();
        if (done) break;
        for (int r = 0; r < M - 1; ++r) {
            // rotation parameters of the M/2 disjoint pairs of this round
            if (tid < M / 2) {
                int x, y;
                if (tid == 0) { x = M - 1; y = r; }
                else { x = (r + tid) % (M - 1); y = (r - tid + (M - 1)) % (M - 1); }
                int pi = x < y ? x : y, qi = x < y ? y : x;
                double c = 1.0;
                cd se = make_double2(0.0, 0.0);
                if (qi >= C) { pi = -1; qi = -1; }          // the dummy player sits out
                else {
                    const cd b = A[pi * C + qi];
                    const double ab = hypot(b.x, b.y);
                    if (ab < 1e-300) { pi = -1; qi = -1; }
                    else {
                        const double tau = (A[qi * C + qi].x - A[pi * C + pi].x) / (2.0 * ab);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + hypot(1.0, tau));
                        c = 1.0 / hypot(1.0, t);
                        const double s = t * c;
                        se = make_double2(s * b.x / ab, s * b.y / ab);
                    }
                }
                rc[tid] = c; rs[tid] = se; rp[2 * tid] = pi; rp[2 * tid + 1] = qi;
            }
            __syncthreads();
            // columns p, q of A and V
            for (int idx = tid; idx < (M / 2) * C; idx += 256) {
                const int pr = idx / C, k = idx % C;
                const int pi = rp[2 * pr], qi = rp[2 * pr + 1];
                if (pi < 0) continue;
                const double c = rc[pr];
                const cd se = rs[pr], sec = make_double2(se.x, -se.y);
                {
                    const cd cp = A[k * C + pi], cq = A[k * C + qi];
                    const cd t1 = g_mul(sec, cq), t2 = g_mul(se, cp);
                    A[k * C + pi] = make_double2(c * cp.x - t1.x, c * cp.y - t1.y);
                    A[k * C + qi] = make_double2(t2.x + c * cq.x, t2.y + c * cq.y);
                }
                {
                    const cd cp = V[k * C + pi], cq = V[k * C + qi];
                    const cd t1 = g_mul(sec, cq), t2 = g_mul(se, cp);
                    V[k * C + pi] = make_double2(c * cp.x - t1.x, c * cp.y - t1.y);
                    V[k * C + qi] = make_double2(t2.x + c * cq.x, t2.y + c * cq.y);
                }
            }
            __syncthreads();
            // rows p, q of A
            for (int idx = tid; idx < (M / 2) * C; idx += 256) {
                const int pr = idx / C, k = idx % C;
                const int pi = rp[2 * pr], qi = rp[2 * pr + 1];
                if (pi < 0) continue;
                const double c = rc[pr];
                const cd se = rs[pr], sec = make_double2(se.x, -se.y);
                const cd r1 = A[pi * C + k], r2 = A[qi * C + k];
                const cd t1 = g_mul(se, r2), t2 = g_mul(sec, r1);
                A[pi * C + k] = make_double2(c * r1.x - t1.x, c * r1.y - t1.y);
                A[qi * C + k] = make_double2(t2.x + c * r2.x, t2.y + c * r2.y);
            }
            __syncthreads();
        }
    }
    // rank the eigenvalues (descending, ties by index)
    for (int i = tid; i < C; i += 256) ev[i] = A[i * C + i].x;
    __syncthreads();
    for (int i = tid; i < C; i += 256) {
        int rank = 0;
        for (int j = 0; j < C; ++j) rank += (ev[j] > ev[i] || (ev[j] == ev[i] && j < i)) ? 1 : 0;
        order[rank] = i;
    }
    __syncthreads();
    const int K = a.max_rank;
    double* val = a.values + (p * a.N + n) * K;
    cd* vec = a.vectors + (p * a.N + n) * (int64_t)C * K;
    for (int k = tid; k < K; k += 256) {
        const int src = a.ascending ? order[K - 1 - k] : order[k];     // svds returns the K largest, smallest first
        const double v = ev[src];
        val[k] = v > 0.0 ? v : 0.0;
    }
    // unit-norm eigenvectors with the phase fixed by making the largest component real and positive
    for (int k = 0; k < K; ++k) {
        const int src = a.ascending ? order[K - 1 - k] : order[k];
        __shared__ cd phase;
        if (tid == 0) {
            double best = -1.0;
            cd b = make_double2(1.0, 0.0);
            for (int i = 0; i < C; ++i) {
                const cd v = V[i * C + src];
                const double m2 = v.x * v.x + v.y * v.y;
                if (m2 > best) { best = m2; b = v; }
            }
            const double ab = sqrt(best);
            phase = ab > 0.0 ? make_double2(b.x / ab, -b.y / ab) : make_double2(1.0, 0.0);
        }
        __syncthreads();
        for (int i = tid; i < C; i += 256) vec[(int64_t)i * K + k] = g_mul(V[i * C + src], phase);
        __syncthreads();
    }
}

// ---- 64 < C <= 128: the matrix alone fills the LDS --------------------------------------------------
// Upper triangle of the Hermitian matrix packed in LDS (C (C+1) / 2 complex128 = 132 KB at C = 128), no room
// for the eigenvector matrix.  One thread per (pair a, pair b) 2x2 block of the round's pairing transforms
// its four entries in place (B' = J_a^H B J_b: the disjoint rotations of a round act on disjoint blocks), and
// the rotation angles of every round are logged to a global scratch; the requested eigenvectors are then
// V e_k = J_1 J_2 ... J_m e_k, evaluated right to left on a single C-vector per eigenvector -- the C x C
// eigenvector matrix is never formed.
#define GC_BIG_CMAX 128      // packed triangle in LDS
#define GC_HUGE_CMAX 256     // packed triangle (526 KB at 256 signals) in a per-workgroup global scratch: the same kernel on
                             // an L2-resident matrix, an order of magnitude slower per rotation but no new algorithm
#define GC_BIG_SWEEPS 12

struct GcBigArgs {
    GcArgs g;
    double* log;           // [slots][GC_BIG_SWEEPS * (M - 1)][M / 2][3]  (cos, Re s, Im s)
    cd* scratch;           // [slots][C (C + 1) / 2] packed triangles when they do not fit LDS (n_signals > 128), else NULL
    int n_bins_total;
    int batch;             // eigenvectors back-applied at a time (what the LDS holds)
};

template <int NT>
__global__ void __launch_bounds__(NT) global_coherence_big_kernel(GcBigArgs b) {
    extern __shared__ __align__(16) unsigned char gc_smem[];
    const GcArgs& a = b.g;
    const int C = a.C, M = C + (C & 1), H = M / 2;
    const size_t tri = (size_t)C * (C + 1) / 2;
    // packed upper triangle: in LDS, or (n_signals > 128) this workgroup's slice of the global scratch -- a workgroup's
    // waves share one L1, and __syncthreads orders its global writes like its LDS writes
    cd* A = b.scratch ? b.scratch + (size_t)blockIdx.x * tri : reinterpret_cast<cd*>(gc_smem);
    double* rc = b.scratch ? reinterpret_cast<double*>(gc_smem) : reinterpret_cast<double*>(A + tri);   // [H] cos
    cd* rs = reinterpret_cast<cd*>(rc + H + (H & 1));                  // [H] sin e^{i phi}
    int* rp = reinterpret_cast<int*>(rs + H);                          // [H][2]
    double* ev = reinterpret_cast<double*>(rp + M + (M & 1));          // [C]
    int* order = reinterpret_cast<int*>(ev + C);                       // [C]
    unsigned short* blk_u = reinterpret_cast<unsigned short*>(order + C + (C & 1));   // [H (H+1) / 2] block -> pair u
    unsigned short* blk_v = blk_u + H * (H + 1) / 2;                                  //                   pair v >= u
    // back-applied vectors, a batch of b.batch at a time: behind the tables (matrix in the global scratch), or IN the
    // space of the packed triangle, which is dead once its diagonal has been copied to ev (matrix in LDS)
    cd* xv = b.scratch ? reinterpret_cast<cd*>((reinterpret_cast<uintptr_t>(blk_v + H * (H + 1) / 2) + 15) & ~(uintptr_t)15)
                       : reinterpret_cast<cd*>(gc_smem);
    __shared__ double red[2][NT];
    __shared__ int done, n_rounds;
    const int tid = threadIdx.x;
    double* mylog = b.log + (size_t)blockIdx.x * GC_BIG_SWEEPS * (M - 1) * H * 3;
    gc_block_table<NT>(blk_u, blk_v, H, tid);           // once per workgroup
    for (int item = blockIdx.x; item < b.n_bins_total; item += gridDim.x) {
        const int64_t p = item / a.N, n = item - p * a.N;
        int64_t bin = n;
        bool conj = false;
        if (!a.two_sided && n > a.N / 2) { bin = a.N - n; conj = true; }
        const ScRec rec = a.accum + (p * a.F + bin) * a.floats_per_bin;
        __syncthreads();
        for (int e = tid; e < C * C; e += NT) {
            const int i = e / C, j = e % C;
            if (i > j) continue;
            const int ti = i >> 4, tj = j >> 4, ii = i & 15, jj = j & 15;      // i <= j: upper triangle, no mirror
            const int64_t off = (int64_t)sc_tile_index(ti, tj, a.NB) * SC_TILE_ELEMS + ii * 16 + jj;
            const double re = (double)rec[(int64_t)a.p_csm * a.n_tiles * SC_TILE_ELEMS + off] / a.n_obs;
            double im = (double)rec[(int64_t)(a.p_csm + 1) * a.n_tiles * SC_TILE_ELEMS + off] / a.n_obs;
            if (conj) im = -im;
            if (i == j) im = 0.0;
            A[gc_tri(i, j, C)] = make_double2(re, im);
        }
        __syncthreads();
        gc_jacobi<NT>(A, C, rc, rs, rp, blk_u, blk_v, red, &done, &n_rounds, mylog, GC_BIG_SWEEPS);
        // rank the eigenvalues (descending, ties by index)
        for (int i = tid; i < C; i += NT) ev[i] = A[gc_tri(i, i, C)].x;
        __syncthreads();                                // (from here on the triangle in LDS may be overwritten by xv)
        for (int i = tid; i < C; i += NT) {
            int rank = 0;
            for (int j = 0; j < C; ++j) rank += (ev[j] > ev[i] || (ev[j] == ev[i] && j < i)) ? 1 : 0;
            order[rank] = i;
        }
        __syncthreads();
        const int K = a.max_rank;
        double* val = a.values + (p * a.N + n) * K;
        cd* vec = a.vectors + (p * a.N + n) * (int64_t)C * K;
        for (int k = tid; k < K; k += NT) {
            const int src = a.ascending ? order[K - 1 - k] : order[k];
            val[k] = ev[src] > 0.0 ? ev[src] : 0.0;
        }
        // eigenvectors: x = J_1 ... J_m e_src, right to left, for a batch of requested components at a time; work item
        // (vector, pair of the round) rotates two entries.  Any max_rank up to n_signals (the reference's full SVD).
        const int total_rounds = n_rounds;
        for (int k0 = 0; k0 < K; k0 += b.batch) {
            const int kb = K - k0 < b.batch ? K - k0 : b.batch;
            for (int e = tid; e < kb * C; e += NT) {
                const int k = k0 + e / C, i = e % C;
                const int src = a.ascending ? order[K - 1 - k] : order[k];
                xv[e] = make_double2(i == src ? 1.0 : 0.0, 0.0);
            }
            __syncthreads();
            for (int rr = total_rounds - 1; rr >= 0; --rr) {
                const int r = rr % (M - 1);
                for (int w = tid; w < kb * H; w += NT) {
                    const int k = w / H, t = w % H;
                    int x, y;
                    if (t == 0) { x = M - 1; y = r; }
                    else { x = (r + t) % (M - 1); y = (r - t + (M - 1)) % (M - 1); }
                    const int pi = x < y ? x : y, qi = x < y ? y : x;
                    if (qi >= C) continue;
                    const double* lg = mylog + ((size_t)rr * H + t) * 3;
                    const double c = lg[0];
                    const cd se = make_double2(lg[1], lg[2]), sec = make_double2(lg[1], -lg[2]);
                    const cd xp = xv[k * C + pi], xq = xv[k * C + qi];
                    const cd t1 = g_mul(se, xq), t2 = g_mul(sec, xp);
                    xv[k * C + pi] = make_double2(c * xp.x + t1.x, c * xp.y + t1.y);
                    xv[k * C + qi] = make_double2(c * xq.x - t2.x, c * xq.y - t2.y);
                }
                __syncthreads();
            }
            // unit phase: the largest component real and positive (one thread per vector finds it)
            cd* phase = reinterpret_cast<cd*>(red);     // [kb <= NT]: the reduction scratch is idle here
            for (int k = tid; k < kb; k += NT) {
                double best = -1.0;
                cd bb = make_double2(1.0, 0.0);
                for (int i = 0; i < C; ++i) {
                    const cd v = xv[k * C + i];
                    const double m2 = v.x * v.x + v.y * v.y;
                    if (m2 > best) { best = m2; bb = v; }
                }
                const double ab = sqrt(best);
                phase[k] = ab > 0.0 ? make_double2(bb.x / ab, -bb.y / ab) : make_double2(1.0, 0.0);
            }
            __syncthreads();
            for (int e = tid; e < kb * C; e += NT) {
                const int k = e / C, i = e % C;
                vec[(int64_t)i * K + k0 + k] = g_mul(xv[k * C + i], phase[k]);
            }
            __syncthreads();
        }
    }
}


// ---- 64 < C <= 256: Householder tridiagonalisation + bisection + inverse iteration ----------------------------------
// The parallel Jacobi above needs ~10 sweeps x (C - 1) rounds, each a full pass over the packed triangle: 2.7 GB of
// scattered L2 traffic per 256 x 256 matrix (190 ms per matrix, 760 ms for the 1024 bins of the cfg5 shape; the LDS
// version 12 ms per 128 x 128 matrix).  The classical dense route moves an order of magnitude less and every pass is a
// coalesced column sweep: reduce the Hermitian matrix to a real symmetric tridiagonal one with C - 2 complex Householder
// reflections (LAPACK zhetd2, lower form: p = tau A v, w = p - (tau / 2)(p^H v) v, A <- A - v w^H - w v^H: two column-major
// passes over the trailing block per step, C^3 x 16 bytes in all = 268 MB at 256 signals), find its eigenvalues by
// bisection on the Sturm count (one thread per eigenvalue, to the last bit of ||T||), the requested eigenvectors of T by
// inverse iteration (one thread per vector: pivoted tridiagonal LU, three solves; vectors of eigenvalues closer than
// 1e-3 ||T|| are orthogonalised against each other, twice, like LAPACK's dstein), and back-transform them through the
// stored reflectors (x = H_0 H_1 ... H_{C-2} y).  One 256-thread workgroup per (window, bin); the matrix (full storage,
// column-major) and the per-vector work arrays live in a per-workgroup global scratch (L2-resident).
struct GcEighArgs {
    GcArgs g;
    cd* A;                 // [slots][C * C]
    double* work;          // [slots][8][C][KS]: tridiagonal LU (u0, u1, u2, l, swap), y, Re x, Im x; thread t fastest
    int n_bins_total, KS;
};

// GE_NT threads: one per row of the matrix / per eigenvalue -- 256 up to 256 signals, 512 beyond (round 5: a 306-channel MEG array
// is an ordinary input, the reference has no limit; 1024 threads would hold 16 elements of every vector per lane in the back-
// transformation: 265 registers spilled)
#define GC_EIGH_CMAX 512
template <int GE_NT>
__device__ __forceinline__ double ge_block_sum(double v, double* red, int tid) {
    red[tid] = v;
    __syncthreads();
    for (int s = GE_NT / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

template <int GE_NT>
__global__ void __launch_bounds__(GE_NT) global_coherence_eigh_kernel(GcEighArgs b) {
    extern __shared__ __align__(16) unsigned char gc_smem[];
    const GcArgs& a = b.g;
    const int C = a.C, K = a.max_rank, KS = b.KS;
    double* d = reinterpret_cast<double*>(gc_smem);        // [C] diagonal of T
    double* e = d + C;                                     // [C] off-diagonal of T (e[k] couples k, k + 1)
    double* e2 = e + C;                                    // [C] e^2
    double* ev = e2 + C;                                   // [C] eigenvalues, ascending
    double* shift = ev + C;                                // [C] perturbed eigenvalues of the requested vectors
    double* coef = shift + C;                              // [C] Gram-Schmidt coefficients
    cd* tau = reinterpret_cast<cd*>(coef + C);             // [C]
    cd* vs = tau + C;                                      // [C] reflector of the step
    cd* ws = vs + C;                                       // [C] p, then w
    double* red = reinterpret_cast<double*>(ws + C);       // [GE_NT]
    __shared__ double sh_scalar[8];
    const int tid = threadIdx.x;
    cd* A = b.A + (size_t)blockIdx.x * C * C;
    double* W = b.work + (size_t)blockIdx.x * 8 * C * KS;
    auto WK = [&](int arr, int i, int t) -> double& { return W[((size_t)arr * C + i) * KS + t]; };
    for (int item = blockIdx.x; item < b.n_bins_total; item += gridDim.x) {
        const int64_t p = item / a.N, n = item - p * a.N;
        int64_t bin = n;
        bool conj = false;
        if (!a.two_sided && n > a.N / 2) { bin = a.N - n; conj = true; }
        const ScRec rec = a.accum + (p * a.F + bin) * a.floats_per_bin;
        __syncthreads();
        for (int el = tid; el < C * C; el += GE_NT) {
            const int i = el % C, j = el / C;              // column-major: consecutive threads walk down a column
            int ti = i >> 4, tj = j >> 4, ii = i & 15, jj = j & 15;
            const bool m = (ti > tj) || (ti == tj && ii > jj);
            if (m) { int t = ti; ti = tj; tj = t; t = ii; ii = jj; jj = t; }
            const int64_t off = (int64_t)sc_tile_index(ti, tj, a.NB) * SC_TILE_ELEMS + ii * 16 + jj;
            const double re = (double)rec[(int64_t)a.p_csm * a.n_tiles * SC_TILE_ELEMS + off] / a.n_obs;
            double im = (double)rec[(int64_t)(a.p_csm + 1) * a.n_tiles * SC_TILE_ELEMS + off] / a.n_obs;
            if (m) im = -im;
            if (conj) im = -im;
            if (i == j) im = 0.0;
            A[el] = make_double2(re, im);
        }
        __syncthreads();
        // ---- tridiagonalisation (zhetd2, lower) ----
        for (int k = 0; k + 1 < C; ++k) {
            const int m = C - k - 1;                       // order of the trailing block; m < C <= GE_NT
            cd* col = A + (size_t)k * C + (k + 1);
            cd xi = make_double2(0.0, 0.0);
            if (tid < m) xi = col[tid];
            const double xn2 = ge_block_sum<GE_NT>((tid >= 1 && tid < m) ? xi.x * xi.x + xi.y * xi.y : 0.0, red, tid);
            if (tid == 0) { sh_scalar[0] = xi.x; sh_scalar[1] = xi.y; d[k] = A[(size_t)k * C + k].x; }
            __syncthreads();
            const double alr = sh_scalar[0], ali = sh_scalar[1];
            if (xn2 == 0.0 && ali == 0.0) {                // H = I (uniform over the workgroup)
                if (tid == 0) { tau[k] = make_double2(0.0, 0.0); e[k] = alr; }
                if (tid < m) col[tid] = make_double2(tid == 0 ? 1.0 : 0.0, 0.0);
                __syncthreads();
                continue;
            }
            const double nrm = sqrt(alr * alr + ali * ali + xn2);
            const double beta = alr >= 0.0 ? -nrm : nrm;
            const cd tk = make_double2((beta - alr) / beta, -ali / beta);
            {
                const double dr = alr - beta, di = ali, dd = dr * dr + di * di;
                const cd scale = make_double2(dr / dd, -di / dd);          // 1 / (alpha - beta)
                cd v = make_double2(1.0, 0.0);
                if (tid >= 1 && tid < m) v = g_mul(xi, scale);
                if (tid < m) { vs[tid] = v; col[tid] = v; }
                if (tid == 0) { tau[k] = tk; e[k] = beta; }
            }
            __syncthreads();
            const cd* A22 = A + (size_t)(k + 1) * C + (k + 1);
            cd pi = make_double2(0.0, 0.0);
            if (tid < m) {
                cd acc = make_double2(0.0, 0.0);
                for (int j = 0; j < m; ++j) {
                    const cd aij = A22[(size_t)j * C + tid], vj = vs[j];
                    acc.x += aij.x * vj.x - aij.y * vj.y;
                    acc.y += aij.x * vj.y + aij.y * vj.x;
                }
                pi = g_mul(tk, acc);
            }
            const cd vi = tid < m ? vs[tid] : make_double2(0.0, 0.0);
            const double dre = ge_block_sum<GE_NT>(pi.x * vi.x + pi.y * vi.y, red, tid);       // p^H v
            const double dim = ge_block_sum<GE_NT>(pi.x * vi.y - pi.y * vi.x, red, tid);
            const cd al2 = g_mul(make_double2(-0.5 * tk.x, -0.5 * tk.y), make_double2(dre, dim));
            cd wi = make_double2(0.0, 0.0);
            if (tid < m) {
                const cd t = g_mul(al2, vi);
                wi = make_double2(pi.x + t.x, pi.y + t.y);
                ws[tid] = wi;
            }
            __syncthreads();
            if (tid < m) {
                cd* row = A + (size_t)(k + 1) * C + (k + 1) + tid;
                for (int j = 0; j < m; ++j) {
                    const cd wj = ws[j], vj = vs[j];
                    cd aij = row[(size_t)j * C];
                    // a_ij -= v_i conj(w_j) + w_i conj(v_j)
                    aij.x -= vi.x * wj.x + vi.y * wj.y + wi.x * vj.x + wi.y * vj.y;
                    aij.y -= vi.y * wj.x - vi.x * wj.y + wi.y * vj.x - wi.x * vj.y;
                    row[(size_t)j * C] = aij;
                }
            }
            __syncthreads();
        }
        if (tid == 0) { d[C - 1] = A[(size_t)(C - 1) * C + (C - 1)].x; e[C - 1] = 0.0; }
        __syncthreads();
        // ---- eigenvalues of T by bisection: thread j finds the j-th smallest ----
        double gl = 0.0, gu = 0.0, tn = 0.0;
        {
            double lo = 1e300, hi = -1e300, nr = 0.0, emax = 0.0;
            if (tid < C) {
                const double el = tid > 0 ? fabs(e[tid - 1]) : 0.0, er = tid + 1 < C ? fabs(e[tid]) : 0.0;
                lo = d[tid] - el - er; hi = d[tid] + el + er; nr = fabs(d[tid]) + el + er;
                e2[tid] = e[tid] * e[tid];
                emax = e2[tid];
            }
            red[tid] = lo; __syncthreads();
            for (int s = GE_NT / 2; s > 0; s >>= 1) { if (tid < s) red[tid] = fmin(red[tid], red[tid + s]); __syncthreads(); }
            gl = red[0]; __syncthreads();
            red[tid] = hi; __syncthreads();
            for (int s = GE_NT / 2; s > 0; s >>= 1) { if (tid < s) red[tid] = fmax(red[tid], red[tid + s]); __syncthreads(); }
            gu = red[0]; __syncthreads();
            red[tid] = nr; __syncthreads();
            for (int s = GE_NT / 2; s > 0; s >>= 1) { if (tid < s) red[tid] = fmax(red[tid], red[tid + s]); __syncthreads(); }
            tn = red[0]; __syncthreads();                  // ||T||_1 (= inf-norm)
            red[tid] = emax; __syncthreads();
            for (int s = GE_NT / 2; s > 0; s >>= 1) { if (tid < s) red[tid] = fmax(red[tid], red[tid + s]); __syncthreads(); }
            emax = red[0]; __syncthreads();
            sh_scalar[2] = 2.2250738585072014e-308 * fmax(1.0, emax);      // pivmin
        }
        __syncthreads();
        const double pivmin = sh_scalar[2];
        const double eps = 2.220446049250313e-16;
        {
            const double pad = 2.0 * tn * eps * C + 2.0 * pivmin;
            gl -= pad; gu += pad;
        }
        if (tid < C) {
            double lo = gl, hi = gu;
            for (int it = 0; it < 120; ++it) {
                const double mid = 0.5 * (lo + hi);
                if (!(hi - lo > 2.0 * eps * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin) || mid <= lo || mid >= hi) break;
                int cnt = 0;
                double q = d[0] - mid;
                if (fabs(q) < pivmin) q = -pivmin;
                cnt += q < 0.0;
                for (int i = 1; i < C; ++i) {
                    q = d[i] - mid - e2[i - 1] / q;
                    if (fabs(q) < pivmin) q = -pivmin;
                    cnt += q < 0.0;
                }
                if (cnt > tid) hi = mid; else lo = mid;    // cnt = eigenvalues below mid
            }
            ev[tid] = 0.5 * (lo + hi);
        }
        __syncthreads();
        double* val = a.values + (p * a.N + n) * K;
        cd* vec = a.vectors + (p * a.N + n) * (int64_t)C * K;
        for (int k = tid; k < K; k += GE_NT) {
            const int r = a.ascending ? K - 1 - k : k;     // rank from the top
            const double v = ev[C - 1 - r];
            val[k] = v > 0.0 ? v : 0.0;
        }
        // ---- eigenvectors of the K largest eigenvalues: thread t <-> eigenvalue index j = C - K + t (ascending) ----
        // shifts: equal / nearly equal eigenvalues are pulled apart by 10 eps ||T|| (dstein), scanned by one thread
        if (tid == 0) {
            const double eps1 = 10.0 * eps * tn;
            double prev = 0.0;
            for (int t = 0; t < K; ++t) {
                double x = ev[C - K + t];
                if (t > 0 && x - prev < eps1) x = prev + eps1;
                shift[t] = x;
                prev = x;
            }
        }
        __syncthreads();
        const bool has_vec = tid < K;
        const int t = tid;
        auto solve_setup = [&]() {                         // pivoted LU of T - shift I, row by row (dlagtf without the scaling)
            const double lam = shift[t];
            const double tol = fmax(eps * tn, pivmin);
            double a0 = d[0] - lam, a1 = C > 1 ? e[0] : 0.0, a2 = 0.0;     // current row i: entries at columns i, i + 1, i + 2
            for (int i = 0; i + 1 < C; ++i) {
                const double c0 = e[i], c1 = d[i + 1] - lam, c2 = i + 2 < C ? e[i + 1] : 0.0;   // next row: columns i, i + 1, i + 2
                double l;
                if (fabs(a0) >= fabs(c0)) {                // no interchange
                    if (fabs(a0) < tol) a0 = a0 < 0.0 ? -tol : tol;
                    l = c0 / a0;
                    WK(0, i, t) = a0; WK(1, i, t) = a1; WK(2, i, t) = a2; WK(3, i, t) = l; WK(4, i, t) = 0.0;
                    a0 = c1 - l * a1; a1 = c2 - l * a2; a2 = 0.0;
                } else {                                   // interchange rows i and i + 1
                    l = a0 / c0;
                    WK(0, i, t) = c0; WK(1, i, t) = c1; WK(2, i, t) = c2; WK(3, i, t) = l; WK(4, i, t) = 1.0;
                    a0 = a1 - l * c1; a1 = a2 - l * c2; a2 = 0.0;
                }
            }
            if (fabs(a0) < tol) a0 = a0 < 0.0 ? -tol : tol;
            WK(0, C - 1, t) = a0; WK(1, C - 1, t) = 0.0; WK(2, C - 1, t) = 0.0;
        };
        auto solve = [&]() {                               // y <- (T - shift I)^-1 y, then unit 2-norm
            for (int i = 0; i + 1 < C; ++i) {              // forward: the row operations of the factorisation
                const double l = WK(3, i, t);
                double yi = WK(5, i, t), yn = WK(5, i + 1, t);
                if (WK(4, i, t) != 0.0) { const double s = yi; yi = yn; yn = s; }
                yn -= l * yi;
                WK(5, i, t) = yi; WK(5, i + 1, t) = yn;
            }
            double nrm2 = 0.0;
            for (int i = C - 1; i >= 0; --i) {             // backward: U has three diagonals
                double s = WK(5, i, t);
                if (i + 1 < C) s -= WK(1, i, t) * WK(5, i + 1, t);
                if (i + 2 < C) s -= WK(2, i, t) * WK(5, i + 2, t);
                s /= WK(0, i, t);
                if (!(fabs(s) < 1e290)) s = s < 0.0 ? -1e290 : 1e290;    // (overflow guard; the vector is normalised below)
                WK(5, i, t) = s;
                nrm2 = fmax(nrm2, fabs(s));
            }
            double ss = 0.0;
            const double inv = nrm2 > 0.0 ? 1.0 / nrm2 : 1.0;
            for (int i = 0; i < C; ++i) { const double s = WK(5, i, t) * inv; WK(5, i, t) = s; ss += s * s; }
            const double sc = ss > 0.0 ? 1.0 / sqrt(ss) : 1.0;
            for (int i = 0; i < C; ++i) WK(5, i, t) *= sc;
        };
        // vectors of one cluster (eigenvalues closer than 1e-3 ||T||) are orthogonalised in ascending order, classical
        // Gram-Schmidt applied twice: thread i computes <y_i, y_j>, then thread `row` updates element `row` of y_j
        auto orthogonalise = [&]() {
            const double ortol = 1e-3 * tn;
            int c0 = 0;                                    // first vector of the current cluster (uniform: read from LDS)
            for (int j = 1; j < K; ++j) {
                if (shift[j] - shift[j - 1] >= ortol) { c0 = j; continue; }
                for (int pass = 0; pass < 2; ++pass) {
                    if (tid >= c0 && tid < j) {
                        double s = 0.0;
                        for (int i = 0; i < C; ++i) s += WK(5, i, tid) * WK(5, i, j);
                        coef[tid] = s;
                    }
                    __syncthreads();
                    if (tid < C) {
                        double s = WK(5, tid, j);
                        for (int i = c0; i < j; ++i) s -= coef[i] * WK(5, tid, i);
                        WK(5, tid, j) = s;
                    }
                    __syncthreads();
                }
                const double nn = ge_block_sum<GE_NT>(tid < C ? WK(5, tid, j) * WK(5, tid, j) : 0.0, red, tid);
                if (tid < C && nn > 0.0) WK(5, tid, j) *= 1.0 / sqrt(nn);
                __syncthreads();
            }
        };
        if (has_vec) {
            solve_setup();
            unsigned seed = 1234567u + 7919u * (unsigned)(C - K + t);
            for (int i = 0; i < C; ++i) {                  // deterministic start vector in (-1, 1)
                seed = seed * 1664525u + 1013904223u;
                WK(5, i, t) = ((double)(seed >> 8) + 0.5) / 8388608.0 - 1.0;
            }
            solve();
            solve();
        }
        __syncthreads();
        orthogonalise();
        if (has_vec) solve();
        __syncthreads();
        orthogonalise();
        // ---- back-transformation x = H_0 H_1 ... H_{C-2} y and output: one WAVE per vector, the vector in registers (four
        // elements per lane), reflector columns read coalesced, the two dot products per step reduced with lane shuffles
        // (no barrier: the chain of C - 1 reflections is sequential, one thread per vector would pay an L2 round trip per
        // element) ----
        __syncthreads();
        {
            const int lane = tid & 63, wv = tid >> 6;
            for (int tv = wv; tv < K; tv += GE_NT / 64) {
                constexpr int XQ = GE_NT / 64;                      // elements of the vector per lane
                double xr[XQ], xim[XQ];
#pragma unroll
                for (int q = 0; q < XQ; ++q) {
                    const int i = lane + 64 * q;
                    xr[q] = i < C ? WK(5, i, tv) : 0.0;
                    xim[q] = 0.0;
                }
                for (int k = C - 2; k >= 0; --k) {
                    const cd tk = tau[k];
                    if (tk.x == 0.0 && tk.y == 0.0) continue;
                    const cd* col = A + (size_t)k * C;             // v_i sits at col[i], i = k + 1 .. C - 1 (v_{k+1} = 1)
                    cd vq[XQ];
                    double sr = 0.0, si = 0.0;                      // v^H x
#pragma unroll
                    for (int q = 0; q < XQ; ++q) {
                        const int i = lane + 64 * q;
                        vq[q] = (i > k && i < C) ? col[i] : make_double2(0.0, 0.0);
                        sr += vq[q].x * xr[q] + vq[q].y * xim[q];
                        si += vq[q].x * xim[q] - vq[q].y * xr[q];
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) { sr += __shfl_xor(sr, off); si += __shfl_xor(si, off); }
                    const cd f = g_mul(tk, make_double2(sr, si));
#pragma unroll
                    for (int q = 0; q < XQ; ++q) {
                        xr[q] -= vq[q].x * f.x - vq[q].y * f.y;
                        xim[q] -= vq[q].x * f.y + vq[q].y * f.x;
                    }
                }
                // unit norm, largest component real and positive
                double ss = 0.0, best = -1.0, br = 1.0, bi = 0.0;
                int bidx = 0;                                       // ties go to the smaller element index: every lane agrees
#pragma unroll
                for (int q = 0; q < XQ; ++q) {
                    const double m2 = xr[q] * xr[q] + xim[q] * xim[q];
                    ss += m2;
                    if (m2 > best) { best = m2; br = xr[q]; bi = xim[q]; bidx = lane + 64 * q; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    ss += __shfl_xor(ss, off);
                    const double ob = __shfl_xor(best, off), obr = __shfl_xor(br, off), obi = __shfl_xor(bi, off);
                    const int oidx = __shfl_xor(bidx, off);
                    if (ob > best || (ob == best && oidx < bidx)) { best = ob; br = obr; bi = obi; bidx = oidx; }
                }
                const double ab = sqrt(best), sc = ss > 0.0 ? 1.0 / sqrt(ss) : 1.0;
                const cd ph = ab > 0.0 ? make_double2(br / ab * sc, -bi / ab * sc) : make_double2(sc, 0.0);
                const int r = K - 1 - tv;                          // rank from the top of eigenvalue index C - K + tv
                const int kout = a.ascending ? K - 1 - r : r;
#pragma unroll
                for (int q = 0; q < XQ; ++q) {
                    const int i = lane + 64 * q;
                    if (i < C) vec[(int64_t)i * K + kout] = g_mul(make_double2(xr[q], xim[q]), ph);
                }
            }
        }
    }
}

extern "C" int sc_global_coherence_max_signals(void) { return GC_EIGH_CMAX; }

extern "C" int sc_global_coherence_f64(const void* d_accum, int64_t n_groups, int64_t n_freq_accum, int64_t N,
                                       int64_t C, uint32_t planes, int64_t n_obs, int max_rank, int ascending,
                                       double* d_values, void* d_vectors, void* stream) {
    ScTimed timed_("global_coherence", stream);
    SC_REQUIRE(d_accum && d_values && d_vectors, "NULL argument");
    SC_REQUIRE(planes & SC_PLANE_CSM, "accumulator record must contain SC_PLANE_CSM");
    SC_REQUIRE(n_freq_accum == N || n_freq_accum == N / 2 + 1, "accumulators must hold N or N/2+1 bins");
    SC_REQUIRE(n_groups >= 1 && n_groups <= 65535 && N >= 1 && n_obs >= 1, "bad problem size");
    if (C < 1 || C > GC_EIGH_CMAX) {
        sc_set_error("global coherence: n_signals <= %d (got %lld)", GC_EIGH_CMAX, (long long)C);
        return SC_EUNSUPPORTED;
    }
    SC_REQUIRE(max_rank >= 1 && max_rank <= C, "max_rank must be in 1..n_signals");
    GcArgs a;
    a.accum = sc_rec(d_accum, planes); a.values = d_values; a.vectors = (cd*)d_vectors;
    a.N = N; a.F = n_freq_accum; a.C = (int)C;
    a.NB = sc_n_blocks(C); a.n_tiles = sc_n_tiles(a.NB);
    a.p_csm = sc_plane_offset(planes, SC_PLANE_CSM);
    a.two_sided = (n_freq_accum == N && N > 1) ? 1 : 0;
    a.floats_per_bin = (int64_t)sc_plane_count(planes) * a.n_tiles * SC_TILE_ELEMS;
    a.max_rank = max_rank; a.ascending = ascending; a.n_obs = (double)n_obs;
    const int M = (int)C + ((int)C & 1);
    const char* eig_env = sc_switch(SC_SW_GLOBAL_EIG);       // "jacobi": the round-2 kernels beyond 64 signals too (cross-check)
    if (C > GC_HUGE_CMAX && eig_env && strcmp(eig_env, "jacobi") == 0) {
        sc_set_error("global coherence: the Jacobi kernels (SC_GLOBAL_EIG=jacobi) take n_signals <= %d (got %lld)", GC_HUGE_CMAX, (long long)C);
        return SC_EUNSUPPORTED;
    }
    // (round 6: from 33 signals on -- the LDS Jacobi kernel below took 22.5 ms for 1024 bins of 64 signals where this one takes 3.3 at 65;
    //  tools/cliff_sweep.py)
    if (C > GC_EIGH_MIN && !(eig_env && strcmp(eig_env, "jacobi") == 0)) {
        // Householder tridiagonalisation + bisection + inverse iteration, matrix and work arrays in a scratch of this call
        const int64_t bins = n_groups * N;
        // slots (workgroups that walk the bins): as many as 1 GB of scratch holds, at most 1024; halved while the allocation fails
        const size_t per_slot = (size_t)C * C * sizeof(cd) + (size_t)8 * C * max_rank * sizeof(double);
        int slots = (int)(bins < 1024 ? bins : 1024);
        const size_t budget_slots = ((size_t)1 << 30) / per_slot;
        if ((size_t)slots > budget_slots) slots = budget_slots < 1 ? 1 : (int)budget_slots;
        char* scratch = nullptr;
        while (hipMalloc((void**)&scratch, (size_t)slots * per_slot) != hipSuccess) {
            (void)hipGetLastError();
            scratch = nullptr;
            if (slots == 1) {
                sc_set_error("global coherence: scratch alloc failed (%zu bytes)", per_slot);
                return SC_ENOMEM;
            }
            slots = (slots + 1) / 2;
        }
        const size_t a_bytes = (size_t)slots * C * C * sizeof(cd);
        GcEighArgs b;
        b.g = a; b.A = (cd*)scratch; b.work = (double*)(scratch + a_bytes); b.n_bins_total = (int)bins; b.KS = max_rank;
        const int nt = C <= 256 ? 256 : 512;
        const size_t lds = ((size_t)12 * C + nt) * sizeof(double) + 64;
        if (nt == 256) {
            hipLaunchKernelGGL(global_coherence_eigh_kernel<256>, dim3((unsigned)slots), dim3(256), lds, (hipStream_t)stream, b);
        } else {
            (void)hipFuncSetAttribute((const void*)global_coherence_eigh_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(global_coherence_eigh_kernel<512>, dim3((unsigned)slots), dim3(512), lds, (hipStream_t)stream, b);
        }
        const hipError_t e1 = hipGetLastError();
        const hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);       // the scratch is freed below
        (void)hipFree(scratch);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            sc_set_error("global coherence (n_signals > 64) failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            return SC_EHIP;
        }
        return SC_OK;
    }
    if (C > GC_CMAX) {
        // matrix only in LDS, rotation log in a device scratch owned by this call
        const int H = M / 2;
        const int64_t bins = n_groups * N;
        const bool huge = C > GC_BIG_CMAX;
        const int max_slots = huge ? 256 : 512;
        const int slots = (int)(bins < max_slots ? bins : max_slots);
        const size_t tri_bytes = (size_t)C * (C + 1) / 2 * sizeof(cd);
        const size_t log_bytes = (size_t)slots * GC_BIG_SWEEPS * (M - 1) * H * 3 * sizeof(double);
        double* log = nullptr;
        if (hipMalloc((void**)&log, log_bytes + (huge ? slots * tri_bytes : 0)) != hipSuccess) {
            sc_set_error("global coherence: rotation log alloc failed");
            return SC_ENOMEM;
        }
        GcBigArgs b;
        b.g = a; b.log = log; b.n_bins_total = (int)bins;
        b.scratch = huge ? reinterpret_cast<cd*>(reinterpret_cast<char*>(log) + log_bytes) : nullptr;
        const size_t tables = (size_t)(H + 2) * 8 + (size_t)H * 16 + (size_t)(M + 2) * 4 + (size_t)C * 8 + (size_t)(C + 4) * 4 +
                              (size_t)H * (H + 1) * 2 + 16 + 64;
        // eigenvector batches: in the dead triangle (matrix in LDS: (C + 1) / 2 vectors), or behind the tables, 96 KB
        const size_t vec_bytes = (size_t)C * sizeof(cd);
        size_t xv_bytes = 0;
        if (huge) {
            b.batch = (int)((size_t)(96 * 1024) / vec_bytes);        // (+ 33 KB of block tables + 16 KB of reduction scratch)
            if (b.batch > max_rank) b.batch = max_rank;
            xv_bytes = (size_t)b.batch * vec_bytes;
        } else {
            b.batch = (int)(tri_bytes / vec_bytes);
        }
        if (b.batch > 1024) b.batch = 1024;
        const size_t lds = (huge ? 0 : tri_bytes) + tables + xv_bytes;
        if (huge) {
            auto k = global_coherence_big_kernel<1024>;
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3((unsigned)slots), dim3(1024), lds, (hipStream_t)stream, b);
        } else if (sc_switch(SC_SW_GLOBAL_NT256)) {      // diagnostic: the round-2 workgroup size
            auto k = global_coherence_big_kernel<256>;
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3((unsigned)slots), dim3(256), lds, (hipStream_t)stream, b);
        } else {
            // 512 threads: 2080 blocks per round at 128 signals are 4 per thread instead of 8 (1024 threads would need
            // 16 KB of reduction scratch next to the 132 KB triangle: over the 160 KB of a CU)
            auto k = global_coherence_big_kernel<512>;
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3((unsigned)slots), dim3(512), lds, (hipStream_t)stream, b);
        }
        const hipError_t e1 = hipGetLastError();
        const hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);       // the log is freed below
        (void)hipFree(log);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            sc_set_error("global coherence (n_signals > 64) failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            return SC_EHIP;
        }
        return SC_OK;
    }
    const size_t lds = (size_t)2 * C * C * sizeof(cd) + (size_t)(M / 2 + 2) * 8 + (size_t)(M / 2) * 16 + (size_t)(M + 2) * 4 +
                       (size_t)C * 8 + (size_t)C * 4 + 64;
    (void)hipFuncSetAttribute((const void*)global_coherence_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(global_coherence_kernel, dim3((unsigned)N, (unsigned)n_groups), dim3(256), lds, (hipStream_t)stream, a);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
