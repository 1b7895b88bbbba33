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
#include "sc_common.h"
#include "sc_jacobi.h"

typedef double2 cd;

#define GC_CMAX 64

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

extern "C" int sc_global_coherence_max_signals(void) { return GC_HUGE_CMAX; }

extern "C" int sc_global_coherence_f64(const void* d_accum, int64_t n_groups, int64_t n_freq_accum, int64_t N,
                                       int64_t C, uint32_t planes, int64_t n_obs, int max_rank, int ascending,
                                       double* d_values, void* d_vectors, void* stream) {
    ScTimed timed_("global_coherence", stream);
    SC_REQUIRE(d_accum && d_values && d_vectors, "NULL argument");
    SC_REQUIRE(planes & SC_PLANE_CSM, "accumulator record must contain SC_PLANE_CSM");
    SC_REQUIRE(n_freq_accum == N || n_freq_accum == N / 2 + 1, "accumulators must hold N or N/2+1 bins");
    SC_REQUIRE(n_groups >= 1 && n_groups <= 65535 && N >= 1 && n_obs >= 1, "bad problem size");
    if (C < 1 || C > GC_HUGE_CMAX) {
        sc_set_error("global coherence: n_signals <= %d (got %lld)", GC_HUGE_CMAX, (long long)C);
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
        } else {
            auto k = global_coherence_big_kernel<256>;
            (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3((unsigned)slots), dim3(256), lds, (hipStream_t)stream, b);
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
