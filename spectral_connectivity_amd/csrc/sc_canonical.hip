// sc_canonical.hip -- canonical coherence between groups of channels from the accumulated
// cross-spectral matrix.
//
// Reference (connectivity.py:745-820, :1953-2032): per group g, whiten the (c_g x n_obs)
// coefficient matrix by its thin SVD (A -> U V^H) and take, per group pair, the squared
// largest singular value of (U V^H)_g (U V^H)_h^H -- n_obs-long SVDs per (window, frequency).
// With S = A A^H / n_obs already accumulated by the MFMA kernel the same quantity is
//     sigma_max( L_g^-1 S_gh L_h^-H )^2,     S_gg = L_g L_g^H (Cholesky),
// (identical when every group has full row rank, n_obs >= c_g), i.e.
// per (bin, group pair) two small Cholesky factorisations, two triangular solves and the top
// eigenvalue of a c_g x c_g Hermitian matrix (cyclic complex Jacobi, eigenvalues only).
// One thread per (bin, group pair), fp64, matrices in per-lane scratch: the work is
// O(bins * pairs * c^3) and tiny next to stage B.
#include <math.h>
#include "sc_common.h"

typedef double2 cd;
__device__ inline cd zmul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline cd zmulc(cd a, cd b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a conj(b)

struct CanonArgs {
    const float* accum;
    const int32_t* members;   // [G][CMAX] channel indices, -1 padded
    const int32_t* sizes;     // [G]
    double* out;              // [n_bins][G][G]
    int32_t* fail;            // [1] count of non positive-definite group blocks
    int64_t n_bins, floats_per_bin;
    int G, n_gpairs, NB, n_tiles, p_csm;
    double n_obs;
};

__device__ inline cd csm_read(const float* rec, const CanonArgs& a, int i, int j) {
    int ti = i >> 4, tj = j >> 4, ii = i & 15, jj = j & 15;
    const bool m = (ti > tj) || (ti == tj && ii > jj);
    if (m) { int t = ti; ti = tj; tj = t; t = ii; ii = jj; jj = t; }
    const int64_t off = ((int64_t)sc_tile_index(ti, tj, a.NB)) * SC_TILE_ELEMS + ii * 16 + jj;
    const double re = (double)rec[(int64_t)a.p_csm * a.n_tiles * SC_TILE_ELEMS + off] / a.n_obs;
    double im = (double)rec[(int64_t)(a.p_csm + 1) * a.n_tiles * SC_TILE_ELEMS + off] / a.n_obs;
    if (m) im = -im;
    if (i == j) im = 0.0;
    return make_double2(re, im);
}

// in-place lower Cholesky of Hermitian positive-definite n x n (row-major, stride CMAX)
template <int CMAX>
__device__ inline bool cholesky(cd (*L)[CMAX], int n) {
    bool ok = true;
    for (int j = 0; j < n; ++j) {
        double d = L[j][j].x;
        for (int k = 0; k < j; ++k) d -= L[j][k].x * L[j][k].x + L[j][k].y * L[j][k].y;
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        const double ljj = sqrt(d);
        L[j][j] = make_double2(ljj, 0.0);
        for (int i = j + 1; i < n; ++i) {
            cd s = L[i][j];
            for (int k = 0; k < j; ++k) { const cd t = zmulc(L[i][k], L[j][k]); s.x -= t.x; s.y -= t.y; }
            L[i][j] = make_double2(s.x / ljj, s.y / ljj);
        }
    }
    return ok;
}

template <int CMAX>
__global__ void __launch_bounds__(64) canonical_kernel(CanonArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (idx >= a.n_bins * a.n_gpairs) return;
    const int64_t bin = idx / a.n_gpairs;
    int gp = (int)(idx - bin * a.n_gpairs);
    int ga = 0, len = a.G - 1;
    while (gp >= len) { gp -= len; ++ga; --len; }
    const int gb = ga + 1 + gp;
    const int na = a.sizes[ga], nb = a.sizes[gb];
    const int32_t* ma = a.members + ga * CMAX;
    const int32_t* mb = a.members + gb * CMAX;
    const float* rec = a.accum + bin * a.floats_per_bin;

    cd La[CMAX][CMAX], Lb[CMAX][CMAX], M[CMAX][CMAX];
    for (int i = 0; i < na; ++i)
        for (int j = 0; j <= i; ++j) La[i][j] = csm_read(rec, a, ma[i], ma[j]);
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j <= i; ++j) Lb[i][j] = csm_read(rec, a, mb[i], mb[j]);
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) M[i][j] = csm_read(rec, a, ma[i], mb[j]);
    const bool ok = cholesky<CMAX>(La, na) & cholesky<CMAX>(Lb, nb);
    if (!ok) atomicAdd(a.fail, 1);
    // M <- La^-1 M  (forward substitution down the rows)
    for (int j = 0; j < nb; ++j)
        for (int i = 0; i < na; ++i) {
            cd s = M[i][j];
            for (int k = 0; k < i; ++k) { const cd t = zmul(La[i][k], M[k][j]); s.x -= t.x; s.y -= t.y; }
            const double d = La[i][i].x;
            M[i][j] = make_double2(s.x / d, s.y / d);
        }
    // M <- M Lb^-H : row vector y = m Lb^-H  <=>  y conj(Lb)^T = m  (forward substitution along columns)
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            cd s = M[i][j];
            for (int k = 0; k < j; ++k) { const cd t = zmulc(M[i][k], Lb[j][k]); s.x -= t.x; s.y -= t.y; }
            const double d = Lb[j][j].x;
            M[i][j] = make_double2(s.x / d, s.y / d);
        }
    // B = M M^H (na x na Hermitian), stored in La
    for (int i = 0; i < na; ++i)
        for (int j = 0; j <= i; ++j) {
            cd s = make_double2(0.0, 0.0);
            for (int k = 0; k < nb; ++k) { const cd t = zmulc(M[i][k], M[j][k]); s.x += t.x; s.y += t.y; }
            La[i][j] = s;
            La[j][i] = make_double2(s.x, -s.y);
        }
    // cyclic complex Jacobi, eigenvalues only
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int p = 0; p < na; ++p) {
            dia += La[p][p].x * La[p][p].x;
            for (int q = p + 1; q < na; ++q) off += La[p][q].x * La[p][q].x + La[p][q].y * La[p][q].y;
        }
        if (off <= 1e-30 * dia || off == 0.0) break;
        for (int p = 0; p < na - 1; ++p)
            for (int q = p + 1; q < na; ++q) {
                const cd bpq = La[p][q];
                const double ab = hypot(bpq.x, bpq.y);
                if (ab < 1e-300) continue;
                const cd e = make_double2(bpq.x / ab, bpq.y / ab);       // e^{i phi}
                const double tau = (La[q][q].x - La[p][p].x) / (2.0 * ab);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + hypot(1.0, tau));
                const double c = 1.0 / hypot(1.0, t), s = t * c;
                const cd se = make_double2(s * e.x, s * e.y), sec = make_double2(s * e.x, -s * e.y);
                for (int k = 0; k < na; ++k) {           // columns p, q
                    const cd cp = La[k][p], cq = La[k][q];
                    const cd t1 = zmul(sec, cq), t2 = zmul(se, cp);
                    La[k][p] = make_double2(c * cp.x - t1.x, c * cp.y - t1.y);
                    La[k][q] = make_double2(t2.x + c * cq.x, t2.y + c * cq.y);
                }
                for (int k = 0; k < na; ++k) {           // rows p, q
                    const cd rp = La[p][k], rq = La[q][k];
                    const cd t1 = zmul(se, rq), t2 = zmul(sec, rp);
                    La[p][k] = make_double2(c * rp.x - t1.x, c * rp.y - t1.y);
                    La[q][k] = make_double2(t2.x + c * rq.x, t2.y + c * rq.y);
                }
            }
    }
    double lmax = La[0][0].x;
    for (int p = 1; p < na; ++p) lmax = fmax(lmax, La[p][p].x);
    if (!ok) lmax = nan("");
    double* o = a.out + bin * a.G * a.G;
    o[ga * a.G + gb] = lmax;
    o[gb * a.G + ga] = lmax;
}

__global__ void canon_fill_nan(double* out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = nan("");
}

extern "C" int sc_canonical_max_group(void) { return 32; }

extern "C" int sc_canonical_coherence_f64(const float* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                                          int64_t n_observations, const int32_t* d_members, const int32_t* d_sizes,
                                          int n_groups, int max_group_size, double* d_out, int32_t* d_fail,
                                          void* stream) {
    SC_REQUIRE(d_accum && d_members && d_sizes && d_out && d_fail, "NULL argument");
    SC_REQUIRE(planes & SC_PLANE_CSM, "accumulator record must contain SC_PLANE_CSM");
    SC_REQUIRE(n_groups >= 1 && n_bins >= 1, "empty problem");
    if (max_group_size > 32) {
        sc_set_error("canonical coherence supports groups of at most 32 channels (got %d)", max_group_size);
        return SC_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    CanonArgs a;
    a.accum = d_accum; a.members = d_members; a.sizes = d_sizes; a.out = d_out; a.fail = d_fail;
    a.n_bins = n_bins; a.G = n_groups; a.n_gpairs = n_groups * (n_groups - 1) / 2;
    a.NB = sc_n_blocks(n_signals); a.n_tiles = sc_n_tiles(a.NB);
    a.floats_per_bin = (int64_t)sc_plane_count(planes) * a.n_tiles * SC_TILE_ELEMS;
    a.p_csm = sc_plane_offset(planes, SC_PLANE_CSM);
    a.n_obs = (double)n_observations;
    const int64_t total_out = n_bins * n_groups * n_groups;
    hipLaunchKernelGGL(canon_fill_nan, dim3((unsigned)((total_out + 255) / 256)), dim3(256), 0, st, d_out, total_out);
    hipMemsetAsync(d_fail, 0, 4, st);
    const int64_t threads = n_bins * a.n_gpairs;
    if (threads > 0) {
        const unsigned blocks = (unsigned)((threads + 63) / 64);
        // members stride must match the instantiated CMAX
        if (max_group_size <= 16)
            hipLaunchKernelGGL(canonical_kernel<16>, dim3(blocks), dim3(64), 0, st, a);
        else
            hipLaunchKernelGGL(canonical_kernel<32>, dim3(blocks), dim3(64), 0, st, a);
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
