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
#include <stdlib.h>
#include "sc_common.h"
#include "sc_jacobi.h"

typedef double2 cd;
__device__ inline cd zmul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline cd zmulc(cd a, cd b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a conj(b)

struct CanonArgs {
    ScRec accum;
    const int32_t* members;   // [G][CMAX] channel indices, -1 padded
    const int32_t* sizes;     // [G]
    double* out;              // [n_bins][G][G]
    int32_t* fail;            // [1] count of non positive-definite group blocks
    int64_t n_bins, floats_per_bin;
    int G, n_gpairs, NB, n_tiles, p_csm;
    int mstride;              // row length of `members` for the workgroup-per-problem kernels (32 or CBIG_C)
    double jtol;
    double n_obs;
};

__device__ inline cd csm_read(ScRec rec, const CanonArgs& a, int i, int j) {
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
    const ScRec rec = a.accum + bin * a.floats_per_bin;

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

// ---- groups of at most 16 channels: factor kernel + pair kernel ------------------------------------
// The thread-per-problem kernel above keeps three 16x16 complex matrices per lane in scratch memory and
// factors every group's block once per PAIR.  Here
//   canonical_factor_kernel  one workgroup per bin, a wave per group: Cholesky factor, inverted in place
//                            (so whitening a pair is two dense 16x16 products with no dependent chains),
//                            written to a [bin][group][16][16] workspace (L2-resident: 4 KB per group);
//   canonical_pair_kernel    a WAVE per (bin, group pair), eight to a workgroup: whitening, B = M M^H
//                            (a lane per entry) and a parallel cyclic Jacobi on B (a lane per 2x2 block of
//                            the round's pairing) in the wave's own 8 KB of LDS, ordered by the wave's
//                            in-order LDS pipe -- no workgroup barrier at all.
// Pair-granular workgroups (7695 at cfg5) fill the 256 CUs evenly and two fit a CU; one workgroup per bin
// with the factors kept in LDS (the first version of this path) left a 513-bin problem waiting for a third
// generation of a single workgroup.
#define CB_C 16
#define CB_WSYNC()                                                  \
    do {                                                            \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
        __builtin_amdgcn_wave_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      \
    } while (0)

__device__ inline cd cb_get(const cd* B, int i, int j) {           // Hermitian, upper triangle stored
    if (i <= j) return B[i * CB_C + j];
    const cd v = B[j * CB_C + i];
    return make_double2(v.x, -v.y);
}
__device__ inline void cb_set(cd* B, int i, int j, cd v) {
    if (i <= j) B[i * CB_C + j] = v;
    else B[j * CB_C + i] = make_double2(v.x, -v.y);
}
__device__ inline double cb_wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

#define CB_WAVES 8
#ifndef CB_SWEEPS
#define CB_SWEEPS 12
#endif
__global__ void __launch_bounds__(64 * CB_WAVES) canonical_factor_kernel(CanonArgs a, cd* Lg, int* okb) {
    extern __shared__ __align__(16) unsigned char cb_smem[];
    const int G = a.G;
    cd* Ls = reinterpret_cast<cd*>(cb_smem);                                   // [wave][16][16]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t bin = blockIdx.x;
    const ScRec rec = a.accum + bin * a.floats_per_bin;

    // phase 1: L_g for every group (a wave per group, left-looking Cholesky, a lane per row)
    for (int g = wave; g < G; g += CB_WAVES) {
        const int n = a.sizes[g];
        const int32_t* mg = a.members + g * CB_C;
        cd* L = Ls + (size_t)wave * CB_C * CB_C;
        for (int e = lane; e < CB_C * CB_C; e += 64) {
            const int i = e / CB_C, j = e % CB_C;
            L[e] = (i < n && j < n) ? csm_read(rec, a, mg[i], mg[j]) : make_double2(0.0, 0.0);
        }
        CB_WSYNC();
        int ok = 1;
        for (int j = 0; j < n; ++j) {
            double d = L[j * CB_C + j].x;                                   // every lane: the pivot of column j
            for (int k = 0; k < j; ++k) { const cd v = L[j * CB_C + k]; d -= v.x * v.x + v.y * v.y; }
            if (!(d > 0.0)) { ok = 0; d = 1.0; }
            const double ljj = sqrt(d);
            const int i = j + 1 + lane;
            cd s = make_double2(0.0, 0.0);
            if (i < n) {
                s = L[i * CB_C + j];
                for (int k = 0; k < j; ++k) { const cd t = zmulc(L[i * CB_C + k], L[j * CB_C + k]); s.x -= t.x; s.y -= t.y; }
            }
            CB_WSYNC();
            if (lane == 0) L[j * CB_C + j] = make_double2(ljj, 0.0);
            if (i < n) L[i * CB_C + j] = make_double2(s.x / ljj, s.y / ljj);
            CB_WSYNC();
        }
        // L <- L^-1 (lower triangular): the whitening of a pair is then two dense 16x16 products with no dependent
        // chains instead of two triangular solves per pair.  Column `lane` of the inverse, top to bottom, kept in
        // registers until every lane is done with L.
        cd inv[CB_C];
#pragma unroll
        for (int i = 0; i < CB_C; ++i) inv[i] = make_double2(0.0, 0.0);
        if (lane < n) {
#pragma unroll
            for (int i = 0; i < CB_C; ++i) {
                if (i >= lane && i < n) {
                    cd sv = make_double2(i == lane ? 1.0 : 0.0, 0.0);
#pragma unroll
                    for (int k = 0; k < CB_C; ++k)
                        if (k >= lane && k < i) { const cd t = zmul(L[i * CB_C + k], inv[k]); sv.x -= t.x; sv.y -= t.y; }
                    const double d = L[i * CB_C + i].x;
                    inv[i] = make_double2(sv.x / d, sv.y / d);
                }
            }
        }
        CB_WSYNC();
        if (lane < CB_C) {
#pragma unroll
            for (int i = 0; i < CB_C; ++i) L[i * CB_C + lane] = (lane < n && i < n) ? inv[i] : make_double2(0.0, 0.0);
        }
        CB_WSYNC();
        cd* out = Lg + ((size_t)bin * G + g) * CB_C * CB_C;
        for (int e = lane; e < CB_C * CB_C; e += 64) out[e] = L[e];
        if (lane == 0) okb[bin * G + g] = ok;
        CB_WSYNC();                                                // L is reused for this wave's next group
    }
}

__global__ void __launch_bounds__(64 * CB_WAVES) canonical_pair_kernel(CanonArgs a, const cd* Lg, const int* okb) {
    extern __shared__ __align__(16) unsigned char cb_smem[];
    const int G = a.G;
    cd* scratch = reinterpret_cast<cd*>(cb_smem);                              // per wave: M, B [16][16]
    double* rot = reinterpret_cast<double*>(scratch + CB_WAVES * 2 * CB_C * CB_C);   // per wave: [8] cos, [8][2] sin
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_chunks = (a.n_gpairs + CB_WAVES - 1) / CB_WAVES;
    const int64_t bin = blockIdx.x / n_chunks;
    const int chunk = blockIdx.x % n_chunks;
    const ScRec rec = a.accum + bin * a.floats_per_bin;
    const cd* Ls = Lg + (size_t)bin * G * CB_C * CB_C;
    const int* okg = okb + bin * G;
    cd* M = scratch + (size_t)wave * 2 * CB_C * CB_C;
    cd* B = M + CB_C * CB_C;
    double* rc = rot + wave * 24;
    cd* rs = reinterpret_cast<cd*>(rc + 8);
    {
        const int pr = chunk * CB_WAVES + wave;
        if (pr >= a.n_gpairs) return;                              // no workgroup barriers below
        int gp = pr, ga = 0, len = G - 1;
        while (gp >= len) { gp -= len; ++ga; --len; }
        const int gb = ga + 1 + gp;
        const int na = a.sizes[ga], nb = a.sizes[gb];
        const int32_t* ma = a.members + ga * CB_C;
        const int32_t* mb = a.members + gb * CB_C;
        const cd* La = Ls + (size_t)ga * CB_C * CB_C;
        const cd* Lb = Ls + (size_t)gb * CB_C * CB_C;
        const bool ok = okg[ga] && okg[gb];
        for (int e = lane; e < CB_C * CB_C; e += 64) {
            const int i = e / CB_C, j = e % CB_C;
            M[e] = (i < na && j < nb) ? csm_read(rec, a, ma[i], mb[j]) : make_double2(0.0, 0.0);
        }
        CB_WSYNC();
        // M <- Linv_a M Linv_b^H as two dense products through B.  The inverse factors come from the workspace (L2):
        // a row of sixteen independent loads per output element, issued together (their upper triangles are zero).
        for (int e = lane; e < CB_C * CB_C; e += 64) {
            const int i = e / CB_C, j = e % CB_C;
            cd la[CB_C];
#pragma unroll
            for (int k = 0; k < CB_C; ++k) la[k] = La[i * CB_C + k];
            cd sv = make_double2(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < CB_C; ++k) { const cd t = zmul(la[k], M[k * CB_C + j]); sv.x += t.x; sv.y += t.y; }
            B[e] = sv;
        }
        CB_WSYNC();
        for (int e = lane; e < CB_C * CB_C; e += 64) {
            const int i = e / CB_C, j = e % CB_C;
            cd lb[CB_C];
#pragma unroll
            for (int k = 0; k < CB_C; ++k) lb[k] = Lb[j * CB_C + k];
            cd sv = make_double2(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < CB_C; ++k) { const cd t = zmulc(B[i * CB_C + k], lb[k]); sv.x += t.x; sv.y += t.y; }
            M[e] = sv;
        }
        CB_WSYNC();
        for (int e = lane; e < CB_C * CB_C; e += 64) {      // B = M M^H, both triangles, zero outside na x na
            const int i = e / CB_C, j = e % CB_C;
            cd sv = make_double2(0.0, 0.0);
            if (i < na && j < na) {
#pragma unroll
                for (int k = 0; k < CB_C; ++k) { const cd t = zmulc(M[i * CB_C + k], M[j * CB_C + k]); sv.x += t.x; sv.y += t.y; }
                if (i == j) sv.y = 0.0;
            }
            B[e] = sv;
        }
        CB_WSYNC();
        // Parallel cyclic Jacobi, eigenvalues only.  Round r of a sweep pairs the Mp indices by the circle method
        // (index Mp-1 fixed, the others rotating); pair k of the round is rotated by lane k, and lane (u, v) applies
        // B' = J_u^H B J_v to the 2x2 block (pair u) x (pair v).  An odd na pairs one
        // index with the zero row na: zero pivot, identity rotation, no special case.
        const int Mp = na + (na & 1), H = Mp / 2, Q = Mp - 1;
        const bool has_block = lane < H * H;               // every block of the H x H grid has a lane: no mirror writes
        const int bu = has_block ? lane / H : 0, bv = has_block ? lane % H : 0;
        // circle method, kept incrementally: pair k of round r is (r + k, r - k) mod Q, pair 0 is (Q, r); the
        // positions advance by one per round and return to the start after the Q rounds of a sweep
        auto start_x = [Q](int k) { return k == 0 ? Q : k; };
        auto start_y = [Q](int k) { return k == 0 ? 0 : Q - k; };
        auto advance = [Q](int k, int& x, int& y) {
            const int x1 = x + 1, y1 = y + 1;
            if (k != 0) x = x1 == Q ? 0 : x1;
            y = y1 == Q ? 0 : y1;
        };
        int xa = start_x(lane < H ? lane : 0), ya = start_y(lane < H ? lane : 0);
        int xu = start_x(bu), yu = start_y(bu), xv = start_x(bv), yv = start_y(bv);
        for (int sweep = 0; sweep < CB_SWEEPS && na > 1; ++sweep) {
            double off = 0.0, dia = 0.0;
            for (int e = lane; e < CB_C * CB_C; e += 64) {
                const int i = e / CB_C, j = e % CB_C;
                const double v = B[e].x * B[e].x + B[e].y * B[e].y;
                if (i == j) dia += v; else off += 0.5 * v;
            }
            off = cb_wave_sum(off); dia = cb_wave_sum(dia);
            if (off <= a.jtol * dia || off == 0.0) break;
            for (int r = 0; r < Q; ++r) {
                if (lane < H) {
                    const int pi = xa < ya ? xa : ya, qi = xa < ya ? ya : xa;
                    advance(lane, xa, ya);
                    double c = 1.0;
                    cd se = make_double2(0.0, 0.0);
                    // Rotation angle in f32 with the hardware's 1-ulp rcp / rsq (no IEEE division sequences, no fp64
                    // Newton steps on 8 lanes), then (c, s) re-normalised in fp64 to first order: the rotation is
                    // unitary to 1e-14, it merely leaves ~1e-7 of the pivot behind, which the next sweep removes.
                    const cd bb = B[pi * CB_C + qi];
                    const float bx = (float)bb.x, by = (float)bb.y;
                    const float ab2 = bx * bx + by * by;
                    if (ab2 > 1e-37f) {
                        const float rab = __builtin_amdgcn_rsqf(ab2);                       // 1 / |b|
                        const float tau = 0.5f * (float)(B[qi * CB_C + qi].x - B[pi * CB_C + pi].x) * rab;
                        const float den = fabsf(tau) + __builtin_amdgcn_sqrtf(1.f + tau * tau);
                        const float t = __builtin_copysignf(__builtin_amdgcn_rcpf(den), tau);
                        const float cf = __builtin_amdgcn_rsqf(1.f + t * t);
                        const float sn = t * cf * rab;
                        const double cd0 = (double)cf, sx = (double)(sn * bx), sy = (double)(sn * by);
                        const double fix = 1.5 - 0.5 * (cd0 * cd0 + sx * sx + sy * sy);     // 1 / sqrt(r), r = 1 + O(1e-7)
                        c = cd0 * fix;
                        se = make_double2(sx * fix, sy * fix);
                    }
                    rc[lane] = c; rs[lane] = se;
                }
                CB_WSYNC();
                const int up = xu < yu ? xu : yu, uq = xu < yu ? yu : xu, vp = xv < yv ? xv : yv, vq = xv < yv ? yv : xv;
                advance(bu, xu, yu);
                advance(bv, xv, yv);
                cd n00, n01, n10, n11;
                if (has_block) {
                    const double cu = rc[bu], cv = rc[bv];
                    const cd su = rs[bu], sv = rs[bv];
                    const cd b00 = B[up * CB_C + vp], b01 = B[up * CB_C + vq], b10 = B[uq * CB_C + vp], b11 = B[uq * CB_C + vq];
                    const cd svc = make_double2(sv.x, -sv.y), suc = make_double2(su.x, -su.y);
                    const cd a0 = zmul(svc, b01), a1 = zmul(sv, b00), a2 = zmul(svc, b11), a3 = zmul(sv, b10);
                    const cd t00 = make_double2(cv * b00.x - a0.x, cv * b00.y - a0.y);
                    const cd t01 = make_double2(a1.x + cv * b01.x, a1.y + cv * b01.y);
                    const cd t10 = make_double2(cv * b10.x - a2.x, cv * b10.y - a2.y);
                    const cd t11 = make_double2(a3.x + cv * b11.x, a3.y + cv * b11.y);
                    const cd c0 = zmul(su, t10), c1 = zmul(su, t11), c2 = zmul(suc, t00), c3 = zmul(suc, t01);
                    n00 = make_double2(cu * t00.x - c0.x, cu * t00.y - c0.y);
                    n01 = make_double2(cu * t01.x - c1.x, cu * t01.y - c1.y);
                    n10 = make_double2(c2.x + cu * t10.x, c2.y + cu * t10.y);
                    n11 = make_double2(c3.x + cu * t11.x, c3.y + cu * t11.y);
                }
                CB_WSYNC();                            // every block has read its inputs before anyone writes
                if (has_block) {
                    if (bu == bv) {
                        B[up * CB_C + up] = make_double2(n00.x, 0.0);
                        B[uq * CB_C + uq] = make_double2(n11.x, 0.0);
                        B[up * CB_C + uq] = make_double2(0.0, 0.0);
                        B[uq * CB_C + up] = make_double2(0.0, 0.0);
                    } else {
                        B[up * CB_C + vp] = n00;
                        B[up * CB_C + vq] = n01;
                        B[uq * CB_C + vp] = n10;
                        B[uq * CB_C + vq] = n11;
                    }
                }
                CB_WSYNC();
            }
        }
        if (lane == 0) {
            double lmax = B[0].x;
            for (int q = 1; q < na; ++q) lmax = fmax(lmax, B[q * CB_C + q].x);
            if (!ok) { lmax = nan(""); atomicAdd(a.fail, 1); }
            double* o = a.out + bin * G * G;
            o[ga * G + gb] = lmax;
            o[gb * G + ga] = lmax;
        }
    }
}

// ---- the same pair problem without the Jacobi sweeps (round 5) ----------------------------------------------------------
// The parallel Jacobi above spends ~7 sweeps x 15 rounds x ~100 wave instructions per problem on ALL sixteen eigenvalues of
// B = M M^H (2.9 ms for the 61 560 problems of BASELINE configs[4]: fp64 issue-bound); only the largest one is asked for.
// Here B lives in REGISTERS (lane (i, jq) = row i, columns 4 jq .. 4 jq + 3: four complex entries), is reduced to a real symmetric
// tridiagonal matrix by fourteen Householder reflections (LAPACK zhetd2, lower form: p = tau B v, w = p - (tau / 2)(p^H v) v,
// B <- B - v w^H - w v^H; v and w travel through 512 bytes of LDS, sums over rows / column groups by lane shuffles), and the
// largest eigenvalue of T is bracketed by multisection on the Sturm sequence: 64 shifts per round, one per lane (x > lambda_max
// iff every leading principal minor of T - x changes sign), nine rounds from the Gershgorin bracket to the last bits.  Only |e_k|^2
// of the subdiagonal is needed, so a column that is already reduced needs no reflection.  LDS: M alone (4 KB a wave).
__device__ inline double cb_sum16(double v) {            // over the sixteen lanes of a row group, result in all of them
    for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ inline double cb_sum4(double v) {             // over the four row groups
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ inline double cb_lane(double v, int lane) {    // (lane: compile-time constant after unrolling)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// largest eigenvalue of the Hermitian positive semi-definite 16 x 16 matrix held as a[c] = B[i][4 jq + c], i = lane & 15, jq = lane >> 4
// (rows / columns from the matrix order on: zero); vb: 32 complex numbers of this wave's LDS
__device__ double cb_top_eigenvalue(cd (&a)[4], cd* vb, int lane) {
    const int i = lane & 15, jq = lane >> 4;
    double d[16], e2[15];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
        const int kq = k >> 2, src = 16 * kq;             // column k lives in row group kq, register k & 3
        const bool owner = jq == kq;
        const cd x = a[k & 3];
        d[k] = cb_lane(x.x, src + k);
        const double alr = cb_lane(x.x, src + k + 1), ali = cb_lane(x.y, src + k + 1);
        double s2 = (owner && i > k + 1) ? x.x * x.x + x.y * x.y : 0.0;
        s2 = cb_sum16(s2);
        const double xn2 = cb_lane(s2, src);
        e2[k] = alr * alr + ali * ali + xn2;
        if (xn2 == 0.0) continue;                         // (wave-uniform) nothing below the subdiagonal: no reflection
        const double beta = alr > 0.0 ? -sqrt(e2[k]) : sqrt(e2[k]), rb = 1.0 / beta;
        const cd tau = make_double2((beta - alr) * rb, -ali * rb);
        const double dr = alr - beta, den = 1.0 / (dr * dr + ali * ali);
        const cd sc = make_double2(dr * den, -ali * den);                       // 1 / (alpha - beta)
        if (owner) vb[i] = i > k + 1 ? zmul(x, sc) : (i == k + 1 ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0));
        CB_WSYNC();
        const cd vi = vb[i];
        cd vc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) vc[c] = vb[4 * jq + c];
        cd acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int c = 0; c < 4; ++c) { const cd t = zmul(a[c], vc[c]); acc.x += t.x; acc.y += t.y; }
        acc.x = cb_sum4(acc.x); acc.y = cb_sum4(acc.y);
        cd pv = zmul(tau, acc);
        if (i <= k) pv = make_double2(0.0, 0.0);
        const double kx = cb_sum16(pv.x * vi.x + pv.y * vi.y), ky = cb_sum16(pv.x * vi.y - pv.y * vi.x);      // p^H v
        const cd hk = zmul(make_double2(0.5 * tau.x, 0.5 * tau.y), make_double2(kx, ky));
        const cd t0 = zmul(hk, vi);
        const cd w = make_double2(pv.x - t0.x, pv.y - t0.y);
        if (jq == 0) vb[16 + i] = w;
        CB_WSYNC();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const cd wc = vb[16 + 4 * jq + c];
            const cd u0 = zmulc(vi, wc), u1 = zmulc(w, vc[c]);
            a[c].x -= u0.x + u1.x;
            a[c].y -= u0.y + u1.y;
        }
        CB_WSYNC();                                       // (v and w are read: the next step may write them)
    }
    d[14] = cb_lane(a[2].x, 48 + 14);
    d[15] = cb_lane(a[3].x, 48 + 15);
    {
        const double er = cb_lane(a[2].x, 48 + 15), ei = cb_lane(a[2].y, 48 + 15);
        e2[14] = er * er + ei * ei;
    }
    double dmax = d[0], emax = e2[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) dmax = fmax(dmax, d[q]);
#pragma unroll
    for (int q = 1; q < 15; ++q) emax = fmax(emax, e2[q]);
    const double top = dmax + 2.0 * sqrt(emax);           // Gershgorin; lambda_max >= the largest diagonal entry
    if (!(top > 0.0)) return dmax > 0.0 ? dmax : 0.0;
    const double sc1 = 1.0 / top, sc2 = sc1 * sc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) d[q] *= sc1;
#pragma unroll
    for (int q = 0; q < 15; ++q) e2[q] *= sc2;
    double lo = dmax * sc1, hi = 1.0;
    for (int round = 0; round < 9 && hi > lo; ++round) {
        const double x = lane == 63 ? hi : lo + (hi - lo) * ((double)(lane + 1) * (1.0 / 64.0));
        double p0 = 1.0, p1 = d[0] - x;
        bool above = p1 < 0.0;
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const double p2 = (d[q] - x) * p1 - e2[q - 1] * p0;
            above = above && (p2 * p1 < 0.0);
            p0 = p1; p1 = p2;
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(above);
        if (mask == 0ull) break;                          // (cannot happen for x = hi; rounding at the last bits)
        const int b = __builtin_ctzll(mask);
        const double xh = __shfl(x, b), xl = __shfl(x, b > 0 ? b - 1 : 0);
        hi = xh;
        if (b > 0) lo = xl;
    }
    return 0.5 * (lo + hi) * top;
}

__global__ void __launch_bounds__(64 * CB_WAVES) canonical_pair_hh_kernel(CanonArgs a, const cd* Lg, const int* okb) {
    extern __shared__ __align__(16) unsigned char cb_smem[];
    const int G = a.G;
    cd* scratch = reinterpret_cast<cd*>(cb_smem);                              // per wave: M, T [16][16]; v, w [32]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_chunks = (a.n_gpairs + CB_WAVES - 1) / CB_WAVES;
    const int64_t bin = blockIdx.x / n_chunks;
    const int chunk = blockIdx.x % n_chunks;
    const ScRec rec = a.accum + bin * a.floats_per_bin;
    const cd* Ls = Lg + (size_t)bin * G * CB_C * CB_C;
    const int* okg = okb + bin * G;
    cd* M = scratch + (size_t)wave * (2 * CB_C * CB_C + 32);
    cd* T = M + CB_C * CB_C;
    cd* vb = T + CB_C * CB_C;
    const int pr = chunk * CB_WAVES + wave;
    if (pr >= a.n_gpairs) return;                                  // no workgroup barriers below
    int gp = pr, ga = 0, len = G - 1;
    while (gp >= len) { gp -= len; ++ga; --len; }
    const int gb = ga + 1 + gp;
    const int na = a.sizes[ga], nb = a.sizes[gb];
    const int32_t* ma = a.members + ga * CB_C;
    const int32_t* mb = a.members + gb * CB_C;
    const cd* La = Ls + (size_t)ga * CB_C * CB_C;
    const cd* Lb = Ls + (size_t)gb * CB_C * CB_C;
    const bool ok = okg[ga] && okg[gb];
    for (int e = lane; e < CB_C * CB_C; e += 64) {
        const int i = e / CB_C, j = e % CB_C;
        M[e] = (i < na && j < nb) ? csm_read(rec, a, ma[i], mb[j]) : make_double2(0.0, 0.0);
    }
    CB_WSYNC();
    // M <- Linv_a M Linv_b^H as two dense products through T (the inverse factors come from the workspace: L2)
    for (int e = lane; e < CB_C * CB_C; e += 64) {
        const int i = e / CB_C, j = e % CB_C;
        cd la[CB_C];
#pragma unroll
        for (int k = 0; k < CB_C; ++k) la[k] = La[i * CB_C + k];
        cd sv = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < CB_C; ++k) { const cd t = zmul(la[k], M[k * CB_C + j]); sv.x += t.x; sv.y += t.y; }
        T[e] = sv;
    }
    CB_WSYNC();
    for (int e = lane; e < CB_C * CB_C; e += 64) {
        const int i = e / CB_C, j = e % CB_C;
        cd lb[CB_C];
#pragma unroll
        for (int k = 0; k < CB_C; ++k) lb[k] = Lb[j * CB_C + k];
        cd sv = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < CB_C; ++k) { const cd t = zmulc(T[i * CB_C + k], lb[k]); sv.x += t.x; sv.y += t.y; }
        M[e] = sv;
    }
    CB_WSYNC();
    // B = M M^H straight into the registers of the reduction: lane (i, jq) holds B[i][4 jq .. 4 jq + 3] (zero outside na x na)
    cd breg[4];
    {
        const int i = lane & 15, jq = lane >> 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = 4 * jq + c;
            cd sv = make_double2(0.0, 0.0);
            if (i < na && j < na) {
#pragma unroll
                for (int k = 0; k < CB_C; ++k) { const cd t = zmulc(M[i * CB_C + k], M[j * CB_C + k]); sv.x += t.x; sv.y += t.y; }
                if (i == j) sv.y = 0.0;
            }
            breg[c] = sv;
        }
    }
    double lmax = cb_top_eigenvalue(breg, vb, lane);
    if (lane == 0) {
        if (!ok) { lmax = nan(""); atomicAdd(a.fail, 1); }
        double* o = a.out + bin * G * G;
        o[ga * G + gb] = lmax;
        o[gb * G + ga] = lmax;
    }
}

__global__ void canon_fill_nan(double* out, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = nan("");
}

// ---- groups of 33 ... 128 channels: a workgroup per (bin, group pair) -------------------------------------------------
// Three C x C matrices per problem no longer fit a lane's scratch (or the LDS): the group blocks S_aa, S_bb and the cross
// block S_ab of a problem live in a per-workgroup global scratch (L2-resident: 3 x 256 KB at 128 channels), factored and
// whitened by the workgroup's 256 threads in place -- right-looking Cholesky (one column per step), M <- L_a^-1 M with a
// thread per column, M <- M L_b^-H with a thread per row --, then B = M M^H goes into LDS as a packed upper triangle and
// the parallel cyclic Jacobi of the global-coherence kernel (sc_jacobi.h) diagonalises it: its largest diagonal entry
// is the squared canonical coherence.  Persistent workgroups (one per CU) walk the (bin, pair) list.
#define CBIG_C 128
#define CBIG_SWEEPS 14

// in-place lower Cholesky of the n x n Hermitian matrix L (row-major, stride CBIG_C, lower triangle valid), all 256 threads
__device__ inline bool cbig_cholesky(cd* L, int n, int* bad) {
    const int tid = threadIdx.x;
    for (int k = 0; k < n; ++k) {
        if (tid == 0) {
            const double d = L[k * CBIG_C + k].x;
            if (!(d > 0.0)) *bad = 1;
            L[k * CBIG_C + k] = make_double2(sqrt(d > 0.0 ? d : 1.0), 0.0);
        }
        __syncthreads();
        const double dk = L[k * CBIG_C + k].x;
        for (int i = k + 1 + tid; i < n; i += 256) {
            const cd v = L[i * CBIG_C + k];
            L[i * CBIG_C + k] = make_double2(v.x / dk, v.y / dk);
        }
        __syncthreads();
        // trailing update of the lower triangle: L[i][j] -= L[i][k] conj(L[j][k]), k < j <= i
        const int m = n - k - 1;
        for (int e = tid; e < m * m; e += 256) {
            const int i = k + 1 + e / m, j = k + 1 + e % m;
            if (j <= i) {
                const cd t = zmulc(L[i * CBIG_C + k], L[j * CBIG_C + k]);
                cd v = L[i * CBIG_C + j];
                v.x -= t.x; v.y -= t.y;
                L[i * CBIG_C + j] = v;
            }
        }
        __syncthreads();
    }
    return *bad == 0;
}

__global__ void __launch_bounds__(256) canonical_big_kernel(CanonArgs a, cd* scratch, int64_t n_items) {
    extern __shared__ __align__(16) unsigned char cb_smem[];
    cd* B = reinterpret_cast<cd*>(cb_smem);                                  // packed upper triangle, <= 128 x 129 / 2
    constexpr int HM = CBIG_C / 2;
    double* rc = reinterpret_cast<double*>(B + (size_t)CBIG_C * (CBIG_C + 1) / 2);    // [HM]
    cd* rs = reinterpret_cast<cd*>(rc + HM);                                 // [HM]
    int* rp = reinterpret_cast<int*>(rs + HM);                               // [2 HM]
    unsigned short* blk_u = reinterpret_cast<unsigned short*>(rp + 2 * HM);  // [HM (HM + 1) / 2]
    unsigned short* blk_v = blk_u + HM * (HM + 1) / 2;
    __shared__ double red[2][256];
    __shared__ int done, n_rounds, bad, tab_for;
    const int tid = threadIdx.x;
    cd* La = scratch + (size_t)blockIdx.x * 3 * CBIG_C * CBIG_C;
    cd* Lb = La + CBIG_C * CBIG_C;
    cd* M = Lb + CBIG_C * CBIG_C;
    if (tid == 0) tab_for = -1;
    __syncthreads();
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int64_t bin = item / a.n_gpairs;
        int gp = (int)(item - bin * a.n_gpairs);
        int ga = 0, len = a.G - 1;
        while (gp >= len) { gp -= len; ++ga; --len; }
        int gb = ga + 1 + gp;
        // the Jacobi runs on the SMALLER group's side: B = M M^H is n_a x n_a with the same non-zero spectrum either way
        if (a.sizes[gb] < a.sizes[ga]) { const int t = ga; ga = gb; gb = t; }
        const int na = a.sizes[ga], nb = a.sizes[gb];
        const int32_t* ma = a.members + ga * a.mstride;
        const int32_t* mb = a.members + gb * a.mstride;
        const ScRec rec = a.accum + bin * a.floats_per_bin;
        if (tid == 0) bad = 0;
        for (int e = tid; e < na * na; e += 256) { const int i = e / na, j = e % na; if (j <= i) La[i * CBIG_C + j] = csm_read(rec, a, ma[i], ma[j]); }
        for (int e = tid; e < nb * nb; e += 256) { const int i = e / nb, j = e % nb; if (j <= i) Lb[i * CBIG_C + j] = csm_read(rec, a, mb[i], mb[j]); }
        for (int e = tid; e < na * nb; e += 256) { const int i = e / nb, j = e % nb; M[i * CBIG_C + j] = csm_read(rec, a, ma[i], mb[j]); }
        __syncthreads();
        cbig_cholesky(La, na, &bad);
        cbig_cholesky(Lb, nb, &bad);
        // M <- La^-1 M: thread j owns column j (forward substitution down the rows)
        for (int j = tid; j < nb; j += 256)
            for (int i = 0; i < na; ++i) {
                cd sacc = M[i * CBIG_C + j];
                for (int k = 0; k < i; ++k) { const cd t = zmul(La[i * CBIG_C + k], M[k * CBIG_C + j]); sacc.x -= t.x; sacc.y -= t.y; }
                const double d = La[i * CBIG_C + i].x;
                M[i * CBIG_C + j] = make_double2(sacc.x / d, sacc.y / d);
            }
        __syncthreads();
        // M <- M Lb^-H: thread i owns row i (forward substitution along the columns)
        for (int i = tid; i < na; i += 256)
            for (int j = 0; j < nb; ++j) {
                cd sacc = M[i * CBIG_C + j];
                for (int k = 0; k < j; ++k) { const cd t = zmulc(M[i * CBIG_C + k], Lb[j * CBIG_C + k]); sacc.x -= t.x; sacc.y -= t.y; }
                const double d = Lb[j * CBIG_C + j].x;
                M[i * CBIG_C + j] = make_double2(sacc.x / d, sacc.y / d);
            }
        __syncthreads();
        // B = M M^H, packed upper triangle in LDS
        for (int e = tid; e < na * na; e += 256) {
            const int i = e / na, j = e % na;
            if (i > j) continue;
            cd sacc = make_double2(0.0, 0.0);
            for (int k = 0; k < nb; ++k) { const cd t = zmulc(M[i * CBIG_C + k], M[j * CBIG_C + k]); sacc.x += t.x; sacc.y += t.y; }
            if (i == j) sacc.y = 0.0;
            B[gc_tri(i, j, na)] = sacc;
        }
        const int H = (na + (na & 1)) / 2;
        if (tab_for != H) {                        // pairing table of this group size (rebuilt only when the size changes)
            __syncthreads();
            gc_block_table(blk_u, blk_v, H, tid);
            if (tid == 0) tab_for = H;
        }
        __syncthreads();
        gc_jacobi(B, na, rc, rs, rp, blk_u, blk_v, red, &done, &n_rounds, nullptr, CBIG_SWEEPS);
        __syncthreads();
        double lmax = -1.0;
        for (int i = tid; i < na; i += 256) lmax = fmax(lmax, B[gc_tri(i, i, na)].x);
        red[0][tid] = lmax;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[0][tid] = fmax(red[0][tid], red[0][tid + s2]);
            __syncthreads();
        }
        if (tid == 0) {
            double v = red[0][0];
            if (bad) { v = nan(""); atomicAdd(a.fail, 1); }
            double* o = a.out + bin * a.G * a.G;
            o[ga * a.G + gb] = v;
            o[gb * a.G + ga] = v;
        }
        __syncthreads();
    }
}

// ---- the same problem with the largest eigenvalue alone (round 6) -------------------------------------------------------------
// canonical_big_kernel spends most of a problem in the parallel Jacobi (14 sweeps x (n - 1) rounds over the packed triangle:
// 2 ms of 3.6 at 64 channels, 21 ms a problem at 128) for ONE number, the largest eigenvalue of B = M M^H.  Here B is reduced
// to a real symmetric tridiagonal matrix by n - 2 Householder reflections (LAPACK zhetd2, lower form, one thread per row of the
// trailing block: p = tau B22 v, w = p - (tau / 2)(p^H v) v, B22 <- B22 - v w^H - w v^H; only the diagonal and the SQUARES of the
// subdiagonal are kept, no reflector is stored), and lambda_max is bracketed by multisection on the Sturm count: 256 shifts a
// round, one per thread, seven rounds from the Gershgorin bracket to the last bits.  B (full storage, column-major) lives in
// LDS up to 96 channels, beyond in the workgroup's global scratch (the block La is free once M is whitened).
#define CBH_LDS_N 96
#define CBH_SMALL 64           // pairs of groups of at most this many channels are processed in LDS entirely
#define CBH_LD 65              // their row length (odd: a column walk touches every bank)
// in-place lower Cholesky of the n x n Hermitian matrix L (row-major, row length ld, lower triangle valid), all 256 threads
__device__ inline void cbh_cholesky(cd* L, int ld, int n, int* bad) {
    const int tid = threadIdx.x;
    for (int k = 0; k < n; ++k) {
        if (tid == 0) {
            const double d = L[k * ld + k].x;
            if (!(d > 0.0)) *bad = 1;
            L[k * ld + k] = make_double2(sqrt(d > 0.0 ? d : 1.0), 0.0);
        }
        __syncthreads();
        const double dk = L[k * ld + k].x;
        for (int i = k + 1 + tid; i < n; i += 256) {
            const cd v = L[i * ld + k];
            L[i * ld + k] = make_double2(v.x / dk, v.y / dk);
        }
        __syncthreads();
        const int m = n - k - 1;
        for (int e = tid; e < m * m; e += 256) {           // trailing lower triangle: L[i][j] -= L[i][k] conj(L[j][k]), k < j <= i
            const int i = k + 1 + e / m, j = k + 1 + e % m;
            if (j <= i) {
                const cd t = zmulc(L[i * ld + k], L[j * ld + k]);
                cd v = L[i * ld + j];
                v.x -= t.x; v.y -= t.y;
                L[i * ld + j] = v;
            }
        }
        __syncthreads();
    }
}
__device__ __forceinline__ double cbh_sum(double v, double* red4, int tid) {      // block sum over 256 threads, two barriers
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((tid & 63) == 0) red4[tid >> 6] = v;
    __syncthreads();
    const double r = red4[0] + red4[1] + red4[2] + red4[3];
    __syncthreads();
    return r;
}
__global__ void __launch_bounds__(256) canonical_big_hh_kernel(CanonArgs a, cd* scratch, int64_t n_items, int small_n, int small_ld) {
    extern __shared__ __align__(16) unsigned char cb_smem[];
    cd* Bl = reinterpret_cast<cd*>(cb_smem);                 // CBH_LDS_N^2 elements: B (column-major) or, for pairs of groups of at
    __shared__ double dg[CBIG_C], e2[CBIG_C];                // most CBH_SMALL channels, the whole problem (one factor block + M)
    __shared__ cd vs[CBIG_C], ws[CBIG_C];
    __shared__ double red4[4], sh[4];
    __shared__ int bad, first_above;
    const int tid = threadIdx.x;
    // (scratch == nullptr: every pair of this launch fits LDS -- small_n is the largest group -- and the three pointers are never used)
    cd* Lag = scratch ? scratch + (size_t)blockIdx.x * 3 * CBIG_C * CBIG_C : nullptr;
    cd* Lbg = scratch ? Lag + CBIG_C * CBIG_C : nullptr;
    cd* Mg = scratch ? Lbg + CBIG_C * CBIG_C : nullptr;
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int64_t bin = item / a.n_gpairs;
        int gp = (int)(item - bin * a.n_gpairs);
        int ga = 0, len = a.G - 1;
        while (gp >= len) { gp -= len; ++ga; --len; }
        int gb = ga + 1 + gp;
        if (a.sizes[gb] < a.sizes[ga]) { const int t = ga; ga = gb; gb = t; }      // B on the smaller group's side
        const int na = a.sizes[ga], nb = a.sizes[gb];
        const int32_t* ma = a.members + ga * a.mstride;
        const int32_t* mb = a.members + gb * a.mstride;
        const ScRec rec = a.accum + bin * a.floats_per_bin;
        // Pairs of groups of at most CBH_SMALL channels stay in LDS from the records to lambda_max: one factor block at a time (L_b takes
        // L_a's place once M is multiplied by L_a^-1, B takes L_b's) beside M; larger pairs keep the three blocks in the global scratch.
        // (small_n: CBH_SMALL, or the largest group when no group is larger -- the launch then asks for 2 small_n small_ld elements of LDS
        //  only and several workgroups share a compute unit: 34 KB at 32 channels)
        const bool small = nb <= small_n;                    // (na <= nb)
        cd* PA = small ? Bl : Lag;
        cd* PB = small ? Bl : Lbg;
        cd* PM = small ? Bl + small_n * small_ld : Mg;
        const int ld = small ? small_ld : CBIG_C;
        if (tid == 0) bad = 0;
        for (int e = tid; e < na * na; e += 256) { const int i = e / na, j = e % na; if (j <= i) PA[i * ld + j] = csm_read(rec, a, ma[i], ma[j]); }
        for (int e = tid; e < na * nb; e += 256) { const int i = e / nb, j = e % nb; PM[i * ld + j] = csm_read(rec, a, ma[i], mb[j]); }
        __syncthreads();
        cbh_cholesky(PA, ld, na, &bad);
        // M <- L_a^-1 M, right-looking: row k is final once it is divided by L_a[k][k]; every row below loses its multiple of it
        for (int k = 0; k < na; ++k) {
            const double dk = PA[k * ld + k].x;
            if (tid < nb) { const cd v = PM[k * ld + tid]; PM[k * ld + tid] = make_double2(v.x / dk, v.y / dk); }
            __syncthreads();
            const int rows = na - k - 1;
            for (int e = tid; e < rows * nb; e += 256) {
                const int i = k + 1 + e / nb, j = e % nb;
                const cd t = zmul(PA[i * ld + k], PM[k * ld + j]);
                cd v = PM[i * ld + j];
                v.x -= t.x; v.y -= t.y;
                PM[i * ld + j] = v;
            }
            __syncthreads();
        }
        for (int e = tid; e < nb * nb; e += 256) { const int i = e / nb, j = e % nb; if (j <= i) PB[i * ld + j] = csm_read(rec, a, mb[i], mb[j]); }
        __syncthreads();
        cbh_cholesky(PB, ld, nb, &bad);
        // M <- M L_b^-H: column k is final once it is divided by L_b[k][k]; every column to its right loses conj(L_b[j][k]) times it
        for (int k = 0; k < nb; ++k) {
            const double dk = PB[k * ld + k].x;
            if (tid < na) { const cd v = PM[tid * ld + k]; PM[tid * ld + k] = make_double2(v.x / dk, v.y / dk); }
            __syncthreads();
            const int cols = nb - k - 1;
            for (int e = tid; e < na * cols; e += 256) {
                const int i = e / cols, j = k + 1 + e % cols;
                const cd t = zmulc(PM[i * ld + k], PB[j * ld + k]);
                cd v = PM[i * ld + j];
                v.x -= t.x; v.y -= t.y;
                PM[i * ld + j] = v;
            }
            __syncthreads();
        }
        // B = M M^H, full storage, column-major with leading dimension na (small pairs: where the factor blocks were)
        cd* B = (small || na <= CBH_LDS_N) ? Bl : Lag;
        for (int e = tid; e < na * na; e += 256) {
            const int i = e % na, j = e / na;
            cd sacc = make_double2(0.0, 0.0);
            for (int k = 0; k < nb; ++k) { const cd t = zmulc(PM[i * ld + k], PM[j * ld + k]); sacc.x += t.x; sacc.y += t.y; }
            if (i == j) sacc.y = 0.0;
            B[(size_t)j * na + i] = sacc;
        }
        __syncthreads();
        // ---- tridiagonalisation (zhetd2, lower): d, e^2 ----
        for (int k = 0; k + 1 < na; ++k) {
            const int m = na - k - 1;                        // order of the trailing block
            const cd* col = B + (size_t)k * na + (k + 1);
            cd xi = make_double2(0.0, 0.0);
            if (tid < m) xi = col[tid];
            if (tid == 0) { sh[0] = xi.x; sh[1] = xi.y; dg[k] = B[(size_t)k * na + k].x; }
            const double xn2 = cbh_sum((tid >= 1 && tid < m) ? xi.x * xi.x + xi.y * xi.y : 0.0, red4, tid);
            const double alr = sh[0], ali = sh[1];
            if (xn2 == 0.0 && ali == 0.0) {                  // H = I (uniform over the workgroup)
                if (tid == 0) e2[k] = alr * alr;
                __syncthreads();
                continue;
            }
            const double nrm = sqrt(alr * alr + ali * ali + xn2);
            const double beta = alr >= 0.0 ? -nrm : nrm;
            const cd tk = make_double2((beta - alr) / beta, -ali / beta);
            cd vi = make_double2(0.0, 0.0);
            {
                const double dr = alr - beta, di = ali, dd = dr * dr + di * di;
                const cd scale = make_double2(dr / dd, -di / dd);          // 1 / (alpha - beta)
                if (tid < m) { vi = tid == 0 ? make_double2(1.0, 0.0) : zmul(xi, scale); vs[tid] = vi; }
                if (tid == 0) e2[k] = beta * beta;
            }
            __syncthreads();
            const cd* B22 = B + (size_t)(k + 1) * na + (k + 1);
            cd pi = make_double2(0.0, 0.0);
            if (tid < m) {
                cd acc = make_double2(0.0, 0.0);
                for (int j = 0; j < m; ++j) {
                    const cd aij = B22[(size_t)j * na + tid], vj = vs[j];
                    acc.x += aij.x * vj.x - aij.y * vj.y;
                    acc.y += aij.x * vj.y + aij.y * vj.x;
                }
                pi = zmul(tk, acc);
            }
            const double dre = cbh_sum(pi.x * vi.x + pi.y * vi.y, red4, tid);       // p^H v
            const double dim = cbh_sum(pi.x * vi.y - pi.y * vi.x, red4, tid);
            const cd al2 = zmul(make_double2(-0.5 * tk.x, -0.5 * tk.y), make_double2(dre, dim));
            cd wi = make_double2(0.0, 0.0);
            if (tid < m) {
                const cd t = zmul(al2, vi);
                wi = make_double2(pi.x + t.x, pi.y + t.y);
                ws[tid] = wi;
            }
            __syncthreads();
            if (tid < m) {
                cd* row = B + (size_t)(k + 1) * na + (k + 1) + tid;
                for (int j = 0; j < m; ++j) {
                    const cd wj = ws[j], vj = vs[j];
                    cd aij = row[(size_t)j * na];
                    aij.x -= vi.x * wj.x + vi.y * wj.y + wi.x * vj.x + wi.y * vj.y;       // a_ij -= v_i conj(w_j) + w_i conj(v_j)
                    aij.y -= vi.y * wj.x - vi.x * wj.y + wi.y * vj.x - wi.x * vj.y;
                    row[(size_t)j * na] = aij;
                }
            }
            __syncthreads();
        }
        if (tid == 0) { dg[na - 1] = B[(size_t)(na - 1) * na + (na - 1)].x; e2[na - 1] = 0.0; }
        __syncthreads();
        // ---- lambda_max by multisection on the Sturm count (x > lambda_max iff all n leading minors of T - x are negative) ----
        double lo, hi;
        {
            double g_hi = -1e300, g_lo = 1e300, emax = 0.0;
            for (int i = 0; i < na; ++i) {                   // (every thread: n <= 128 LDS reads)
                const double el = i > 0 ? sqrt(e2[i - 1]) : 0.0, er = i + 1 < na ? sqrt(e2[i]) : 0.0;
                g_hi = fmax(g_hi, dg[i] + el + er); g_lo = fmin(g_lo, dg[i] - el - er);
                emax = fmax(emax, e2[i]);
            }
            const double tn = fmax(fabs(g_hi), fabs(g_lo));
            const double pivmin = 2.2250738585072014e-308 * fmax(1.0, emax);
            const double pad = 2.0 * tn * 2.220446049250313e-16 * na + 2.0 * pivmin;
            lo = g_lo - pad; hi = g_hi + pad;
            for (int round = 0; round < 8; ++round) {
                const double x = lo + (hi - lo) * ((double)(tid + 1) / 257.0);
                int cnt = 0;
                double q = dg[0] - x;
                if (fabs(q) < pivmin) q = -pivmin;
                cnt += q < 0.0;
                for (int i = 1; i < na; ++i) {
                    q = dg[i] - x - e2[i - 1] / q;
                    if (fabs(q) < pivmin) q = -pivmin;
                    cnt += q < 0.0;
                }
                if (tid == 0) first_above = 256;
                __syncthreads();
                if (cnt == na) atomicMin(&first_above, tid);     // every eigenvalue lies below x
                __syncthreads();
                const int f = first_above;
                const double nlo = f == 0 ? lo : lo + (hi - lo) * ((double)f / 257.0);
                const double nhi = f == 256 ? hi : lo + (hi - lo) * ((double)(f + 1) / 257.0);
                __syncthreads();
                if (!(nhi - nlo < hi - lo)) break;           // (uniform: the bracket is down to neighbouring doubles)
                lo = nlo; hi = nhi;
            }
        }
        if (tid == 0) {
            double v = 0.5 * (lo + hi);
            if (bad) { v = nan(""); atomicAdd(a.fail, 1); }
            double* o = a.out + bin * a.G * a.G;
            o[ga * a.G + gb] = v;
            o[gb * a.G + ga] = v;
        }
        __syncthreads();
    }
}

extern "C" int sc_canonical_max_group(void) { return CBIG_C; }

extern "C" int sc_canonical_coherence_f64(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                                          int64_t n_observations, const int32_t* d_members, const int32_t* d_sizes,
                                          int n_groups, int max_group_size, double* d_out, int32_t* d_fail,
                                          void* stream) {
    ScTimed timed_("canonical_coherence", stream);
    SC_REQUIRE(d_accum && d_members && d_sizes && d_out && d_fail, "NULL argument");
    SC_REQUIRE(planes & SC_PLANE_CSM, "accumulator record must contain SC_PLANE_CSM");
    SC_REQUIRE(n_groups >= 1 && n_bins >= 1, "empty problem");
    if (max_group_size > CBIG_C) {
        sc_set_error("canonical coherence supports groups of at most %d channels (got %d)", CBIG_C, max_group_size);
        return SC_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    CanonArgs a;
    a.accum = sc_rec(d_accum, planes); a.members = d_members; a.sizes = d_sizes; a.out = d_out; a.fail = d_fail;
    a.n_bins = n_bins; a.G = n_groups; a.n_gpairs = n_groups * (n_groups - 1) / 2;
    a.NB = sc_n_blocks(n_signals); a.n_tiles = sc_n_tiles(a.NB);
    a.floats_per_bin = (int64_t)sc_plane_count(planes) * a.n_tiles * SC_TILE_ELEMS;
    a.p_csm = sc_plane_offset(planes, SC_PLANE_CSM);
    a.n_obs = (double)n_observations;
    a.mstride = max_group_size <= 32 ? 32 : CBIG_C;      // (the member table's row length is the caller's: sc_hip.h)
    a.jtol = 1e-24;          // off^2 <= jtol dia^2: eigenvalues to ~1e-12 relative (quadratic convergence)
    const int64_t total_out = n_bins * n_groups * n_groups;
    hipLaunchKernelGGL(canon_fill_nan, dim3((unsigned)((total_out + 255) / 256)), dim3(256), 0, st, d_out, total_out);
    (void)hipMemsetAsync(d_fail, 0, 4, st);
    const int64_t threads = n_bins * a.n_gpairs;
    if (threads > 0) {
        const unsigned blocks = (unsigned)((threads + 63) / 64);
        // members stride must match the instantiated CMAX
        if (max_group_size <= 16) {
            // group factors once per bin into a stream-ordered workspace, then a wave per (bin, group pair)
            cd* Lg = nullptr;
            const size_t lg_bytes = (size_t)n_bins * n_groups * CB_C * CB_C * sizeof(cd);
            const size_t ok_bytes = (((size_t)n_bins * n_groups * sizeof(int)) + 255) & ~(size_t)255;
            if (hipMallocAsync((void**)&Lg, lg_bytes + ok_bytes, st) != hipSuccess) {
                sc_set_error("canonical coherence: workspace allocation of %zu bytes failed", lg_bytes + ok_bytes);
                return SC_ENOMEM;
            }
            int* okb = reinterpret_cast<int*>(reinterpret_cast<char*>(Lg) + lg_bytes);
            const size_t lds_f = (size_t)CB_WAVES * CB_C * CB_C * sizeof(cd);
            const size_t lds_p = (size_t)CB_WAVES * 2 * CB_C * CB_C * sizeof(cd) + CB_WAVES * 24 * 8;
            (void)hipFuncSetAttribute((const void*)canonical_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p);
            hipLaunchKernelGGL(canonical_factor_kernel, dim3((unsigned)n_bins), dim3(64 * CB_WAVES), lds_f, st, a, Lg, okb);
            const int64_t n_gp = a.n_gpairs;
            const int64_t n_wg = ((n_gp + CB_WAVES - 1) / CB_WAVES) * n_bins;
            if (n_wg > 0x7fffffffLL) { (void)hipFreeAsync(Lg, st); sc_set_error("canonical coherence: too many (bin, pair) tasks"); return SC_EINVAL; }
            // (SC_CANON_EIG=jacobi: the parallel Jacobi of rounds 1-4, all sixteen eigenvalues -- A/B and cross-check)
            const char* eig = sc_switch(SC_SW_CANON_EIG);
            if (eig && eig[0] == 'j') {
                hipLaunchKernelGGL(canonical_pair_kernel, dim3((unsigned)n_wg), dim3(64 * CB_WAVES), lds_p, st, a, (const cd*)Lg,
                                   (const int*)okb);
            } else {
                const size_t lds_h = (size_t)CB_WAVES * (2 * CB_C * CB_C + 32) * sizeof(cd);
                (void)hipFuncSetAttribute((const void*)canonical_pair_hh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
                hipLaunchKernelGGL(canonical_pair_hh_kernel, dim3((unsigned)n_wg), dim3(64 * CB_WAVES), lds_h, st, a, (const cd*)Lg,
                                   (const int*)okb);
            }
            (void)hipFreeAsync(Lg, st);
        } else if (max_group_size <= 32 && sc_switch(SC_SW_CANON_EIG) && sc_switch(SC_SW_CANON_EIG)[0] == 'j') {
            // (rounds 1-5: a wave per problem, its three blocks in the lanes' scratch, Jacobi -- 134 ms for 8 groups of 32 x 513 bins
            //  where the workgroup-per-problem kernel below takes 22: kept for A/B and cross-check only)
            hipLaunchKernelGGL(canonical_kernel<32>, dim3(blocks), dim3(64), 0, st, a);
        } else if (max_group_size <= CBH_SMALL && !(sc_switch(SC_SW_CANON_EIG) && sc_switch(SC_SW_CANON_EIG)[0] == 'j')) {
            // every pair fits LDS: no scratch, and as many persistent workgroups per compute unit as their LDS allows
            const int sn = max_group_size, sld = max_group_size | 1;
            const size_t lds = (size_t)2 * sn * sld * sizeof(cd);
            int per_cu = (int)((160 * 1024 - 8 * 1024) / (lds + 7 * 1024));          // (static arrays: ~7 KB a workgroup)
            per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
            const int64_t want = (int64_t)256 * per_cu;
            const int slots = (int)(threads < want ? threads : want);
            SC_CHECK_HIP(hipFuncSetAttribute((const void*)canonical_big_hh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(canonical_big_hh_kernel, dim3((unsigned)slots), dim3(256), lds, st, a, (cd*)nullptr, threads, sn, sld);
        } else {
            // (members stride CBIG_C) persistent workgroups, three C x C matrices each in a stream-ordered scratch
            const int slots = (int)(threads < 256 ? threads : 256);
            cd* scratch = nullptr;
            const size_t sbytes = (size_t)slots * 3 * CBIG_C * CBIG_C * sizeof(cd);
            if (hipMallocAsync((void**)&scratch, sbytes, st) != hipSuccess) {
                sc_set_error("canonical coherence: scratch allocation of %zu bytes failed", sbytes);
                return SC_ENOMEM;
            }
            const char* eig = sc_switch(SC_SW_CANON_EIG);       // (=jacobi: the round-2 kernel, every eigenvalue of B -- A/B and cross-check)
            if (eig && eig[0] == 'j') {
                constexpr size_t HM = CBIG_C / 2;
                const size_t lds = (size_t)CBIG_C * (CBIG_C + 1) / 2 * sizeof(cd) + HM * 8 + HM * 16 + 2 * HM * 4 +
                                   HM * (HM + 1) * 2 + 64;
                SC_CHECK_HIP(hipFuncSetAttribute((const void*)canonical_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(canonical_big_kernel, dim3((unsigned)slots), dim3(256), lds, st, a, scratch, threads);
            } else {
                const size_t lds = (size_t)CBH_LDS_N * CBH_LDS_N * sizeof(cd);
                SC_CHECK_HIP(hipFuncSetAttribute((const void*)canonical_big_hh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(canonical_big_hh_kernel, dim3((unsigned)slots), dim3(256), lds, st, a, scratch, threads, CBH_SMALL, CBH_LD);
            }
            (void)hipFreeAsync(scratch, st);
        }
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
