// sc_jacobi.h -- parallel cyclic Jacobi on the packed upper triangle of a Hermitian matrix, one 256-thread workgroup
// (shared by global coherence, sc_global.hip, and the large-group canonical coherence, sc_canonical.hip).
// One thread per (pair u, pair v) 2 x 2 block of the round's pairing transforms its four entries in place
// (B' = J_u^H B J_v: the disjoint rotations of a round act on disjoint blocks); the rotation angles of every round can be
// logged so that eigenvectors are recovered as J_1 J_2 ... J_m e_k without ever forming the eigenvector matrix.
#pragma once
#include "sc_common.h"

typedef double2 jcd;
__device__ inline jcd g_mul(jcd a, jcd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
#define cd jcd
__device__ inline int gc_tri(int i, int j, int C) { return i * C - i * (i - 1) / 2 + (j - i); }   // i <= j
__device__ inline cd gc_get(const cd* A, int i, int j, int C) {
    if (i <= j) return A[gc_tri(i, j, C)];
    const cd v = A[gc_tri(j, i, C)];
    return make_double2(v.x, -v.y);
}
__device__ inline void gc_set(cd* A, int i, int j, int C, cd v) {
    if (i <= j) A[gc_tri(i, j, C)] = v;
    else A[gc_tri(j, i, C)] = make_double2(v.x, -v.y);
}

// block table of the round-robin pairing: block -> (pair u, pair v >= u); H = ceil(C / 2) pairs
template <int NT = 256>
__device__ inline void gc_block_table(unsigned short* blk_u, unsigned short* blk_v, int H, int tid) {
    for (int u = tid; u < H; u += NT) {
        const int start = u * H - u * (u - 1) / 2;
        for (int v = u; v < H; ++v) { blk_u[start + v - u] = (unsigned short)u; blk_v[start + v - u] = (unsigned short)v; }
    }
}

// Sweeps until the off-diagonal mass is below 1e-14 of the diagonal's (or max_sweeps).  A: packed triangle (LDS or global);
// rc [H], rs [H], rp [2 H]: rotation scratch; red [2][256], done, n_rounds: workgroup-shared; mylog: rotation log of this
// workgroup ([max_sweeps (M - 1)][H][3]) or NULL.  All NT threads must call (NT >= H); the diagonal of A holds the
// eigenvalues after.  NT = 1024 for matrices beyond 128 x 128: 8256 blocks per round at 256 signals are 32 dependent
// global round trips per thread with 256 threads, 8 with 1024.
template <int NT = 256>
__device__ inline void gc_jacobi(cd* A, int C, double* rc, cd* rs, int* rp, const unsigned short* blk_u,
                                 const unsigned short* blk_v, double (*red)[NT], int* done, int* n_rounds, double* mylog,
                                 int max_sweeps) {
    const int tid = threadIdx.x, M = C + (C & 1), H = M / 2;
    if (tid == 0) *n_rounds = 0;
    __syncthreads();
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        double off = 0.0, dia = 0.0;              // off = the whole packed triangle, dia = its diagonal
        for (int i = tid; i < C; i += NT) { const cd v = A[gc_tri(i, i, C)]; dia += v.x * v.x; }
        for (int e = tid; e < C * (C + 1) / 2; e += NT) off += A[e].x * A[e].x + A[e].y * A[e].y;
        red[0][tid] = off; red[1][tid] = dia;
        __syncthreads();
        for (int s = NT / 2; s > 0; s >>= 1) {
            if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
            __syncthreads();
        }
        if (tid == 0) { const double o = red[0][0] - red[1][0]; *done = (o <= 1e-28 * red[1][0] || o <= 0.0) ? 1 : 0; }
        __syncthreads();
        if (*done) break;
        for (int r = 0; r < M - 1; ++r) {
            if (tid < H) {
                int x, y;
                if (tid == 0) { x = M - 1; y = r; }
                else { x = (r + tid) % (M - 1); y = (r - tid + (M - 1)) % (M - 1); }
                int pi = x < y ? x : y, qi = x < y ? y : x;
                double c = 1.0;
                cd se = make_double2(0.0, 0.0);
                if (qi < C) {
                    const cd bb = A[gc_tri(pi, qi, C)];
                    const double ab = hypot(bb.x, bb.y);
                    if (ab >= 1e-300) {
                        const double tau = (A[gc_tri(qi, qi, C)].x - A[gc_tri(pi, pi, C)].x) / (2.0 * ab);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + hypot(1.0, tau));
                        c = 1.0 / hypot(1.0, t);
                        const double sn = t * c;
                        se = make_double2(sn * bb.x / ab, sn * bb.y / ab);
                    }
                } else {
                    qi = -1;                      // the dummy player sits out (identity rotation)
                }
                rc[tid] = c; rs[tid] = se; rp[2 * tid] = pi; rp[2 * tid + 1] = qi;
                if (mylog) {
                    double* lg = mylog + ((size_t)(sweep * (M - 1) + r) * H + tid) * 3;
                    lg[0] = c; lg[1] = se.x; lg[2] = se.y;
                }
            }
            __syncthreads();
            // 2x2 blocks (pair u <= pair v): B' = J_u^H B J_v with J = [[c, s], [-conj s, c]]
            for (int blk = tid; blk < H * (H + 1) / 2; blk += NT) {
                const int u = blk_u[blk], v = blk_v[blk];
                const int up = rp[2 * u], uq = rp[2 * u + 1], vp = rp[2 * v], vq = rp[2 * v + 1];
                const double cu = rc[u], cv = rc[v];
                const cd su = rs[u], sv = rs[v];
                const bool uhas = uq >= 0, vhas = vq >= 0;
                // B = [[a(up,vp), a(up,vq)], [a(uq,vp), a(uq,vq)]]
                cd b00 = gc_get(A, up, vp, C);
                cd b01 = vhas ? gc_get(A, up, vq, C) : make_double2(0, 0);
                cd b10 = uhas ? gc_get(A, uq, vp, C) : make_double2(0, 0);
                cd b11 = (uhas && vhas) ? gc_get(A, uq, vq, C) : make_double2(0, 0);
                // T = B J_v : columns vp, vq
                const cd svc = make_double2(sv.x, -sv.y);
                cd t00 = make_double2(cv * b00.x - g_mul(svc, b01).x, cv * b00.y - g_mul(svc, b01).y);
                cd t01 = make_double2(g_mul(sv, b00).x + cv * b01.x, g_mul(sv, b00).y + cv * b01.y);
                cd t10 = make_double2(cv * b10.x - g_mul(svc, b11).x, cv * b10.y - g_mul(svc, b11).y);
                cd t11 = make_double2(g_mul(sv, b10).x + cv * b11.x, g_mul(sv, b10).y + cv * b11.y);
                // B' = J_u^H T : rows up, uq  (row p' = c row p - s row q ; row q' = conj(s) row p + c row q)
                const cd suc = make_double2(su.x, -su.y);
                const cd n00 = make_double2(cu * t00.x - g_mul(su, t10).x, cu * t00.y - g_mul(su, t10).y);
                const cd n01 = make_double2(cu * t01.x - g_mul(su, t11).x, cu * t01.y - g_mul(su, t11).y);
                const cd n10 = make_double2(g_mul(suc, t00).x + cu * t10.x, g_mul(suc, t00).y + cu * t10.y);
                const cd n11 = make_double2(g_mul(suc, t01).x + cu * t11.x, g_mul(suc, t01).y + cu * t11.y);
                if (u == v) {
                    // diagonal block: Hermitian, off-diagonal annihilated by construction
                    gc_set(A, up, up, C, make_double2(n00.x, 0.0));
                    if (uhas) {
                        gc_set(A, uq, uq, C, make_double2(n11.x, 0.0));
                        gc_set(A, up, uq, C, make_double2(0.0, 0.0));
                    }
                } else {
                    gc_set(A, up, vp, C, n00);
                    if (vhas) gc_set(A, up, vq, C, n01);
                    if (uhas) gc_set(A, uq, vp, C, n10);
                    if (uhas && vhas) gc_set(A, uq, vq, C, n11);
                }
            }
            __syncthreads();
        }
        if (tid == 0) *n_rounds = (sweep + 1) * (M - 1);
        __syncthreads();
    }
}
#undef cd
