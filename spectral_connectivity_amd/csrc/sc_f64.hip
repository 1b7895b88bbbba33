// sc_f64.hip -- stage B of the float64 engine: the reference's own arithmetic on the device.
//
// The reference is float64 / complex128 end to end (transforms.py:1402-1405, connectivity.py:277-285, :1799-1822).
// The headline path computes in float32 (north_star's tolerance: 1e-5); measured at the full BASELINE depths that
// gives power to ~1e-6 relative but cancellation-small cross-spectra (coherency of nearly independent channels, the
// Im S numerator of wPLI) only to ~1e-7 of the array scale, i.e. 3e-5 ... 1e-4 relative on entries 1000x below the
// maximum (tests/test_gpu_full_depth.py).  `Connectivity(dtype=complex128)` -- the reference's default -- therefore runs
// THIS engine: float64 transform (sc_taper_windows_f64 + rocFFT double), complex128 spectra, the cross-spectral matrix
// on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), the per-observation non-linear planes on the fp64 VALU, double
// accumulator records (SC_RECORD_F64) and sc_measure_f64.  Elementwise relative error ~1e-13.
//
//   csm_f64_kernel        S += X^H X per (group, bin): Re S_ij = sum ar_i ar_j + ai_i ai_j, Im S_ij = sum ai_i ar_j - ar_i ai_j,
//                         four real fp64 MFMAs per 16x16 channel tile and 4 observations (connectivity.py:447-492)
//   nonlinear_f64_kernel  sum |Im s|, (Im s)^2, sign(Im s), s/|s| per observation (connectivity.py:897-1159)
// Same record layout as the f32 engine (upper-triangular 16x16 tiles, un-normalised sums: trial shards add), double
// elements.  Workgroup = 4 waves = one bin x one tile group; observation rows staged HBM -> registers -> LDS in chunks
// of 16 (double buffered, one barrier per chunk); XCD-aware blockIdx -> (bin, tile group) like sc_csm.hip.
#include <cstdlib>
#include <mutex>
#include "sc_common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

#define F64_OC 16            // observation rows per staged chunk

struct F64Args {
    const double2* base;     // X (group / bin offsets are added in-kernel)
    ScAxes ax;
    int64_t obs_stride;      // > 0: offset(o) = o * obs_stride
    double* accum;
    int64_t elems_per_bin;
    int n_bins, F, C, CP, NB, n_tiles, n_groups_of_tiles;
    int NB32, n_blocks32;
    int plane;               // record plane the CSM kernel fills (re; im = plane + 1)
    uint32_t planes;
    int n_obs;
    // Observations of a bin split over n_split workgroups (matrix-core CSM kernel and the 64 x 64 block plane kernel): part k
    // sums its share of the staged chunks into its own record -- part 0 into the caller's, the others into a stream-
    // ordered scratch ws[k - 1][bin][...] -- and f64_combine_kernel adds them in a fixed order.  903 bins on 256 CUs x 2
    // resident workgroups ran 2 rounds for 1.76 rounds of work; five parts per bin run 9 rounds of a fifth each (1.8).
    int n_split;
    double* ws;
    int64_t ws_part;         // doubles per part: n_bins * elems_per_bin
};

__device__ __forceinline__ int64_t f64_obs_offset(const F64Args& p, int o) {
    return p.obs_stride > 0 ? (int64_t)o * p.obs_stride : sc_obs_offset(p.ax, o);
}

// E double2 elements per thread cover E rows x CP channels (CP <= 256); element e = tid + 256 i sits in row e / CP --
// walked incrementally from (row0, c0) = (tid / CP, tid % CP), computed once per kernel: no division per element
struct F64Walk { int row0, c0, dq, dr; };
__device__ __forceinline__ F64Walk f64_walk(const F64Args& p, int tid) {
    F64Walk w;
    w.row0 = tid / p.CP; w.c0 = tid - w.row0 * p.CP;
    w.dq = 256 / p.CP; w.dr = 256 - w.dq * p.CP;
    return w;
}
// UNIT: the coefficient is replaced by its unit phasor x / |x| (0 / 0 = NaN like the reference's x / abs(x)); the zero
// fill of rows past n_obs and of absent channels stays zero
template <int E, bool UNIT = false>
__device__ __forceinline__ void f64_load(const F64Args& p, const F64Walk& w, const double2* base, int o0, double2 (&r)[E]) {
    int row = w.row0, c = w.c0;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        double2 v = make_double2(0.0, 0.0);
        const int o = o0 + row;
        if (row < E && o < p.n_obs && c < p.C) {
            v = base[f64_obs_offset(p, o) + c];
            if constexpr (UNIT) {
                const double ia = 1.0 / sqrt(v.x * v.x + v.y * v.y);
                v = make_double2(v.x * ia, v.y * ia);
            }
        }
        r[i] = v;
        row += w.dq; c += w.dr;
        if (c >= p.CP) { c -= p.CP; ++row; }
    }
}
template <int E>
__device__ __forceinline__ void f64_store(const F64Args& p, double2* lds, int tid, const double2 (&r)[E]) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int e = tid + i * 256;
        if (e < E * p.CP) lds[e] = r[i];          // row stride = CP
    }
}

template <int MAX_SLOTS, int E, bool UNIT>
__global__ void __launch_bounds__(256, MAX_SLOTS <= 9 ? 2 : 1) csm_f64_kernel(F64Args p) {
    extern __shared__ __align__(16) unsigned char f64_smem[];
    double2* lds = reinterpret_cast<double2*>(f64_smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int tg = j % p.n_groups_of_tiles;
    const int unit = (j / p.n_groups_of_tiles) * 8 + xcd;             // (bin, part)
    if (unit >= p.n_bins * p.n_split) return;
    const int bin = unit / p.n_split, part = unit - bin * p.n_split;
    const int g = bin / p.F, f = bin - g * p.F;
    const double2* base = p.base + (int64_t)f * p.ax.sF + sc_group_offset(p.ax, g);

    int bi[MAX_SLOTS], bj[MAX_SLOTS];
    bool valid[MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s) {
        const int t = (tg * MAX_SLOTS + s) * 4 + wave;
        valid[s] = t < p.n_tiles;
        int r = 0, rem = valid[s] ? t : 0, len = p.NB;
        while (rem >= len) { rem -= len; ++r; --len; }
        bi[s] = r; bj[s] = r + rem;
    }
    f64x4 re[MAX_SLOTS], im[MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s) { re[s] = (f64x4){0.0, 0.0, 0.0, 0.0}; im[s] = re[s]; }

    const int buf = E * p.CP;
    const int all_chunks = (p.n_obs + E - 1) / E;
    const int ch0 = (int)((int64_t)part * all_chunks / p.n_split), ch1 = (int)((int64_t)(part + 1) * all_chunks / p.n_split);
    const int n_chunks = ch1 - ch0;                                   // this part's chunks: [ch0, ch1)
    const F64Walk walk = f64_walk(p, tid);
    double2 regs[E];
    f64_load<E, UNIT>(p, walk, base, ch0 * E, regs);
    f64_store<E>(p, lds, tid, regs);
    __syncthreads();
    // A operand: lane l holds A[i = l & 15][k = l >> 4]; B operand: B[k = l >> 4][j = l & 15]  (one f64 each)
    const int k = lane >> 4, c16 = lane & 15;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const double2* cur = lds + (ch & 1) * buf;
        double2* nxt = lds + ((ch + 1) & 1) * buf;
        const bool more = ch + 1 < n_chunks;
        if (more) f64_load<E, UNIT>(p, walk, base, (ch0 + ch + 1) * E, regs);
        // operands of slot s + 1 are requested before the four MFMAs of slot s are issued (left to itself the compiler
        // reloads one register pair per slot and waits for it: an LDS round trip between every two groups of MFMAs)
        constexpr int NSTEP = (E / 4) * MAX_SLOTS;
        double2 an, bn;
        {
            const double2* rowp = cur + k * p.CP + c16;
            an = rowp[16 * bi[0]];
            bn = rowp[16 * bj[0]];
        }
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {            // invalid slots recompute tile (0, 0) and are never stored
            const int s = st % MAX_SLOTS;
            const double2 a = an, b = bn;
            if (st + 1 < NSTEP) {
                const int kk1 = (st + 1) / MAX_SLOTS, s1 = (st + 1) % MAX_SLOTS;
                const double2* rowp = cur + (kk1 * 4 + k) * p.CP + c16;
                an = rowp[16 * bi[s1]];
                bn = rowp[16 * bj[s1]];
            }
            re[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, re[s], 0, 0, 0);
            im[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, im[s], 0, 0, 0);
            re[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, re[s], 0, 0, 0);
            im[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.x, b.y, im[s], 0, 0, 0);
        }
        if (more) f64_store<E>(p, nxt, tid, regs);
        __syncthreads();
    }
    // C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg  (NOT the f32 map)
    double* out = (part == 0 ? p.accum : p.ws + (int64_t)(part - 1) * p.ws_part) + (int64_t)bin * p.elems_per_bin +
                  (int64_t)p.plane * p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s) {
        if (!valid[s]) continue;
        const int t = (tg * MAX_SLOTS + s) * 4 + wave;
        double* o_re = out + (int64_t)t * SC_TILE_ELEMS;
        double* o_im = o_re + (int64_t)p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = ((lane >> 4) + 4 * r) * 16 + (lane & 15);
            o_re[idx] = re[s][r];
            o_im[idx] = im[s][r];
        }
    }
}

// ---- per-observation non-linear planes, fp64 VALU ---------------------------------------------------------------
// Workgroup = one bin and MAXB upper-triangular 32x32 channel blocks; a wave is an 8x8 lane grid, a lane owns a 4x4
// tile of pairs; the 4 waves take observation rows o = wave (mod 4) of every chunk and are summed through LDS.
template <uint32_t WHICH>
struct F64Planes {
    static constexpr int N = ((WHICH & SC_PLANE_ABS_IM) ? 1 : 0) + ((WHICH & SC_PLANE_IM_SQ) ? 1 : 0) +
                             ((WHICH & SC_PLANE_SIGN_IM) ? 1 : 0) + ((WHICH & SC_PLANE_UNIT) ? 2 : 0);
};

template <uint32_t WHICH, int MAXB, int E>
__global__ void __launch_bounds__(256) nonlinear_f64_kernel(F64Args p) {
    extern __shared__ __align__(16) unsigned char f64_smem[];
    double2* lds = reinterpret_cast<double2*>(f64_smem);
    constexpr int NP = F64Planes<WHICH>::N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int n_sets = (p.n_blocks32 + MAXB - 1) / MAXB;
    const int bs = jj % n_sets;
    const int bin = (jj / n_sets) * 8 + xcd;
    if (bin >= p.n_bins) return;
    const int g = bin / p.F, f = bin - g * p.F;
    const double2* base = p.base + (int64_t)f * p.ax.sF + sc_group_offset(p.ax, g);

    int BI[MAXB], BJ[MAXB];
    bool valid[MAXB];
#pragma unroll
    for (int s = 0; s < MAXB; ++s) {
        const int t = bs * MAXB + s;
        valid[s] = t < p.n_blocks32;
        int r = 0, rem = valid[s] ? t : 0, len = p.NB32;
        while (rem >= len) { rem -= len; ++r; --len; }
        BI[s] = r; BJ[s] = r + rem;
    }
    double acc[MAXB][NP][16];
#pragma unroll
    for (int s = 0; s < MAXB; ++s)
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][q][e] = 0.0;

    const int li = lane >> 3, lj = lane & 7;
    const int buf = F64_OC * p.CP;
    const int n_chunks = (p.n_obs + F64_OC - 1) / F64_OC;
    static_assert(E == F64_OC, "one staged row per register");
    const F64Walk walk = f64_walk(p, tid);
    double2 regs[E];
    f64_load<E>(p, walk, base, 0, regs);
    f64_store<E>(p, lds, tid, regs);
    __syncthreads();
    for (int ch = 0; ch < n_chunks; ++ch) {
        const double2* cur = lds + (ch & 1) * buf;
        double2* nxt = lds + ((ch + 1) & 1) * buf;
        const bool more = ch + 1 < n_chunks;
        if (more) f64_load<E>(p, walk, base, (ch + 1) * F64_OC, regs);
        // rows past n_obs are zero: +0 for every plane except UNIT (0 / 0 = NaN), so bound the loop by the real count
        const int rows = min(F64_OC, p.n_obs - ch * F64_OC);
        for (int row = wave; row < rows; row += 4) {
            const double2* rp = cur + row * p.CP;
#pragma unroll
            for (int s = 0; s < MAXB; ++s) {
                double2 xi[4], xj[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) { xi[a] = rp[BI[s] * 32 + li * 4 + a]; xj[a] = rp[BJ[s] * 32 + lj * 4 + a]; }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const double imv = xi[a].y * xj[b].x - xi[a].x * xj[b].y;
                        int q = 0;
                        if constexpr (WHICH & SC_PLANE_ABS_IM) { acc[s][q][a * 4 + b] += fabs(imv); ++q; }
                        if constexpr (WHICH & SC_PLANE_IM_SQ) { acc[s][q][a * 4 + b] += imv * imv; ++q; }
                        if constexpr (WHICH & SC_PLANE_SIGN_IM) {
                            acc[s][q][a * 4 + b] += (imv > 0.0 ? 1.0 : 0.0) - (imv < 0.0 ? 1.0 : 0.0);
                            ++q;
                        }
                        if constexpr (WHICH & SC_PLANE_UNIT) {
                            const double rev = xi[a].x * xj[b].x + xi[a].y * xj[b].y;
                            const double mag = sqrt(rev * rev + imv * imv);       // 0 / 0 -> NaN like the reference
                            acc[s][q][a * 4 + b] += rev / mag;
                            acc[s][q + 1][a * 4 + b] += imv / mag;
                        }
                    }
            }
        }
        if (more) f64_store<E>(p, nxt, tid, regs);
        __syncthreads();
    }
    // cross-wave sum through LDS in a fixed order, one (block, plane) at a time: 4 waves x 16 x 64 doubles
    double* red = reinterpret_cast<double*>(f64_smem);
    double* out_bin = p.accum + (int64_t)bin * p.elems_per_bin;
#pragma unroll
    for (int s = 0; s < MAXB; ++s) {
        if (!valid[s]) continue;       // identical for all 4 waves
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[s][q][e];
            __syncthreads();
            int plane = -1;
            {
                int kq = 0;
                if constexpr (WHICH & SC_PLANE_ABS_IM) { if (q == kq) plane = sc_plane_offset(p.planes, SC_PLANE_ABS_IM); ++kq; }
                if constexpr (WHICH & SC_PLANE_IM_SQ) { if (q == kq) plane = sc_plane_offset(p.planes, SC_PLANE_IM_SQ); ++kq; }
                if constexpr (WHICH & SC_PLANE_SIGN_IM) { if (q == kq) plane = sc_plane_offset(p.planes, SC_PLANE_SIGN_IM); ++kq; }
                if constexpr (WHICH & SC_PLANE_UNIT) {
                    if (q == kq) plane = sc_plane_offset(p.planes, SC_PLANE_UNIT);
                    if (q == kq + 1) plane = sc_plane_offset(p.planes, SC_PLANE_UNIT) + 1;
                }
            }
            double* out = out_bin + (int64_t)plane * p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int e = wave * 4 + e4;
                const double v = red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane] +
                                 red[(2 * 16 + e) * 64 + lane] + red[(3 * 16 + e) * 64 + lane];
                const int i = BI[s] * 32 + li * 4 + (e >> 2), jx = BJ[s] * 32 + lj * 4 + (e & 3);
                const int ti = i >> 4, tj = jx >> 4;
                if (ti <= tj && tj < p.NB)
                    out[(int64_t)sc_tile_index(ti, tj, p.NB) * SC_TILE_ELEMS + (i & 15) * 16 + (jx & 15)] = v;
            }
            __syncthreads();
        }
    }
}

// ---- the same planes for 48+ channels: 64 x 64 channel blocks, 8 x 8 pairs per lane ---------------------------------
// nonlinear_f64_kernel reads (4 + 4) x 16 bytes of LDS per 16 pairs and observation -- 2.7 bytes per fp64 instruction
// against the 2 the LDS pipe delivers at the fp64 VALU rate -- keeps 264 registers (one wave per SIMD) and spends as
// many instructions on the integer divisions of its staging as on the products: 20 ms for the |Im s| plane of cfg3.  Here a
// workgroup owns ONE upper-triangular 64 x 64 block of a bin (blocks of a bin on the same XCD: they re-read the same
// rows), a wave is an 8 x 8 lane grid and a lane owns the 8 x 8 pairs (li + 8 a, lj + 8 b): 16 reads of 16 bytes per 64
// pairs (1.3 bytes per instruction), lanes of a read walk consecutive channels (no bank conflicts); the four waves take
// the observation rows of a 16-row chunk round-robin and are summed through LDS in a fixed order at the end.  Staging:
// thread t pulls channel t & 127 (64 of the block's row range, 64 of its column range) of rows t >> 7, + 2, ...: whole
// 2 KB runs, no division.
#define F64B_OC 16
template <uint32_t WHICH, bool DIAG>
__global__ void __launch_bounds__(256, 2) nonlinear_f64_block_kernel(F64Args p) {
    static_assert(WHICH == SC_PLANE_ABS_IM || WHICH == SC_PLANE_IM_SQ || WHICH == SC_PLANE_SIGN_IM, "one plane per launch");
    extern __shared__ __align__(16) unsigned char f64_smem[];
    double2* lds = reinterpret_cast<double2*>(f64_smem);            // [2][F64B_OC][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // diagonal blocks and off-diagonal blocks are separate launches (DIAG): their loop nests differ, and both in one
    // kernel cost 300 spilled registers
    const int NB64 = (p.C + 63) >> 6, n_blk = DIAG ? NB64 : NB64 * (NB64 - 1) / 2;
    const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
    const int blk = jj % n_blk;
    const int unit = (jj / n_blk) * 8 + xcd;                          // (bin, part): see F64Args::n_split
    if (unit >= p.n_bins * p.n_split) return;
    const int bin = unit / p.n_split, part = unit - bin * p.n_split;
    int BI = 0, BJ = 0;
    if constexpr (DIAG) { BI = BJ = blk; }
    else { int rem = blk, len = NB64 - 1; while (rem >= len) { rem -= len; ++BI; --len; } BJ = BI + 1 + rem; }
    const int g = bin / p.F, f = bin - g * p.F;
    const double2* base = p.base + (int64_t)f * p.ax.sF + sc_group_offset(p.ax, g);
    // staging: channel slot cs = tid & 127 (0-63: row range of the block, 64-127: its column range), rows tid >> 7 + 2 u
    const int cs = tid & 127, r0 = tid >> 7;
    const int cg = (cs < 64 ? BI * 64 + cs : BJ * 64 + (cs - 64));
    const bool have = cg < p.C && (!DIAG || cs < 64);             // a diagonal block reads its row range for both operands
    double2 regs[F64B_OC / 2];
    auto fetch = [&](int o0) {
#pragma unroll
        for (int u = 0; u < F64B_OC / 2; ++u) {
            const int o = o0 + r0 + 2 * u;
            regs[u] = (have && o < p.n_obs) ? base[f64_obs_offset(p, o) + cg] : make_double2(0.0, 0.0);
        }
    };
    auto park = [&](double2* dst) {
#pragma unroll
        for (int u = 0; u < F64B_OC / 2; ++u) dst[(r0 + 2 * u) * 128 + cs] = regs[u];
    };
    double acc[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) acc[e] = 0.0;
    const int li = lane >> 3, lj = lane & 7;
    constexpr int joff = DIAG ? 0 : 64;
    const int all_chunks = (p.n_obs + F64B_OC - 1) / F64B_OC;
    const int ch0 = (int)((int64_t)part * all_chunks / p.n_split), ch1 = (int)((int64_t)(part + 1) * all_chunks / p.n_split);
    const int n_chunks = ch1 - ch0;                                   // this part's chunks: [ch0, ch1)
    fetch(ch0 * F64B_OC);
    park(lds);
    __syncthreads();
    for (int ch = 0; ch < n_chunks; ++ch) {
        const double2* cur = lds + (ch & 1) * (F64B_OC * 128);
        const bool more = ch + 1 < n_chunks;
        if (more) fetch((ch0 + ch + 1) * F64B_OC);
        // rows past n_obs are zero: they add |0|, 0^2, sign(0) = 0
#pragma unroll 1
        for (int row = wave; row < F64B_OC; row += 4) {
            const double2* rp = cur + row * 128;
            // all sixteen operands of the row are requested before the first product (one LDS round trip per row, hidden by
            // the SIMD's other wave; loaded one column at a time the compiler waits for each of them).  (Tried in round 3:
            // the next row's operands requested under this row's products -- 128 accumulator + 2 x 64 operand + 32 staging
            // registers exceed the 256 of a lane: 230-560 spilled dwords, not kept.)
            double2 xi[8], xjv[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) xi[a] = rp[li + 8 * a];
#pragma unroll
            for (int b = 0; b < 8; ++b) xjv[b] = rp[joff + lj + 8 * b];
            // A diagonal block needs i <= j only: with i = li + 8 a, j = lj + 8 b every pair of a sub-tile a > b lies below
            // the diagonal -- 28 of a lane's 64 sub-tiles are skipped there (their mirror images are filled on the way out).
            if constexpr (DIAG) {
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const double2 xj = xjv[b];
#pragma unroll
                    for (int a = 0; a <= b; ++a) {
                        const double imv = xi[a].y * xj.x - xi[a].x * xj.y;
                        if constexpr (WHICH == SC_PLANE_ABS_IM) acc[a * 8 + b] += fabs(imv);
                        else if constexpr (WHICH == SC_PLANE_IM_SQ) acc[a * 8 + b] = fma(imv, imv, acc[a * 8 + b]);
                        else acc[a * 8 + b] += (imv > 0.0 ? 1.0 : 0.0) - (imv < 0.0 ? 1.0 : 0.0);
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const double2 xj = xjv[b];
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        const double imv = xi[a].y * xj.x - xi[a].x * xj.y;
                        if constexpr (WHICH == SC_PLANE_ABS_IM) acc[a * 8 + b] += fabs(imv);
                        else if constexpr (WHICH == SC_PLANE_IM_SQ) acc[a * 8 + b] = fma(imv, imv, acc[a * 8 + b]);
                        else acc[a * 8 + b] += (imv > 0.0 ? 1.0 : 0.0) - (imv < 0.0 ? 1.0 : 0.0);
                    }
                }
            }
        }
        if (more) park(lds + ((ch + 1) & 1) * (F64B_OC * 128));
        __syncthreads();
    }
    // waves 1-3 -> LDS, wave 0 adds them in a fixed order; two halves of 32 accumulators (3 x 32 x 64 doubles = 48 KB)
    double* red = reinterpret_cast<double*>(f64_smem);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (wave > 0) {
#pragma unroll
            for (int e = 0; e < 32; ++e) red[((wave - 1) * 32 + e) * 64 + lane] = acc[h * 32 + e];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < 32; ++e)
                acc[h * 32 + e] += red[(0 * 32 + e) * 64 + lane] + red[(1 * 32 + e) * 64 + lane] + red[(2 * 32 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    double* out = (part == 0 ? p.accum : p.ws + (int64_t)(part - 1) * p.ws_part) + (int64_t)bin * p.elems_per_bin +
                  (int64_t)sc_plane_offset(p.planes, WHICH) * p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (DIAG && a > b) continue;                  // not computed in a diagonal block (mirror image written below)
            const int i = BI * 64 + li + 8 * a, jx = BJ * 64 + lj + 8 * b;
            const int ti = i >> 4, tj = jx >> 4;
            if (ti <= tj && tj < p.NB) {
                double* tile = out + (int64_t)sc_tile_index(ti, tj, p.NB) * SC_TILE_ELEMS;
                tile[(i & 15) * 16 + (jx & 15)] = acc[a * 8 + b];
                // the lower half of a diagonal 16 x 16 tile (no consumer reads it -- they mirror the upper half -- but it
                // is not left uninitialised): |Im s| and (Im s)^2 are symmetric, sign(Im s) antisymmetric
                if (DIAG && a < b && ti == tj)
                    tile[(jx & 15) * 16 + (i & 15)] = (WHICH == SC_PLANE_SIGN_IM) ? -acc[a * 8 + b] : acc[a * 8 + b];
            }
        }
}

// ---------------------------------------------------------------------------------------------- host side
static int f64_setup(const void* d_X, const sc_spectra_desc* desc, uint32_t planes, double* d_accum, F64Args* a) {
    SC_REQUIRE(d_X && desc && d_accum, "NULL argument");
    SC_REQUIRE(((uintptr_t)d_X % 16) == 0, "complex128 spectra must be 16-byte aligned");
    ScAxes ax;
    sc_make_axes(desc, &ax);
    SC_REQUIRE(ax.C >= 1 && ax.F >= 1 && ax.n_obs >= 1 && ax.n_groups >= 1, "empty dimension");
    if (ax.C > SC_MAX_SIGNALS) {
        sc_set_error("n_signals=%d exceeds SC_MAX_SIGNALS=%d", ax.C, SC_MAX_SIGNALS);
        return SC_EUNSUPPORTED;
    }
    a->base = (const double2*)d_X;
    a->ax = ax;
    {   // are the reduced axes one linear run?  (same rule as sc_stage_linear_stride)
        int64_t s = 0, span = 1;
        bool ok = true;
        if (ax.rK > 1) { s = ax.sK; span = ax.rK; }
        if (ax.rR > 1) { if (span == 1) { s = ax.sR; span = ax.rR; } else { ok = ok && (ax.sR == s * span); span *= ax.rR; } }
        if (ax.rW > 1) { if (span == 1) { s = ax.sW; span = ax.rW; } else { ok = ok && (ax.sW == s * span); span *= ax.rW; } }
        a->obs_stride = span == 1 ? 1 : ((ok && s > 0) ? s : 0);
    }
    a->accum = d_accum;
    a->C = ax.C;
    a->NB = sc_n_blocks(ax.C);
    a->n_tiles = sc_n_tiles(a->NB);
    a->NB32 = (ax.C + 31) / 32;
    a->n_blocks32 = a->NB32 * (a->NB32 + 1) / 2;
    a->CP = a->NB32 * 32;
    a->n_bins = ax.n_groups * ax.F;
    a->F = ax.F;
    a->elems_per_bin = (int64_t)sc_plane_count(planes) * a->n_tiles * SC_TILE_ELEMS;
    a->planes = planes;
    a->n_obs = ax.n_obs;
    a->plane = 0;
    a->n_groups_of_tiles = 1;
    a->n_split = 1;
    a->ws = nullptr;
    a->ws_part = (int64_t)a->n_bins * a->elems_per_bin;
    return SC_OK;
}

// accum[bin][plane0 .. plane0 + n_planes) += ws[0][bin][...] + ws[1][bin][...] + ... (fixed order)
__global__ void __launch_bounds__(256) f64_combine_kernel(F64Args p, int plane0, int n_planes) {
    const int64_t plane = (int64_t)p.n_tiles * SC_TILE_ELEMS, per_bin = (int64_t)n_planes * plane / 2;     // double2 items
    const int64_t total = per_bin * p.n_bins;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t bin = i / per_bin, e = (i - bin * per_bin) * 2;
        const int64_t off = bin * p.elems_per_bin + (int64_t)plane0 * plane + e;
        double2 a = *reinterpret_cast<const double2*>(p.accum + off);
        for (int k = 0; k + 1 < p.n_split; ++k) {
            const double2 b = *reinterpret_cast<const double2*>(p.ws + (int64_t)k * p.ws_part + off);
            a.x += b.x; a.y += b.y;
        }
        *reinterpret_cast<double2*>(p.accum + off) = a;
    }
}
static void f64_combine(const F64Args& a, int plane0, int n_planes, hipStream_t st) {
    if (a.n_split > 1) hipLaunchKernelGGL(f64_combine_kernel, dim3(2048), dim3(256), 0, st, a, plane0, n_planes);
}

// parts per bin for W workgroups per part on `slots` resident workgroups: the S with the fewest rounds / S, at least 16
// staged chunks per part
static int f64_pick_split(int64_t W, int n_obs, int rows_per_chunk) {
    const char* e = sc_switch(SC_SW_F64_SPLIT);
    if (e && atoi(e) >= 1 && atoi(e) <= 16) return atoi(e);
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
    }
    const int64_t slots = (int64_t)n_cu * 2;
    const int nc = (n_obs + rows_per_chunk - 1) / rows_per_chunk;
    int best = 1;
    double best_cost = 1e30;
    for (int S = 1; S <= 12; ++S) {
        if (S > 1 && nc / S < 16) break;
        const double cost = (double)((W * S + slots - 1) / slots) / S * (1.0 + 0.01 * (S - 1));
        if (cost < best_cost - 1e-9) { best_cost = cost; best = S; }
    }
    return best;
}

template <int MAX_SLOTS, bool UNIT, int OC>
static int launch_csm_f64_oc(F64Args a, hipStream_t st) {
    a.n_groups_of_tiles = (a.n_tiles + 4 * MAX_SLOTS - 1) / (4 * MAX_SLOTS);
    const int64_t units = (int64_t)a.n_bins * a.n_split;
    const unsigned grid = (unsigned)(((units + 7) / 8) * 8 * a.n_groups_of_tiles);
    const size_t shmem = (size_t)2 * OC * a.CP * sizeof(double2);
    auto k = csm_f64_kernel<MAX_SLOTS, OC, UNIT>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), shmem, st, a);
    f64_combine(a, a.plane, 2, st);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
template <int MAX_SLOTS, bool UNIT = false>
static int launch_csm_f64(F64Args a, hipStream_t st) {
    // observation rows per staged chunk: 8 registers of staging, two workgroups per CU.  (SC_F64_OC=16, diagnostic: half
    // the barriers, but 256 registers with 24 spilled and one workgroup per CU -- 11.2 against 8.8 ms at cfg3.)
    const char* e = sc_switch(SC_SW_F64_OC);
    if (e && atoi(e) == 16) return launch_csm_f64_oc<MAX_SLOTS, UNIT, 16>(a, st);
    return launch_csm_f64_oc<MAX_SLOTS, UNIT, 8>(a, st);
}

template <uint32_t WHICH, int MAXB>
static int launch_nl_f64(const F64Args& a, hipStream_t st) {
    const int n_sets = (a.n_blocks32 + MAXB - 1) / MAXB;
    const unsigned grid = (unsigned)(((a.n_bins + 7) / 8) * 8 * n_sets);
    size_t shmem = (size_t)2 * F64_OC * a.CP * sizeof(double2);
    if (shmem < (size_t)4 * 16 * 64 * sizeof(double)) shmem = (size_t)4 * 16 * 64 * sizeof(double);
    auto k = nonlinear_f64_kernel<WHICH, MAXB, 16>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), shmem, st, a);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

template <uint32_t WHICH>
static int launch_nl_f64_block(const F64Args& a, hipStream_t st) {
    const int NB64 = (a.C + 63) / 64, n_off = NB64 * (NB64 - 1) / 2;
    const unsigned bins8 = (unsigned)((((int64_t)a.n_bins * a.n_split + 7) / 8) * 8);        // (bin, part) units
    const size_t shmem = (size_t)2 * F64B_OC * 128 * sizeof(double2);      // 64 KB (the final reduction needs 48 KB)
    auto kd = nonlinear_f64_block_kernel<WHICH, true>;
    auto ko = nonlinear_f64_block_kernel<WHICH, false>;
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)ko, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    if (n_off > 0) hipLaunchKernelGGL(ko, dim3(bins8 * n_off), dim3(256), shmem, st, a);
    hipLaunchKernelGGL(kd, dim3(bins8 * NB64), dim3(256), shmem, st, a);
    f64_combine(a, sc_plane_offset(a.planes, WHICH), 1, st);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

// Fills the planes named by `which` (a subset of `planes`) of the double records in d_accum.  SC_PLANE_CSM runs on the
// fp64 matrix cores, every other plane on the fp64 VALU.
extern "C" int sc_accumulate_f64(const void* d_X, const sc_spectra_desc* desc, uint32_t planes, uint32_t which,
                                 double* d_accum, void* stream) {
    ScTimed timed_("accumulate_f64", stream);
    planes &= ~SC_RECORD_F64;
    which &= ~SC_RECORD_F64;
    SC_REQUIRE(which != 0 && (which & planes) == which, "which must be a non-empty subset of planes");
    F64Args a;
    const int rc0 = f64_setup(d_X, desc, planes, d_accum, &a);
    if (rc0 != SC_OK) return rc0;
    hipStream_t st = (hipStream_t)stream;
    int rc = SC_OK;
    // With both the CSM and per-observation planes requested, the plane kernels go to a side stream forked from and
    // joined back into the caller's: they fill the CUs the CSM kernel's tail leaves idle (-1 ms of 19 at cfg3).  There is
    // no more to gain from overlapping them: forced to share every CU (one workgroup of each) the pair runs SLOWER
    // (22 ms) -- on MI355X the fp64 matrix rate equals the fp64 vector rate, the two kernels compete for the same
    // arithmetic, and their sum, 0.47 T lane-operations at cfg3 = 11.9 ms at 2.4 GHz, is the bound of this engine.
    // One side stream per DEVICE (created under a lock on the device that is current at the call), and a fresh event
    // pair per CALL: two host threads, or two devices, never share an event -- a shared pair would let one call's join
    // wait on the other's record.
    const bool fork = (which & SC_PLANE_CSM) && (which & ~SC_PLANE_CSM) && a.C >= 48 && !sc_switch(SC_SW_F64_NO_FORK);
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    struct EventPair {            // destroyed when the call returns (hipEventDestroy defers the release past pending work)
        hipEvent_t &a, &b;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev_guard{ev_fork, ev_join};
    if (fork) {
        static std::mutex mu;
        static hipStream_t side_of_device[64] = {nullptr};
        int dev = 0;
        SC_CHECK_HIP(hipGetDevice(&dev));
        SC_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!side_of_device[dev] && hipStreamCreateWithFlags(&side_of_device[dev], hipStreamNonBlocking) != hipSuccess) {
                sc_set_error("sc_accumulate_f64: side stream creation failed");
                return SC_EHIP;
            }
            side = side_of_device[dev];
        }
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) {
            sc_set_error("sc_accumulate_f64: event creation failed");
            return SC_EHIP;
        }
    }
    // observations of a bin split over several workgroups (F64Args::n_split): partial records in a stream-ordered scratch
    // of this call, allocated BEFORE the fork (both streams write disjoint planes of it) and released after the join
    const bool block_planes = a.C >= 48 && !sc_switch(SC_SW_F64_NO_BLOCK) && (which & (SC_PLANE_ABS_IM | SC_PLANE_IM_SQ | SC_PLANE_SIGN_IM));
    double* ws = nullptr;
    struct ScratchGuard {         // the scratch goes back to the pool on EVERY way out of this call, behind the work queued on `s`
        double*& p;
        hipStream_t s;
        ~ScratchGuard() { if (p) (void)hipFreeAsync(p, s); }
    } ws_guard{ws, (hipStream_t)stream};
    if ((which & (SC_PLANE_CSM | SC_PLANE_UNIT)) || block_planes) {
        int S = f64_pick_split((int64_t)a.n_bins * ((a.n_tiles + 35) / 36), a.n_obs, F64B_OC);
        const int64_t part_bytes = a.ws_part * (int64_t)sizeof(double);
        while (S > 1 && (int64_t)(S - 1) * part_bytes > ((int64_t)2 << 30)) --S;
        if (S > 1 && hipMallocAsync((void**)&ws, (size_t)(S - 1) * part_bytes, st) == hipSuccess) {
            a.n_split = S;
            a.ws = ws;
        } else {
            (void)hipGetLastError();       // no scratch: one workgroup per bin, as before
            ws = nullptr;
        }
    }
    hipStream_t st_nl = st;
    if (fork) {
        // (a fork that cannot be set up is not an error: the planes then run behind the CSM on the caller's stream)
        if (hipEventRecord(ev_fork, st) == hipSuccess && hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess) st_nl = side;
        else (void)hipGetLastError();
    }
    if (which & SC_PLANE_CSM) {
        a.plane = sc_plane_offset(planes, SC_PLANE_CSM);
        const int need = (a.n_tiles + 3) / 4;
        if (need <= 1) rc = launch_csm_f64<1>(a, st);
        else if (need <= 3) rc = launch_csm_f64<3>(a, st);
        else if (need <= 5) rc = launch_csm_f64<5>(a, st);
        else rc = launch_csm_f64<9>(a, st);
    }
    uint32_t w = rc ? 0u : (which & ~SC_PLANE_CSM);
    st = st_nl;
    do {        // (one exit: the side stream is joined below whether or not a launch failed)
        if (a.C >= 48 && !sc_switch(SC_SW_F64_NO_BLOCK)) {      // 64 x 64 blocks, one plane per launch (see nonlinear_f64_block_kernel)
            if (w & SC_PLANE_ABS_IM) { if ((rc = launch_nl_f64_block<SC_PLANE_ABS_IM>(a, st))) break; w &= ~SC_PLANE_ABS_IM; }
            if (w & SC_PLANE_IM_SQ) { if ((rc = launch_nl_f64_block<SC_PLANE_IM_SQ>(a, st))) break; w &= ~SC_PLANE_IM_SQ; }
            if (w & SC_PLANE_SIGN_IM) { if ((rc = launch_nl_f64_block<SC_PLANE_SIGN_IM>(a, st))) break; w &= ~SC_PLANE_SIGN_IM; }
        }
        if ((w & (SC_PLANE_ABS_IM | SC_PLANE_IM_SQ)) == (SC_PLANE_ABS_IM | SC_PLANE_IM_SQ)) {
            if ((rc = launch_nl_f64<SC_PLANE_ABS_IM | SC_PLANE_IM_SQ, 2>(a, st))) break;
            w &= ~(SC_PLANE_ABS_IM | SC_PLANE_IM_SQ);
        }
        if (w & SC_PLANE_ABS_IM) { if ((rc = launch_nl_f64<SC_PLANE_ABS_IM, 3>(a, st))) break; }
        if (w & SC_PLANE_IM_SQ) { if ((rc = launch_nl_f64<SC_PLANE_IM_SQ, 3>(a, st))) break; }
        if (w & SC_PLANE_SIGN_IM) { if ((rc = launch_nl_f64<SC_PLANE_SIGN_IM, 3>(a, st))) break; }
        if (w & SC_PLANE_UNIT) {
            // sum s / |s| = the cross-spectral matrix of the unit phasors x / |x|: the matrix-core kernel with the
            // normalisation in its staging (per pair and observation on the VALU -- square root and two divisions in
            // fp64 -- it took 130 ms at cfg3 against 9 for the CSM)
            F64Args u = a;
            u.plane = sc_plane_offset(planes, SC_PLANE_UNIT);
            const int need = (u.n_tiles + 3) / 4;
            if (need <= 1) rc = launch_csm_f64<1, true>(u, st);
            else if (need <= 3) rc = launch_csm_f64<3, true>(u, st);
            else if (need <= 5) rc = launch_csm_f64<5, true>(u, st);
            else rc = launch_csm_f64<9, true>(u, st);
            if (rc) break;
        }
    } while (0);
    if (st_nl != (hipStream_t)stream) {
        // the join must happen whatever the launches returned: the side stream may hold work that reads the scratch
        if (hipEventRecord(ev_join, side) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, ev_join, 0) != hipSuccess) {
            (void)hipStreamSynchronize(side);
            if (rc == SC_OK) { sc_set_error("sc_accumulate_f64: joining the side stream failed"); rc = SC_EHIP; }
        }
    }
    return rc;            // (ws_guard releases the scratch behind everything queued on the caller's stream)
}
