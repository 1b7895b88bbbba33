// sc_nonlinear.hip -- per-observation non-linear accumulator planes on the VALU.
//
// PLI / wPLI / PLV / PPC apply a non-linearity to every per-observation cross-spectrum
// s = x_i conj(x_j) BEFORE the expectation (connectivity.py:897-1159), so they are not a
// GEMM: this kernel forms s in registers and accumulates
//     ABS_IM  sum |Im s|      IM_SQ  sum (Im s)^2      SIGN_IM  sum sign(Im s)
//     UNIT    sum s/|s|  (0/0 -> NaN exactly like the reference's x/abs(x))
// without ever materialising the (W,R,K,N,C,C) temporary of the reference.
//
// Workgroup = one bin and one set of up to MAXB 32x32 channel blocks (upper triangle of
// the block grid).  A wave is an 8x8 lane grid, each lane owns a 4x4 register tile of
// pairs: 8 LDS complex loads (4x ds_read_b128) feed 16 pairs per observation.  The 4 waves
// take observations o = wave (mod 4) of every staged chunk and are summed through LDS at
// the end, so the work is balanced for any channel count.
#include "sc_stage.h"

struct NlArgs {
    ScStage st;
    float* accum;
    int64_t floats_per_bin;
    int n_bins, F, NB, n_tiles, NB32, n_blocks32, n_block_sets;
    uint32_t planes;
};

template <uint32_t WHICH>
struct NlPlanes {
    static constexpr int N = ((WHICH & SC_PLANE_ABS_IM) ? 1 : 0) + ((WHICH & SC_PLANE_IM_SQ) ? 1 : 0) +
                             ((WHICH & SC_PLANE_SIGN_IM) ? 1 : 0) + ((WHICH & SC_PLANE_UNIT) ? 2 : 0);
};

template <uint32_t WHICH, int MAXB, int OC, int CPMAX, bool VEC>
__global__ void __launch_bounds__(256) nonlinear_kernel(NlArgs p) {
    extern __shared__ __align__(16) float lds[];
    constexpr int NP = NlPlanes<WHICH>::N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int bs = jj % p.n_block_sets;
    const int bin = (jj / p.n_block_sets) * 8 + xcd;
    if (bin >= p.n_bins) return;
    const int g = bin / p.F, f = bin - g * p.F;

    ScStage st = p.st;
    st.base = p.st.base + (int64_t)f * st.ax.sF + sc_group_offset(st.ax, g);

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

    float acc[MAXB][NP][16];
#pragma unroll
    for (int s = 0; s < MAXB; ++s)
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][q][e] = 0.f;

    const int li = lane >> 3, lj = lane & 7;
    const int buf_floats = OC * st.RS;
    const int n_chunks = (st.n_obs + OC - 1) / OC;
    ScStageRegs<OC, CPMAX, VEC> regs;
    sc_stage_load<OC, CPMAX, VEC>(st, 0, tid, regs);
    sc_stage_store<OC, CPMAX, VEC>(st, lds, tid, regs);
    __syncthreads();

    for (int ch = 0; ch < n_chunks; ++ch) {
        const float* cur = lds + (ch & 1) * buf_floats;
        float* nxt = lds + ((ch + 1) & 1) * buf_floats;
        const bool more = ch + 1 < n_chunks;
        if (more) sc_stage_load<OC, CPMAX, VEC>(st, (ch + 1) * OC, tid, regs);
        // rows past n_obs are zero: every plane gets +0 (sign(0)=0) except UNIT (0/0=NaN),
        // so bound the row loop by the real observation count.
        const int rows = min(OC, st.n_obs - ch * OC);
        for (int row = wave; row < rows; row += 4) {
            const float* rp = cur + row * st.RS;
#pragma unroll
            for (int s = 0; s < MAXB; ++s) {
                {   // invalid slots (last block set) recompute block (0,0) and are never stored
                    float2 xi[4], xj[4];
                    const float4* pi = reinterpret_cast<const float4*>(rp + (BI[s] * 32 + li * 4) * 2);
                    const float4* pj = reinterpret_cast<const float4*>(rp + (BJ[s] * 32 + lj * 4) * 2);
                    const float4 i0 = pi[0], i1 = pi[1], j0 = pj[0], j1 = pj[1];
                    xi[0] = make_float2(i0.x, i0.y); xi[1] = make_float2(i0.z, i0.w);
                    xi[2] = make_float2(i1.x, i1.y); xi[3] = make_float2(i1.z, i1.w);
                    xj[0] = make_float2(j0.x, j0.y); xj[1] = make_float2(j0.z, j0.w);
                    xj[2] = make_float2(j1.x, j1.y); xj[3] = make_float2(j1.z, j1.w);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const float imv = xi[a].y * xj[b].x - xi[a].x * xj[b].y;
                            int q = 0;
                            if constexpr (WHICH & SC_PLANE_ABS_IM) { acc[s][q][a * 4 + b] += fabsf(imv); ++q; }
                            if constexpr (WHICH & SC_PLANE_IM_SQ) { acc[s][q][a * 4 + b] += imv * imv; ++q; }
                            if constexpr (WHICH & SC_PLANE_SIGN_IM) {
                                acc[s][q][a * 4 + b] += (imv > 0.f ? 1.f : 0.f) - (imv < 0.f ? 1.f : 0.f);
                                ++q;
                            }
                            if constexpr (WHICH & SC_PLANE_UNIT) {
                                const float rev = xi[a].x * xj[b].x + xi[a].y * xj[b].y;
                                const float inv = rsqrtf(rev * rev + imv * imv);  // 0 -> inf -> 0*inf = NaN
                                acc[s][q][a * 4 + b] += rev * inv;
                                acc[s][q + 1][a * 4 + b] += imv * inv;
                            }
                        }
                    }
                }
            }
        }
        if (more) sc_stage_store<OC, CPMAX, VEC>(st, nxt, tid, regs);
        __syncthreads();
    }

    // cross-wave reduction through LDS, one (block, plane) at a time: 4 waves x 16 x 64 floats
    float* red = lds;
    float* out_bin = p.accum + (int64_t)bin * p.floats_per_bin;
#pragma unroll
    for (int s = 0; s < MAXB; ++s) {
        if (!valid[s]) continue;       // wave-uniform and identical for all 4 waves
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[s][q][e];
            __syncthreads();
            // plane offset of accumulator q
            int plane;
            {
                int k = 0;
                plane = -1;
                if constexpr (WHICH & SC_PLANE_ABS_IM) { if (q == k) plane = sc_plane_offset(p.planes, SC_PLANE_ABS_IM); ++k; }
                if constexpr (WHICH & SC_PLANE_IM_SQ) { if (q == k) plane = sc_plane_offset(p.planes, SC_PLANE_IM_SQ); ++k; }
                if constexpr (WHICH & SC_PLANE_SIGN_IM) { if (q == k) plane = sc_plane_offset(p.planes, SC_PLANE_SIGN_IM); ++k; }
                if constexpr (WHICH & SC_PLANE_UNIT) {
                    if (q == k) plane = sc_plane_offset(p.planes, SC_PLANE_UNIT);
                    if (q == k + 1) plane = sc_plane_offset(p.planes, SC_PLANE_UNIT) + 1;
                }
            }
            float* out = out_bin + (int64_t)plane * p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int e = wave * 4 + e4;
                const float v = red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane] +
                                red[(2 * 16 + e) * 64 + lane] + red[(3 * 16 + e) * 64 + lane];
                const int i = BI[s] * 32 + li * 4 + (e >> 2), j = BJ[s] * 32 + lj * 4 + (e & 3);
                const int ti = i >> 4, tj = j >> 4;
                if (ti <= tj && tj < p.NB)
                    out[(int64_t)sc_tile_index(ti, tj, p.NB) * SC_TILE_ELEMS + (i & 15) * 16 + (j & 15)] = v;
            }
            __syncthreads();
        }
    }
}

template <uint32_t WHICH, int MAXB, int OC, int CPMAX>
static int launch_nl(const NlArgs& a, bool vec, hipStream_t stream) {
    NlArgs args = a;
    args.n_block_sets = (a.n_blocks32 + MAXB - 1) / MAXB;
    const int bins8 = (a.n_bins + 7) / 8;
    const unsigned grid = (unsigned)(bins8 * 8 * args.n_block_sets);
    size_t shmem = (size_t)2 * OC * a.st.RS * sizeof(float);
    if (shmem < (size_t)4 * 16 * 64 * sizeof(float)) shmem = (size_t)4 * 16 * 64 * sizeof(float);
    if (vec) {
        auto k = nonlinear_kernel<WHICH, MAXB, OC, CPMAX, true>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), shmem, stream, args);
    } else {
        auto k = nonlinear_kernel<WHICH, MAXB, OC, CPMAX, false>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), shmem, stream, args);
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

template <uint32_t WHICH, int MAXB>
static int dispatch_nl(const NlArgs& a, bool vec, hipStream_t stream) {
    if (a.st.CP <= 128) return launch_nl<WHICH, MAXB, 16, 128>(a, vec, stream);
    return launch_nl<WHICH, MAXB, 8, 256>(a, vec, stream);
}

extern "C" int sc_nonlinear_accumulate_f32(const void* d_X, const sc_spectra_desc* desc, uint32_t planes,
                                           uint32_t which, float* d_accum, void* stream) {
    ScTimed timed_("nonlinear_valu", stream);
    SC_REQUIRE(d_X && desc && d_accum, "NULL argument");
    const uint32_t nl_mask = SC_PLANE_ABS_IM | SC_PLANE_IM_SQ | SC_PLANE_SIGN_IM | SC_PLANE_UNIT;
    SC_REQUIRE((which & ~nl_mask) == 0 && which != 0, "which must name non-linear planes only");
    SC_REQUIRE((which & planes) == which, "which must be a subset of planes");
    ScAxes ax;
    sc_make_axes(desc, &ax);
    SC_REQUIRE(ax.C >= 1 && ax.F >= 1 && ax.n_obs >= 1 && ax.n_groups >= 1, "empty dimension");
    if (ax.C > SC_MAX_SIGNALS) {
        sc_set_error("n_signals=%d exceeds SC_MAX_SIGNALS=%d", ax.C, SC_MAX_SIGNALS);
        return SC_EUNSUPPORTED;
    }
    NlArgs a;
    a.NB = sc_n_blocks(ax.C);
    a.n_tiles = sc_n_tiles(a.NB);
    a.NB32 = (ax.C + 31) / 32;
    a.n_blocks32 = a.NB32 * (a.NB32 + 1) / 2;
    a.n_bins = ax.n_groups * ax.F;
    a.F = ax.F;
    a.planes = planes;
    a.floats_per_bin = (int64_t)sc_plane_count(planes) * a.n_tiles * SC_TILE_ELEMS;
    a.accum = d_accum;
    a.n_block_sets = 1;
    a.st.base = (const float2*)d_X;
    a.st.ax = ax;
    a.st.obs_stride = sc_stage_linear_stride(ax);
    a.st.C = ax.C;
    a.st.CP = a.NB32 * 32;
    a.st.RS = sc_row_stride(a.st.CP);
    a.st.n_obs = ax.n_obs;
    const bool vec = sc_stage_vec_ok(d_X, ax);
    hipStream_t s = (hipStream_t)stream;
    int rc = SC_OK;
    // decompose into the instantiated plane sets (register budget: <= 2 planes x 5 blocks)
    uint32_t w = which;
    if ((w & (SC_PLANE_ABS_IM | SC_PLANE_IM_SQ)) == (SC_PLANE_ABS_IM | SC_PLANE_IM_SQ)) {
        rc = dispatch_nl<SC_PLANE_ABS_IM | SC_PLANE_IM_SQ, 5>(a, vec, s);
        if (rc) return rc;
        w &= ~(SC_PLANE_ABS_IM | SC_PLANE_IM_SQ);
    }
    if (w & SC_PLANE_ABS_IM) { rc = dispatch_nl<SC_PLANE_ABS_IM, 5>(a, vec, s); if (rc) return rc; }
    if (w & SC_PLANE_IM_SQ) { rc = dispatch_nl<SC_PLANE_IM_SQ, 5>(a, vec, s); if (rc) return rc; }
    if (w & SC_PLANE_SIGN_IM) { rc = dispatch_nl<SC_PLANE_SIGN_IM, 5>(a, vec, s); if (rc) return rc; }
    if (w & SC_PLANE_UNIT) { rc = dispatch_nl<SC_PLANE_UNIT, 5>(a, vec, s); if (rc) return rc; }
    return rc;
}
