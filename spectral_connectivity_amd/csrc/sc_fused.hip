// sc_fused.hip -- stage B in ONE pass over the spectra: cross-spectral matrix on the matrix
// cores AND the sum |Im s| plane (wPLI weights) on the VALU, concurrently.
//
// Workgroup = 8 waves = one output bin (C <= 128).  The CDNA4 CU places two waves on each of
// its 4 SIMDs; here every SIMD gets one MFMA wave (waves 0-3: upper 16x16 tiles of
// S = sum_o x_o x_o^H, exactly sc_csm.hip's inner loop) and one VALU wave (waves 4-7: 32x32
// channel blocks, 4x4 register tile of pairs per lane, acc += |Im(x_i conj x_j)| at 3 VALU
// instructions per pair, exactly sc_nonlinear.hip's inner loop).  The matrix pipe and the
// VALU arbitrate separately, so the two roles overlap instead of queueing, and both read the
// SAME LDS-staged observation rows: the 6.5 GB spectra of the headline configuration are
// read from HBM once for both products (the unfused path reads them twice).
//
// VALU waves: the <= 10 upper 32x32 blocks are split into n_sets sets of <= 5 blocks
// (80 accumulator VGPRs); the 4/n_sets waves of a set take interleaved rows of every staged
// chunk and are summed through LDS at the end.
#include "sc_stage.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FusedArgs {
    ScStage st;
    float* accum;
    int64_t floats_per_bin;
    int n_bins, F, NB, n_tiles, NB32, n_blocks32, n_sets;
    int csm_plane, abs_plane;
};

template <int OC, int CPMAX, bool VEC>
struct ScStageRegs512 {   // 512 threads move the chunk: half the per-thread elements
    static constexpr int E = VEC ? (OC * CPMAX / 2 / 512) : (OC * CPMAX / 512);
    float4 v4[VEC ? E : 1];
    float2 v2[VEC ? 1 : E];
};

template <int OC, int CPMAX, bool VEC>
__device__ inline void stage_load512(const ScStage& st, int o0, int tid, ScStageRegs512<OC, CPMAX, VEC>& r) {
    constexpr int E = ScStageRegs512<OC, CPMAX, VEC>::E;
    if constexpr (VEC) {
        const int half = st.CP >> 1, total = OC * half;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int e = tid + i * 512;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < total) {
                const int row = e / half, q = e - row * half;
                const int o = o0 + row, c = 2 * q;
                if (o < st.n_obs && c < st.C)
                    v = *reinterpret_cast<const float4*>(st.base + sc_stage_obs_offset(st, o) + c);
            }
            r.v4[i] = v;
        }
    } else {
        const int total = OC * st.CP;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int e = tid + i * 512;
            float2 v = make_float2(0.f, 0.f);
            if (e < total) {
                const int row = e / st.CP, c = e - row * st.CP;
                const int o = o0 + row;
                if (o < st.n_obs && c < st.C) v = st.base[sc_stage_obs_offset(st, o) + c];
            }
            r.v2[i] = v;
        }
    }
}

template <int OC, int CPMAX, bool VEC>
__device__ inline void stage_store512(const ScStage& st, float* lds, int tid,
                                      const ScStageRegs512<OC, CPMAX, VEC>& r) {
    constexpr int E = ScStageRegs512<OC, CPMAX, VEC>::E;
    if constexpr (VEC) {
        const int half = st.CP >> 1, total = OC * half;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int e = tid + i * 512;
            if (e < total) {
                const int row = e / half, q = e - row * half;
                *reinterpret_cast<float4*>(lds + row * st.RS + 4 * q) = r.v4[i];
            }
        }
    } else {
        const int total = OC * st.CP;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int e = tid + i * 512;
            if (e < total) {
                const int row = e / st.CP, c = e - row * st.CP;
                *reinterpret_cast<float2*>(lds + row * st.RS + 2 * c) = r.v2[i];
            }
        }
    }
}

// The two roles live in separate functions so their accumulator registers never coexist
// (register allocation = max of the two bodies, not the sum).  Both bodies execute the same
// sequence of workgroup barriers: 1 (prologue) + n_chunks + 2*log2(waves per set).
template <int MAX_SLOTS, bool VEC>
__device__ __forceinline__ void fused_mfma_role(const FusedArgs& p, const ScStage& st, float* lds, int tid,
                                                int wave, int bin) {
    constexpr int OC = 32, CPMAX = 128;
    const int lane = tid & 63;
    int bi[MAX_SLOTS], bj[MAX_SLOTS];
    f32x4 re[MAX_SLOTS], im[MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s) {
        const int t = s * 4 + wave;
        int r = 0, rem = (t < p.n_tiles) ? t : 0, len = p.NB;
        while (rem >= len) { rem -= len; ++r; --len; }
        bi[s] = r; bj[s] = r + rem;
        re[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; im[s] = re[s];
    }
    const int buf_floats = OC * st.RS;
    const int n_chunks = (st.n_obs + OC - 1) / OC;
    ScStageRegs512<OC, CPMAX, VEC> regs;
    stage_load512<OC, CPMAX, VEC>(st, 0, tid, regs);
    stage_store512<OC, CPMAX, VEC>(st, lds, tid, regs);
    __syncthreads();
    const int frag_row = lane >> 4, frag_col = lane & 15;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const float* cur = lds + (ch & 1) * buf_floats;
        float* nxt = lds + ((ch + 1) & 1) * buf_floats;
        const bool more = ch + 1 < n_chunks;
        if (more) stage_load512<OC, CPMAX, VEC>(st, (ch + 1) * OC, tid, regs);
#pragma unroll 2
        for (int kk = 0; kk < OC / 4; ++kk) {
            const float* rowp = cur + (kk * 4 + frag_row) * st.RS + 2 * frag_col;
#pragma unroll
            for (int s = 0; s < MAX_SLOTS; ++s) {
                const float2 a = *reinterpret_cast<const float2*>(rowp + 32 * bi[s]);
                const float2 b = *reinterpret_cast<const float2*>(rowp + 32 * bj[s]);
                re[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, re[s], 0, 0, 0);
                im[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.x, im[s], 0, 0, 0);
                re[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, re[s], 0, 0, 0);
                im[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a.x, b.y, im[s], 0, 0, 0);
            }
        }
        if (more) stage_store512<OC, CPMAX, VEC>(st, nxt, tid, regs);
        __syncthreads();
    }
    float* out = p.accum + (int64_t)bin * p.floats_per_bin + (int64_t)p.csm_plane * p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s) {
        const int t = s * 4 + wave;
        if (t < p.n_tiles) {
            float* o_re = out + (int64_t)t * SC_TILE_ELEMS;
            float* o_im = o_re + (int64_t)p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = ((lane >> 4) * 4 + r) * 16 + (lane & 15);
                o_re[idx] = re[s][r];
                o_im[idx] = im[s][r];
            }
        }
    }
    const int wps = 4 / p.n_sets;
    for (int half = wps >> 1; half >= 1; half >>= 1) { __syncthreads(); __syncthreads(); }
}

template <bool VEC>
__device__ __forceinline__ void fused_valu_role(const FusedArgs& p, const ScStage& st, float* lds, int tid,
                                                int vw, int bin) {
    constexpr int OC = 32, CPMAX = 128, MAXB = 5;
    const int lane = tid & 63;
    const int wps = 4 / p.n_sets;                         // VALU waves per block set
    const int set = vw / wps, rsub = vw % wps;
    int BI[MAXB], BJ[MAXB];
    float acc[MAXB][16];
#pragma unroll
    for (int s = 0; s < MAXB; ++s) {
        const int t = set * MAXB + s;
        int r = 0, rem = (t < p.n_blocks32) ? t : 0, len = p.NB32;
        while (rem >= len) { rem -= len; ++r; --len; }
        BI[s] = r; BJ[s] = r + rem;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
    }
    const int buf_floats = OC * st.RS;
    const int n_chunks = (st.n_obs + OC - 1) / OC;
    ScStageRegs512<OC, CPMAX, VEC> regs;
    stage_load512<OC, CPMAX, VEC>(st, 0, tid, regs);
    stage_store512<OC, CPMAX, VEC>(st, lds, tid, regs);
    __syncthreads();
    const int li = lane >> 3, lj = lane & 7;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const float* cur = lds + (ch & 1) * buf_floats;
        float* nxt = lds + ((ch + 1) & 1) * buf_floats;
        const bool more = ch + 1 < n_chunks;
        if (more) stage_load512<OC, CPMAX, VEC>(st, (ch + 1) * OC, tid, regs);
        // zero rows past n_obs contribute |0| = 0: no bound needed for this plane
        for (int row = rsub; row < OC; row += wps) {
            const float* rp = cur + row * st.RS;
#pragma unroll
            for (int s = 0; s < MAXB; ++s) {
                const float4* pi = reinterpret_cast<const float4*>(rp + (BI[s] * 32 + li * 4) * 2);
                const float4* pj = reinterpret_cast<const float4*>(rp + (BJ[s] * 32 + lj * 4) * 2);
                const float4 i0 = pi[0], i1 = pi[1], j0 = pj[0], j1 = pj[1];
                const float xi_re[4] = {i0.x, i0.z, i1.x, i1.z}, xi_im[4] = {i0.y, i0.w, i1.y, i1.w};
                const float xj_re[4] = {j0.x, j0.z, j1.x, j1.z}, xj_im[4] = {j0.y, j0.w, j1.y, j1.w};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[s][a * 4 + b] += fabsf(xi_im[a] * xj_re[b] - xi_re[a] * xj_im[b]);
            }
        }
        if (more) stage_store512<OC, CPMAX, VEC>(st, nxt, tid, regs);
        __syncthreads();
    }
    // tree-sum the row-split partials of a set through LDS
    float* red = lds;   // [set*2 + writer][MAXB*16][64]
    for (int half = wps >> 1; half >= 1; half >>= 1) {
        if (rsub >= half && rsub < 2 * half) {
            float* dst = red + (size_t)(set * 2 + (rsub - half)) * (MAXB * 16 * 64);
#pragma unroll
            for (int s = 0; s < MAXB; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[(s * 16 + e) * 64 + lane] = acc[s][e];
        }
        __syncthreads();
        if (rsub < half) {
            const float* src = red + (size_t)(set * 2 + rsub) * (MAXB * 16 * 64);
#pragma unroll
            for (int s = 0; s < MAXB; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s][e] += src[(s * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (rsub == 0) {
        float* out = p.accum + (int64_t)bin * p.floats_per_bin + (int64_t)p.abs_plane * p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
        for (int s = 0; s < MAXB; ++s) {
            if (set * MAXB + s < p.n_blocks32) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = BI[s] * 32 + li * 4 + (e >> 2), j = BJ[s] * 32 + lj * 4 + (e & 3);
                    const int ti = i >> 4, tj = j >> 4;
                    if (ti <= tj && tj < p.NB)
                        out[(int64_t)sc_tile_index(ti, tj, p.NB) * SC_TILE_ELEMS + (i & 15) * 16 + (j & 15)] =
                            acc[s][e];
                }
            }
        }
    }
}

template <int MAX_SLOTS, bool VEC>
__global__ void __launch_bounds__(512) fused_csm_absim_kernel(FusedArgs p) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bin = blockIdx.x;   // one workgroup per bin: consecutive bins on consecutive XCDs
    const int g = bin / p.F, f = bin - g * p.F;
    ScStage st = p.st;
    st.base = p.st.base + (int64_t)f * st.ax.sF + sc_group_offset(st.ax, g);
    if (wave < 4) fused_mfma_role<MAX_SLOTS, VEC>(p, st, lds, tid, wave, bin);
    else fused_valu_role<VEC>(p, st, lds, tid, wave - 4, bin);
}

template <int MAX_SLOTS>
static int launch_fused(const FusedArgs& a, bool vec, hipStream_t stream) {
    size_t shmem = (size_t)2 * 32 * a.st.RS * sizeof(float);
    const size_t red = (size_t)4 * 5 * 16 * 64 * sizeof(float);
    if (shmem < red) shmem = red;
    if (vec) {
        auto k = fused_csm_absim_kernel<MAX_SLOTS, true>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3((unsigned)a.n_bins), dim3(512), shmem, stream, a);
    } else {
        auto k = fused_csm_absim_kernel<MAX_SLOTS, false>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3((unsigned)a.n_bins), dim3(512), shmem, stream, a);
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

extern "C" int sc_fused_supported(int64_t n_signals) { return (n_signals >= 1 && n_signals <= 128) ? 1 : 0; }

extern "C" int sc_fused_csm_absim_f32(const void* d_X, const sc_spectra_desc* desc, uint32_t planes,
                                      float* d_accum, void* stream) {
    SC_REQUIRE(d_X && desc && d_accum, "NULL argument");
    SC_REQUIRE((planes & (SC_PLANE_CSM | SC_PLANE_ABS_IM)) == (SC_PLANE_CSM | SC_PLANE_ABS_IM),
               "planes must contain SC_PLANE_CSM and SC_PLANE_ABS_IM");
    ScAxes ax;
    sc_make_axes(desc, &ax);
    SC_REQUIRE(ax.C >= 1 && ax.F >= 1 && ax.n_obs >= 1 && ax.n_groups >= 1, "empty dimension");
    if (!sc_fused_supported(ax.C)) {
        sc_set_error("fused CSM+|Im| kernel supports n_signals <= 128 (got %d)", ax.C);
        return SC_EUNSUPPORTED;
    }
    FusedArgs a;
    a.NB = sc_n_blocks(ax.C);
    a.n_tiles = sc_n_tiles(a.NB);
    a.NB32 = (ax.C + 31) / 32;
    a.n_blocks32 = a.NB32 * (a.NB32 + 1) / 2;
    a.n_sets = (a.n_blocks32 + 4) / 5;        // 1 or 2
    a.n_bins = ax.n_groups * ax.F;
    a.F = ax.F;
    a.floats_per_bin = (int64_t)sc_plane_count(planes) * a.n_tiles * SC_TILE_ELEMS;
    a.csm_plane = sc_plane_offset(planes, SC_PLANE_CSM);
    a.abs_plane = sc_plane_offset(planes, SC_PLANE_ABS_IM);
    a.accum = d_accum;
    a.st.base = (const float2*)d_X;
    a.st.ax = ax;
    a.st.obs_stride = sc_stage_linear_stride(ax);
    a.st.C = ax.C;
    a.st.CP = a.NB32 * 32;                    // VALU blocks need 32-channel padding
    a.st.RS = sc_row_stride(a.st.CP);
    a.st.n_obs = ax.n_obs;
    const bool vec = sc_stage_vec_ok(d_X, ax);
    hipStream_t s = (hipStream_t)stream;
    const int need = (a.n_tiles + 3) / 4;
    if (need <= 1) return launch_fused<1>(a, vec, s);
    if (need <= 3) return launch_fused<3>(a, vec, s);
    if (need <= 5) return launch_fused<5>(a, vec, s);
    return launch_fused<9>(a, vec, s);
}
