// sc_csm.hip -- expectation of the cross-spectral matrix on the matrix cores.
//
// For every output bin (group g of kept axes, frequency f) the reference forms one C x C
// outer product per observation and averages them (connectivity.py:447-492, :1799-1822).
// Here the same sum is a Hermitian rank-n_obs update  S = sum_o x_o x_o^H  per bin:
//     Re S_ij = sum_o (xr_i xr_j + xi_i xi_j)      Im S_ij = sum_o (xi_i xr_j - xr_i xi_j)
// i.e. four real f32 MFMA accumulations (v_mfma_f32_16x16x4_f32, exact f32 FMA chains) per
// 16x16 channel tile and 4 observations.  Only upper tiles (bi <= bj) are computed and
// stored (packed tile layout of sc_hip.h); the epilogue mirrors them.
//
// Workgroup = 4 waves = one bin and one group of up to 4*MAX_SLOTS tiles.  Observation rows
// are staged HBM -> registers -> LDS in chunks of OC rows (double buffered, one barrier per
// chunk); each wave reads its A/B fragments with ds_read_b64 (re,im interleaved).
// blockIdx -> (bin, tile group) is XCD-aware: the tile groups of one bin are consecutive
// workgroups of the SAME XCD (block b runs on XCD b % 8), so the slab they share is read
// from HBM once and re-served by that XCD's L2.
#include "sc_stage.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct CsmArgs {
    ScStage st;          // st.base = X (group/bin offsets are added in-kernel)
    float* accum;
    int64_t floats_per_bin;
    int n_bins, F, NB, n_tiles, n_tile_groups;
    int csm_plane;       // plane offset of the CSM planes inside a bin record
};

template <int MAX_SLOTS, int OC, int CPMAX, bool VEC>
__global__ void __launch_bounds__(256, (CPMAX == 256 && MAX_SLOTS <= 5) ? 2 : 1) csm_mfma_kernel(CsmArgs p) {
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware decode of blockIdx.x -> (bin, tile group)
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int tg = j % p.n_tile_groups;
    const int bin = (j / p.n_tile_groups) * 8 + xcd;
    if (bin >= p.n_bins) return;
    const int g = bin / p.F, f = bin - g * p.F;

    ScStage st = p.st;
    st.base = p.st.base + (int64_t)f * st.ax.sF + sc_group_offset(st.ax, g);

    // tiles of this wave: t = (tg*MAX_SLOTS + s)*4 + wave
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

    f32x4 re[MAX_SLOTS], im[MAX_SLOTS];
#pragma unroll
    for (int s = 0; s < MAX_SLOTS; ++s) { re[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; im[s] = re[s]; }

    const int buf_floats = OC * st.RS;
    const int n_chunks = (st.n_obs + OC - 1) / OC;
    ScStageRegs<OC, CPMAX, VEC> regs;
    sc_stage_load<OC, CPMAX, VEC>(st, 0, tid, regs);
    sc_stage_store<OC, CPMAX, VEC>(st, lds, tid, regs);
    __syncthreads();

    const int frag_row = lane >> 4, frag_col = lane & 15;
    float* out = p.accum + (int64_t)bin * p.floats_per_bin + (int64_t)p.csm_plane * p.n_tiles * SC_TILE_ELEMS;
    for (int ch = 0; ch < n_chunks; ++ch) {
        const float* cur = lds + (ch & 1) * buf_floats;
        float* nxt = lds + ((ch + 1) & 1) * buf_floats;
        const bool more = ch + 1 < n_chunks;
        if (more) sc_stage_load<OC, CPMAX, VEC>(st, (ch + 1) * OC, tid, regs);
#pragma unroll 2
        for (int kk = 0; kk < OC / 4; ++kk) {
            const float* rowp = cur + (kk * 4 + frag_row) * st.RS + 2 * frag_col;
#pragma unroll
            // invalid slots (last tile group only) recompute tile (0,0) and are never stored:
            // keeping the body branch-free lets the compiler run the ds_reads of slot s+1
            // under the MFMAs of slot s.
            for (int s = 0; s < MAX_SLOTS; ++s) {
                const float2 a = *reinterpret_cast<const float2*>(rowp + 32 * bi[s]);
                const float2 b = *reinterpret_cast<const float2*>(rowp + 32 * bj[s]);
                re[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, re[s], 0, 0, 0);
                im[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.x, im[s], 0, 0, 0);
                re[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, re[s], 0, 0, 0);
                im[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a.x, b.y, im[s], 0, 0, 0);
            }
        }
        // Two-level summation (see sc_fused.hip): fold the accumulators into the output record
        // every 512 observations so no f32 chain grows with n_obs.
        // D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
        constexpr int FLUSH = 512 / OC;
        if (((ch + 1) % FLUSH) == 0 || !more) {
            const bool first = ch < FLUSH;
#pragma unroll
            for (int s = 0; s < MAX_SLOTS; ++s) {
                if (valid[s]) {
                    const int t = (tg * MAX_SLOTS + s) * 4 + wave;
                    float* o_re = out + (int64_t)t * SC_TILE_ELEMS;
                    float* o_im = o_re + (int64_t)p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = ((lane >> 4) * 4 + r) * 16 + (lane & 15);
                        o_re[idx] = first ? re[s][r] : o_re[idx] + re[s][r];
                        o_im[idx] = first ? im[s][r] : o_im[idx] + im[s][r];
                    }
                }
                re[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; im[s] = re[s];
            }
        }
        if (more) sc_stage_store<OC, CPMAX, VEC>(st, nxt, tid, regs);
        __syncthreads();
    }

}

template <int MAX_SLOTS, int OC, int CPMAX>
static int launch_csm(const CsmArgs& a, bool vec, hipStream_t stream) {
    const int tiles_per_group = 4 * MAX_SLOTS;
    CsmArgs args = a;
    args.n_tile_groups = (a.n_tiles + tiles_per_group - 1) / tiles_per_group;
    const int bins8 = (a.n_bins + 7) / 8;
    const unsigned grid = (unsigned)(bins8 * 8 * args.n_tile_groups);
    const size_t shmem = (size_t)2 * OC * a.st.RS * sizeof(float);
    if (vec) {
        auto k = csm_mfma_kernel<MAX_SLOTS, OC, CPMAX, true>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), shmem, stream, args);
    } else {
        auto k = csm_mfma_kernel<MAX_SLOTS, OC, CPMAX, false>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), shmem, stream, args);
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

static int csm_accumulate(const void* d_X, const sc_spectra_desc* desc, uint32_t planes, uint32_t into, float* d_accum,
                          void* stream) {
    ScTimed timed_("csm_mfma", stream);
    SC_REQUIRE(d_X && desc && d_accum, "NULL argument");
    SC_REQUIRE(planes & into, "planes must contain the plane to fill");
    ScAxes ax;
    sc_make_axes(desc, &ax);
    SC_REQUIRE(ax.C >= 1 && ax.F >= 1 && ax.n_obs >= 1 && ax.n_groups >= 1, "empty dimension");
    if (ax.C > SC_MAX_SIGNALS) {
        sc_set_error("n_signals=%d exceeds SC_MAX_SIGNALS=%d", ax.C, SC_MAX_SIGNALS);
        return SC_EUNSUPPORTED;
    }
    CsmArgs a;
    a.NB = sc_n_blocks(ax.C);
    a.n_tiles = sc_n_tiles(a.NB);
    a.n_bins = ax.n_groups * ax.F;
    a.F = ax.F;
    a.floats_per_bin = (int64_t)sc_plane_count(planes) * a.n_tiles * SC_TILE_ELEMS;
    a.csm_plane = sc_plane_offset(planes, into);
    a.accum = d_accum;
    a.n_tile_groups = 1;
    a.st.base = (const float2*)d_X;
    a.st.ax = ax;
    a.st.obs_stride = sc_stage_linear_stride(ax);
    a.st.C = ax.C;
    a.st.CP = a.NB * SC_TILE;
    a.st.RS = sc_row_stride(a.st.CP);
    a.st.n_obs = ax.n_obs;
    const bool vec = sc_stage_vec_ok(d_X, ax);
    hipStream_t st = (hipStream_t)stream;
    // slots per wave: smallest instantiation that covers all tiles in one tile group
    const int need = (a.n_tiles + 3) / 4;
    if (a.st.CP <= 128) {
        if (need <= 1) return launch_csm<1, 32, 128>(a, vec, st);
        if (need <= 3) return launch_csm<3, 32, 128>(a, vec, st);
        if (need <= 5) return launch_csm<5, 32, 128>(a, vec, st);
        return launch_csm<9, 32, 128>(a, vec, st);
    }
    // 129..256 channels: five tile slots per wave keep the kernel within 256 registers (arch + acc), so two
    // workgroups share a CU.  Nine slots take 455 and run ONE wave per SIMD -- nothing overlaps the MFMA chain's
    // operand waits and VALU work: 7.1 ms against 5.2 ms at 256 ch x 2500 observations x 513 bins.
    return launch_csm<5, 16, 256>(a, vec, st);
}

extern "C" int sc_csm_accumulate_f32(const void* d_X, const sc_spectra_desc* desc, uint32_t planes,
                                     float* d_accum, void* stream) {
    return csm_accumulate(d_X, desc, planes, SC_PLANE_CSM, d_accum, stream);
}

// U = X / |X| over a linear span of float2 (0 -> NaN, like the reference's x / abs(x))
__global__ void __launch_bounds__(256) csm_unit_normalize_kernel(const float2* __restrict__ X, float2* __restrict__ U, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float2 v = X[i];
        const float inv = rsqrtf(v.x * v.x + v.y * v.y);
        U[i] = make_float2(v.x * inv, v.y * inv);
    }
}

static int64_t csm_span(const ScAxes& ax) {
    return (int64_t)(ax.F - 1) * ax.sF + (int64_t)(ax.W - 1) * ax.sW + (int64_t)(ax.R - 1) * ax.sR +
           (int64_t)(ax.K - 1) * ax.sK + ax.C;
}

// SC_PLANE_UNIT for the shapes the one-pass kernels do not take (odd or > 128 channels): sum s/|s| is the cross-spectral
// matrix of x/|x| (sc_fused_unit_ws_f32), so a normalised copy of the spectra in d_scratch goes through the f32-MFMA
// kernel -- 27 -> 9.5 ms at 160 channels against the per-pair rsqrt of the VALU kernel.
extern "C" int64_t sc_unit_scratch_bytes(const sc_spectra_desc* desc) {
    ScAxes ax;
    if (!desc || sc_make_axes(desc, &ax) != SC_OK) return 0;
    return csm_span(ax) * (int64_t)sizeof(float2);
}

extern "C" int sc_unit_accumulate_f32(const void* d_X, const sc_spectra_desc* desc, uint32_t planes, float* d_accum,
                                      void* d_scratch, int64_t scratch_bytes, void* stream) {
    SC_REQUIRE(d_X && desc && d_accum && d_scratch, "NULL argument");
    ScAxes ax;
    sc_make_axes(desc, &ax);
    const int64_t span = csm_span(ax);
    SC_REQUIRE(scratch_bytes >= span * (int64_t)sizeof(float2), "scratch smaller than sc_unit_scratch_bytes()");
    hipLaunchKernelGGL(csm_unit_normalize_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, (const float2*)d_X,
                       (float2*)d_scratch, span);
    SC_CHECK_HIP(hipGetLastError());
    return csm_accumulate(d_scratch, desc, planes, SC_PLANE_UNIT, d_accum, stream);
}
