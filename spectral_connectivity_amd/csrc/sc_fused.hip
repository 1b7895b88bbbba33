// sc_fused.hip -- stage B in ONE pass over the spectra: the cross-spectral matrix on the
// matrix cores AND the sum |Im s| plane (wPLI weights) on the VALU, concurrently.
//
// Measured on MI355X (profiles/r01_pipe_overlap.txt): f32-input MFMA and f32 VALU do NOT
// overlap -- two waves of one SIMD issuing v_mfma_f32_16x16x4_f32 and v_fma_f32 take the SUM
// of their times (the "f32 matrix rate = f32 vector rate" of the ISA is one shared FMA pipe)
// -- while bf16 MFMA and f32 VALU do overlap.  So the Hermitian rank-n_obs update runs on the
// bf16 matrix pipe with every f32 coefficient split EXACTLY into three bf16 pieces
//     x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)   (round to nearest)
// and each product expanded into the six leading cross terms hh, hm, mh, hl, lh, mm
// (v_mfma_f32_16x16x32_bf16, products exact, f32 accumulation; the dropped ml, lm, ll terms
// are <= 2^-25 relative, below f32 rounding), leaving the f32 VALU free for the per-observation
// non-linearity  acc += |Im(x_i conj x_j)|  at 3 instructions per channel pair.
//
// Workgroup = 12 waves = one output bin (C <= 128, even).  Every SIMD hosts 1 MFMA wave and
// 2 VALU waves (a single wave issues one VALU op per ~5 cycles; two interleave to the pipe
// rate).  All waves stage a chunk of 32 observation rows HBM -> registers -> LDS twice:
//   rows   [obs][channel] float2            for the VALU waves (ds_read_b128, 4x4 pair tiles)
//   planes [6][channel][obs] bf16           for the MFMA waves (ds_read_b128 = 8 obs of one
//                                           channel = one 16x16x32 operand fragment)
// so the 6.5 GB of spectra of the headline configuration are read from HBM once for both
// products (the unfused path reads them twice).
#include "sc_stage.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define FU_OC 32            // observation rows per chunk (= K of the bf16 MFMA)
#define FU_THREADS 768
#define FU_MAXB 5
#define FU_FLUSH 16         // chunks between folds of the MFMA accumulators into the output record
#define FU_PSTRIDE 40       // bf16 elements per (plane, channel): 32 obs + 8 pad -> 80 B, conflict-free b128

struct FusedArgs {
    ScStage st;
    float* accum;
    int64_t floats_per_bin;
    int n_bins, F, NB, n_tiles, NB32, n_blocks32, n_sets;
    int csm_plane, abs_plane;
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf16lo_to_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// 4 consecutive observations of one real component -> three 8-byte groups (h, m, l) of 4 bf16
__device__ __forceinline__ void split4(const float x[4], uint2& h, uint2& m, uint2& l) {
    h.x = cvt_pk_bf16(x[0], x[1]); h.y = cvt_pk_bf16(x[2], x[3]);
    const float r0 = x[0] - bf16lo_to_f32(h.x), r1 = x[1] - bf16hi_to_f32(h.x);
    const float r2 = x[2] - bf16lo_to_f32(h.y), r3 = x[3] - bf16hi_to_f32(h.y);
    m.x = cvt_pk_bf16(r0, r1); m.y = cvt_pk_bf16(r2, r3);
    const float s0 = r0 - bf16lo_to_f32(m.x), s1 = r1 - bf16hi_to_f32(m.x);
    const float s2 = r2 - bf16lo_to_f32(m.y), s3 = r3 - bf16hi_to_f32(m.y);
    l.x = cvt_pk_bf16(s0, s1); l.y = cvt_pk_bf16(s2, s3);
}

// Staging work item = (channel pair q, observation quad oq): 64 x 8 = 512 items per chunk, one
// per thread of the 8 VALU waves (threads 256..767); the MFMA waves keep their registers for
// accumulators and fragments.
struct FuRegs { float4 v[4]; };

__device__ __forceinline__ void fu_load(const ScStage& st, int o0, int tid, FuRegs& r) {
    const int q = tid & 63, oq = (tid >> 6) - 4;
    const int c = 2 * q;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = o0 + oq * 4 + k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o < st.n_obs && c < st.C)
            v = *reinterpret_cast<const float4*>(st.base + sc_stage_obs_offset(st, o) + c);
        r.v[k] = v;
    }
}

__device__ __forceinline__ void fu_store(const ScStage& st, float* rows, unsigned short* planes, int tid,
                                         const FuRegs& r) {
    const int q = tid & 63, oq = (tid >> 6) - 4;
    if (2 * q >= st.CP) return;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4*>(rows + (oq * 4 + k) * st.RS + 4 * q) = r.v[k];
    // planes: 0 re_h 1 re_m 2 re_l 3 im_h 4 im_m 5 im_l ; element (plane, ch, obs)
    const int plane_elems = st.CP * FU_PSTRIDE;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        float re[4], im[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            re[k] = cc ? r.v[k].z : r.v[k].x;
            im[k] = cc ? r.v[k].w : r.v[k].y;
        }
        uint2 h, m, l;
        unsigned short* base = planes + (2 * q + cc) * FU_PSTRIDE + oq * 4;
        split4(re, h, m, l);
        *reinterpret_cast<uint2*>(base) = h;
        *reinterpret_cast<uint2*>(base + plane_elems) = m;
        *reinterpret_cast<uint2*>(base + 2 * plane_elems) = l;
        split4(im, h, m, l);
        *reinterpret_cast<uint2*>(base + 3 * plane_elems) = h;
        *reinterpret_cast<uint2*>(base + 4 * plane_elems) = m;
        *reinterpret_cast<uint2*>(base + 5 * plane_elems) = l;
    }
}

__device__ __forceinline__ bf16x8 neg8(bf16x8 v) {
    u32x4 u = __builtin_bit_cast(u32x4, v);
    u ^= 0x80008000u;
    return __builtin_bit_cast(bf16x8, u);
}

#define FU_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// The two roles are separate functions so their accumulators never coexist in registers.
// Both execute the same barrier sequence: per chunk 2 barriers, then 2*log2(waves per set).
template <int MAX_SLOTS>
__device__ __forceinline__ void fused_mfma_role(const FusedArgs& p, const ScStage& st, float* rows,
                                                unsigned short* planes, int tid, int wave, int bin) {
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
    const int n_chunks = (st.n_obs + FU_OC - 1) / FU_OC;
    const int plane_elems = st.CP * FU_PSTRIDE;
    // fragment address of lane (channel c = lane & 15, obs group g = lane >> 4) inside a block
    const unsigned short* frag0 = planes + (lane & 15) * FU_PSTRIDE + (lane >> 4) * 8;
    float* out = p.accum + (int64_t)bin * p.floats_per_bin + (int64_t)p.csm_plane * p.n_tiles * SC_TILE_ELEMS;
    for (int ch = 0; ch < n_chunks; ++ch) {
        __syncthreads();          // chunk ch staged by the VALU waves
#pragma unroll
        for (int s = 0; s < MAX_SLOTS; ++s) {
            // recompute the two fragment addresses per chunk (opaque SGPR copies): keeping 18
            // loop-invariant address VGPRs alive would spill at the 168-register budget
            int bis = bi[s], bjs = bj[s];
            asm volatile("" : "+s"(bis), "+s"(bjs));
            const unsigned short* fa = frag0 + bis * 16 * FU_PSTRIDE;
            const unsigned short* fb = frag0 + bjs * 16 * FU_PSTRIDE;
            // six leading terms of (h+m+l)(h+m+l): hh hm mh mm with the h and m planes, then hl lh
            // with the l planes loaded over the m registers (keeps <= 8 fragments live).
            // Re += ar*br + ai*bi ; Im += ai*br - ar*bi : two independent accumulate chains, interleaved.
#define FU_LD(ptr, k) (*reinterpret_cast<const bf16x8*>((ptr) + (k) * plane_elems))
            const bf16x8 arh = FU_LD(fa, 0), aih = FU_LD(fa, 3), brh = FU_LD(fb, 0), bih = FU_LD(fb, 3);
            const bf16x8 nrh = neg8(arh);
            {
                const bf16x8 arm = FU_LD(fa, 1), aim = FU_LD(fa, 4), brm = FU_LD(fb, 1), bimm = FU_LD(fb, 4);
                const bf16x8 nrm = neg8(arm);
                FU_MFMA(arh, brh, re[s]);  FU_MFMA(aih, brh, im[s]);
                FU_MFMA(aih, bih, re[s]);  FU_MFMA(nrh, bih, im[s]);
                FU_MFMA(arh, brm, re[s]);  FU_MFMA(aih, brm, im[s]);
                FU_MFMA(aih, bimm, re[s]); FU_MFMA(nrh, bimm, im[s]);
                FU_MFMA(arm, brh, re[s]);  FU_MFMA(aim, brh, im[s]);
                FU_MFMA(aim, bih, re[s]);  FU_MFMA(nrm, bih, im[s]);
                FU_MFMA(arm, brm, re[s]);  FU_MFMA(aim, brm, im[s]);
                FU_MFMA(aim, bimm, re[s]); FU_MFMA(nrm, bimm, im[s]);
            }
            {
                const bf16x8 arl = FU_LD(fa, 2), ail = FU_LD(fa, 5), brl = FU_LD(fb, 2), bil = FU_LD(fb, 5);
                const bf16x8 nrl = neg8(arl);
                FU_MFMA(arh, brl, re[s]);  FU_MFMA(aih, brl, im[s]);
                FU_MFMA(aih, bil, re[s]);  FU_MFMA(nrh, bil, im[s]);
                FU_MFMA(arl, brh, re[s]);  FU_MFMA(ail, brh, im[s]);
                FU_MFMA(ail, bih, re[s]);  FU_MFMA(nrl, bih, im[s]);
            }
#undef FU_LD
            __builtin_amdgcn_sched_barrier(0);   // keep the next slot's fragment loads from piling up registers
        }
        // Two-level summation: every FU_FLUSH chunks (512 observations) the f32 accumulators are
        // folded into the output record (owned by this wave, L2-resident) and cleared, so no f32
        // chain is longer than 16 chunk-sums + n_obs/512 partials: at n_obs = 7000 the power error
        // drops from 8e-6 (one 219-long chain) to < 1e-6 relative.
        if (((ch + 1) % FU_FLUSH) == 0 || ch + 1 == n_chunks) {
            const bool first = ch < FU_FLUSH;
#pragma unroll
            for (int s = 0; s < MAX_SLOTS; ++s) {
                const int t = s * 4 + wave;
                if (t < p.n_tiles) {
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
        __syncthreads();
    }
    const int wps = 8 / p.n_sets;
    for (int half = wps >> 1; half >= 1; half >>= 1) { __syncthreads(); __syncthreads(); }
}

__device__ __forceinline__ void fused_valu_role(const FusedArgs& p, const ScStage& st, float* rows,
                                                unsigned short* planes, int tid, int vw, int bin) {
    const int lane = tid & 63;
    const int wps = 8 / p.n_sets;                         // VALU waves per block set (8 or 4)
    const int set = vw / wps, rsub = vw % wps;
    int BI[FU_MAXB], BJ[FU_MAXB];
    float acc[FU_MAXB][16];
#pragma unroll
    for (int s = 0; s < FU_MAXB; ++s) {
        const int t = set * FU_MAXB + s;
        int r = 0, rem = (t < p.n_blocks32) ? t : 0, len = p.NB32;
        while (rem >= len) { rem -= len; ++r; --len; }
        BI[s] = r; BJ[s] = r + rem;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
    }
    const int n_chunks = (st.n_obs + FU_OC - 1) / FU_OC;
    const int li = lane >> 3, lj = lane & 7;
    FuRegs regs;
    fu_load(st, 0, tid, regs);
    for (int ch = 0; ch < n_chunks; ++ch) {
        fu_store(st, rows, planes, tid, regs);
        __syncthreads();
        if (ch + 1 < n_chunks) fu_load(st, (ch + 1) * FU_OC, tid, regs);
        // zero rows past n_obs contribute |0| = 0: no bound needed for this plane.
        // Operands of the NEXT (row, block) are fetched before the current block is consumed, so
        // the ~100-cycle LDS latency hides under the 48 VALU ops instead of stalling the wave.
        {
            float4 ci0, ci1, cj0, cj1;
            {
                const float* rp = rows + rsub * st.RS;
                const float4* pi = reinterpret_cast<const float4*>(rp + (BI[0] * 32 + li * 4) * 2);
                const float4* pj = reinterpret_cast<const float4*>(rp + (BJ[0] * 32 + lj * 4) * 2);
                ci0 = pi[0]; ci1 = pi[1]; cj0 = pj[0]; cj1 = pj[1];
            }
            for (int row = rsub; row < FU_OC; row += wps) {
                const float* rp = rows + row * st.RS;
                const float* rn = rows + ((row + wps < FU_OC) ? row + wps : row) * st.RS;
#pragma unroll
                for (int s = 0; s < FU_MAXB; ++s) {
                    const float* rq = (s + 1 < FU_MAXB) ? rp : rn;
                    const int sn = (s + 1 < FU_MAXB) ? s + 1 : 0;
                    const float4* pi = reinterpret_cast<const float4*>(rq + (BI[sn] * 32 + li * 4) * 2);
                    const float4* pj = reinterpret_cast<const float4*>(rq + (BJ[sn] * 32 + lj * 4) * 2);
                    const float4 ni0 = pi[0], ni1 = pi[1], nj0 = pj[0], nj1 = pj[1];
                    const float xi_re[4] = {ci0.x, ci0.z, ci1.x, ci1.z}, xi_im[4] = {ci0.y, ci0.w, ci1.y, ci1.w};
                    const float xj_re[4] = {cj0.x, cj0.z, cj1.x, cj1.z}, xj_im[4] = {cj0.y, cj0.w, cj1.y, cj1.w};
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[s][a * 4 + b] += fabsf(xi_im[a] * xj_re[b] - xi_re[a] * xj_im[b]);
                    ci0 = ni0; ci1 = ni1; cj0 = nj0; cj1 = nj1;
                }
            }
        }
        __syncthreads();
    }
    // tree-sum the row-split partials of a set through LDS (rows + planes region, 20 KB per writer)
    float* red = rows;
    for (int half = wps >> 1; half >= 1; half >>= 1) {
        if (rsub >= half && rsub < 2 * half) {
            float* dst = red + (size_t)(set * (wps >> 1) + (rsub - half)) * (FU_MAXB * 16 * 64);
#pragma unroll
            for (int s = 0; s < FU_MAXB; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[(s * 16 + e) * 64 + lane] = acc[s][e];
        }
        __syncthreads();
        if (rsub < half) {
            const float* src = red + (size_t)(set * (wps >> 1) + rsub) * (FU_MAXB * 16 * 64);
#pragma unroll
            for (int s = 0; s < FU_MAXB; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s][e] += src[(s * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (rsub == 0) {
        float* out = p.accum + (int64_t)bin * p.floats_per_bin + (int64_t)p.abs_plane * p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
        for (int s = 0; s < FU_MAXB; ++s) {
            if (set * FU_MAXB + s < p.n_blocks32) {
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

template <int MAX_SLOTS>
__global__ void __launch_bounds__(FU_THREADS) fused_csm_absim_kernel(FusedArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bin = blockIdx.x;   // one workgroup per bin: consecutive bins on consecutive XCDs
    const int g = bin / p.F, f = bin - g * p.F;
    ScStage st = p.st;
    st.base = p.st.base + (int64_t)f * st.ax.sF + sc_group_offset(st.ax, g);
    float* rows = reinterpret_cast<float*>(smem);
    unsigned short* planes = reinterpret_cast<unsigned short*>(smem + (size_t)FU_OC * st.RS * sizeof(float));
    if (wave < 4) fused_mfma_role<MAX_SLOTS>(p, st, rows, planes, tid, wave, bin);
    else fused_valu_role(p, st, rows, planes, tid, wave - 4, bin);
}

template <int MAX_SLOTS>
static int launch_fused(const FusedArgs& a, hipStream_t stream) {
    size_t shmem = (size_t)FU_OC * a.st.RS * sizeof(float) + (size_t)6 * a.st.CP * FU_PSTRIDE * 2;
    const size_t red = (size_t)4 * FU_MAXB * 16 * 64 * sizeof(float);
    if (shmem < red) shmem = red;
    auto k = fused_csm_absim_kernel<MAX_SLOTS>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3((unsigned)a.n_bins), dim3(FU_THREADS), shmem, stream, a);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

// d_X may be NULL when only the shape is known: alignment is then assumed.
static bool fused_ok(const void* d_X, const ScAxes& ax) {
    if (ax.C < 1 || ax.C > 128 || (ax.C & 1)) return false;
    if ((ax.sW | ax.sR | ax.sK | ax.sF) & 1) return false;
    return d_X == nullptr || (((uintptr_t)d_X) % 16 == 0);
}

extern "C" int sc_fused_supported(int64_t n_signals) {
    return (n_signals >= 2 && n_signals <= 128 && (n_signals % 2) == 0) ? 1 : 0;
}

extern "C" int sc_fused_csm_absim_f32(const void* d_X, const sc_spectra_desc* desc, uint32_t planes,
                                      float* d_accum, void* stream) {
    SC_REQUIRE(d_X && desc && d_accum, "NULL argument");
    SC_REQUIRE((planes & (SC_PLANE_CSM | SC_PLANE_ABS_IM)) == (SC_PLANE_CSM | SC_PLANE_ABS_IM),
               "planes must contain SC_PLANE_CSM and SC_PLANE_ABS_IM");
    ScAxes ax;
    sc_make_axes(desc, &ax);
    SC_REQUIRE(ax.C >= 1 && ax.F >= 1 && ax.n_obs >= 1 && ax.n_groups >= 1, "empty dimension");
    if (!fused_ok(d_X, ax)) {
        sc_set_error("fused CSM+|Im| kernel needs an even n_signals <= 128 and 16-byte aligned rows (got C=%d)", ax.C);
        return SC_EUNSUPPORTED;
    }
    FusedArgs a;
    a.NB = sc_n_blocks(ax.C);
    a.n_tiles = sc_n_tiles(a.NB);
    a.NB32 = (ax.C + 31) / 32;
    a.n_blocks32 = a.NB32 * (a.NB32 + 1) / 2;
    a.n_sets = (a.n_blocks32 + FU_MAXB - 1) / FU_MAXB;        // 1 or 2
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
    hipStream_t s = (hipStream_t)stream;
    const int need = (a.n_tiles + 3) / 4;
    if (need <= 1) return launch_fused<1>(a, s);
    if (need <= 3) return launch_fused<3>(a, s);
    if (need <= 5) return launch_fused<5>(a, s);
    return launch_fused<9>(a, s);
}
