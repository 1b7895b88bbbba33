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
// Workgroup = 12 waves = one output bin (C <= 128, even).  Every SIMD hosts 1 "CSM" wave and
// 2 "abs" waves.  A chunk of 32 observation rows is staged HBM -> registers -> LDS as bf16 pieces
// (h, m, l of Re and Im, exact 3-way split of every f32 coefficient) in ONE layout, double-buffered:
//   planes [2][channel][6][obs] bf16        (6 = h, m, l of Re then of Im)
// The CSM waves read it with K = observations: operand fragments of the rank-n_obs update
// S += X^H X (v_mfma_f32_16x16x32_bf16, six leading cross terms hh hm mh mm hl lh per product,
// f32 accumulate).  The abs waves read the same planes four observations at a time (one 8-byte
// read per plane and channel) and rebuild, with byte permutes, operands whose K axis is the six
// cross terms of ONE observation: a single v_mfma_f32_32x32x16_bf16 with C = 0 returns the per-
// observation Im(x_i conj x_j) of a 32x32 channel block at f32 accuracy, and the VALU only does
// acc += |d| (16 instructions per 1024 pairs instead of 3 per pair).
// The abs waves also do the staging (VALU only): chunk n+1 is split and written to the other
// buffer while the CSM waves already run the MFMAs of chunk n, one barrier per chunk.
// The non-linearity (|.| before the expectation) is what keeps this from being a GEMM; moving its
// products onto the idle bf16 matrix pipe cut the VALU work 2.5x.  The spectra are read from HBM
// once for both products.
#include <stdlib.h>
#include "sc_stage.h"
#include "sc_fused_common.h"

// Optional phase timers (tools/fused_trace.py builds a copy of the library with -DFU_TRACE): per wave of
// workgroup 0, shader-clock cycles summed over the chunks for up to 4 phases.
#ifdef FU_TRACE
__device__ unsigned long long fu_trace_buf[12 * 4];
#define FU_T0() unsigned long long fu_t = __builtin_readcyclecounter()
#define FU_TICK(slot)                                                                  \
    do {                                                                               \
        const unsigned long long fu_n = __builtin_readcyclecounter();                  \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) fu_trace_buf[(threadIdx.x >> 6) * 4 + (slot)] += fu_n - fu_t; \
        fu_t = fu_n;                                                                   \
    } while (0)
extern "C" int sc_debug_fused_trace(unsigned long long* out, int reset) {
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fu_trace_buf), sizeof(unsigned long long) * 48);
    if (reset) { unsigned long long z[48] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(fu_trace_buf), z, sizeof z); }
    return 0;
}
#else
#define FU_T0() do {} while (0)
#define FU_TICK(slot) do {} while (0)
#endif

#define FU_PLANE 36         // bf16 per (channel, plane): 32 obs + 4 pad = 72 B
#define FU_CSTRIDE 220      // bf16 per channel: its 6 planes (432 B) + 8 B pad = 440 B = 110 dwords.  110 = 14
                            // (mod 32): 16 consecutive channels at one obs quad hit 16 distinct even banks, so
                            // the 8-byte reads of both roles and the staging writes (each 16-lane group mixes
                            // the even and odd channel of its pairs) are conflict-free; and all six planes
                            // of a channel sit within the 8-bit offsets of ds_read2_b64 -- one address add
                            // per 16-channel fragment set instead of one per plane on this VALU-bound kernel

// Staging (done by the CSM waves, see fused_mfma_role).  A staging wave owns eight observation rows of
// every chunk (two quads vw): it pulls them from HBM straight into an f32 LDS buffer (global_load_lds_dwordx4: lane l of the wave lands at base + 16 l,
// i.e. one 1 KB row = 64 channel pairs per instruction, no VGPRs held while the loads are in flight),
// later reads its own rows back, splits them and writes the bf16 planes.  Because nobody else touches
// those raw rows, re-filling them needs no barrier.
#define FU_NPLANES 6
#define FU_RAW_ROW 256      // floats per raw row: 64 slots x (Re, Im) x 2 channels
__device__ __forceinline__ void fu_fetch(const ScStage& st, const FuMap& mp, float* raw, int o0, int vw) {
    const int lane = fu_lane();
    const int c = 2 * (lane & 15), b32 = lane >> 4;                     // channel pair c of local block b32
    const bool have = c < fu_byte(mp.n32, b32);                         // this lane's channel pair exists
    const int goff = fu_byte(mp.off32, b32) * 32 + c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = 4 * vw + k, o = o0 + row;
        float* dst = raw + row * FU_RAW_ROW;             // wave-uniform
        if (o < st.n_obs) {
            if (have)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(st.base + sc_stage_obs_offset(st, o) + goff),
                    (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else {
            *reinterpret_cast<float4*>(dst + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);   // rows past n_obs
        }
    }
}

// lane l stages channel pair l of the wave's four rows.  Half of each 16-lane group takes the even
// channel of its pair first and the other half the odd one: the 8-byte raw reads and the 8-byte plane
// writes (channel stride 18 dwords) then touch every LDS bank exactly once.
// UNIT: the rows are normalised to unit phasors x / |x| on the way (phase_locking_value, pairwise_phase_consistency:
// the sum of s / |s| is the cross-spectral matrix of the unit phasors) -- rsq + two multiplies per coefficient on the
// staging waves instead of a normalised copy of the spectra in HBM.  0 / 0 is NaN like the reference's; zero-filled rows
// past n_obs (rows_valid) and slots of absent channels stay zero.
template <int NB32, bool UNIT = false>
__device__ __forceinline__ void fu_split(const float* raw, unsigned short* planes, int vw, const FuMap* mp = nullptr,
                                         int rows_valid = FU_OC) {
    const int lane = fu_lane();
    constexpr int CP = NB32 * 32;                  // channels staged (C rounded up to 32)
    if (2 * lane >= CP) return;
    // planes: 0 re_h 1 re_m 2 re_l 3 im_h 4 im_m 5 im_l ; element (plane, ch, obs)
    constexpr int plane_elems = FU_PLANE;          // plane stride inside a channel
    const int first = (lane >> 3) & 1;
    float2 v[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[t][k] = *reinterpret_cast<const float2*>(raw + (4 * vw + k) * FU_RAW_ROW + 4 * lane + 2 * (first ^ t));
    if constexpr (UNIT) {
        // (both channels of a pair exist or neither: even counts)
        const bool have = 2 * (lane & 15) < fu_byte(mp->n32, lane >> 4);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (have && 4 * vw + k < rows_valid) {
                    const float ia = rsqrtf(v[t][k].x * v[t][k].x + v[t][k].y * v[t][k].y);
                    v[t][k].x *= ia; v[t][k].y *= ia;
                }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float re[4], im[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { re[k] = v[t][k].x; im[k] = v[t][k].y; }
        unsigned short* base = planes + (2 * lane + (first ^ t)) * FU_CSTRIDE + vw * 4;
        uint2 h, m, l;
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

// Prologue of a staging wave (quad vw): clear its raw rows (slots of absent channels stay zero for
// good), fetch and stage chunk 0, put chunk 1 in flight.
template <int NB32, bool UNIT = false>
__device__ __forceinline__ void fu_stage_first(const ScStage& st, const FuMap& mp, float* raw, unsigned short* planes, int vw,
                                               int o_lo, int n_chunks, bool loads) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4*>(raw + (4 * vw + k) * FU_RAW_ROW + 4 * fu_lane()) = make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fu_fetch(st, mp, raw, o_lo, vw);   // (debug "no HBM loads" keeps re-using this chunk: realistic operand values)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    fu_split<NB32, UNIT>(raw, planes, vw, &mp, st.n_obs - o_lo);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (n_chunks > 1 && loads) fu_fetch(st, mp, raw, o_lo + FU_OC, vw);
}

// Stage chunk ch + 1 into the other plane buffer, then put the loads of chunk ch + 2 in flight.
template <int NB32, bool UNIT = false>
__device__ __forceinline__ void fu_stage_next(const ScStage& st, const FuMap& mp, float* raw, unsigned short* planes, int vw,
                                              int o_lo, int ch, int n_chunks, bool loads) {
    constexpr int buf_elems = NB32 * 32 * FU_CSTRIDE;
    if (ch + 1 < n_chunks) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // raw rows of chunk ch + 1 landed
        fu_split<NB32, UNIT>(raw, planes + ((ch + 1) & 1) * buf_elems, vw, &mp, st.n_obs - (o_lo + (ch + 1) * FU_OC));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // raw rows read before the refill
        if (ch + 2 < n_chunks && loads) fu_fetch(st, mp, raw, o_lo + (ch + 2) * FU_OC, vw);
    }
}

__device__ __forceinline__ unsigned perm_b32(unsigned a, unsigned b, unsigned sel) {
    return __builtin_amdgcn_perm(a, b, sel);     // bytes 0-3 of sel pick from b, 4-7 from a
}

// sign flip of eight bf16: one v_xor_b32 per dword (spelled in asm: through the builtin vector types the
// compiler legalises the xor per 16-bit half -- xor, sdwa xor and a permute for every dword)
__device__ __forceinline__ bf16x8 neg8(bf16x8 v) {
    const u32x4 u = __builtin_bit_cast(u32x4, v);
    const unsigned m = 0x80008000u;
    unsigned a, b, c, d;
    asm("v_xor_b32 %0, %1, %2" : "=v"(a) : "v"(u[0]), "v"(m));
    asm("v_xor_b32 %0, %1, %2" : "=v"(b) : "v"(u[1]), "v"(m));
    asm("v_xor_b32 %0, %1, %2" : "=v"(c) : "v"(u[2]), "v"(m));
    asm("v_xor_b32 %0, %1, %2" : "=v"(d) : "v"(u[3]), "v"(m));
    const u32x4 r = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, r);
}

// eight consecutive observations of one (plane, channel): 72-byte channel stride -> two 8-byte reads
__device__ __forceinline__ bf16x8 fu_ld8(const unsigned short* ptr) {
    const uint2 a = *reinterpret_cast<const uint2*>(ptr), b = *reinterpret_cast<const uint2*>(ptr + 4);
    const u32x4 u = {a.x, a.y, b.x, b.y};
    return __builtin_bit_cast(bf16x8, u);
}

// Workgroup barrier that publishes LDS writes only.  __syncthreads() would also drain vmcnt, i.e. make
// every wave sit out the HBM->LDS row loads that were just put in flight for a later chunk.
#define FU_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#define FU_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

static_assert(2 * 4 + 1 <= FU_FLUSH, "one fold slot per tile of a wave");
// The two roles are separate functions so their accumulators never coexist in registers.
// Both execute the same barrier sequence: one before the first chunk, one per chunk, then
// 2*log2(waves per set).
//
// MFMA role.  Wave w owns tile rows w and NB-1-w of the upper triangle (NB+1 tiles for even NB):
// the six A fragments of a row (Re, Im x h,m,l) are loaded once per row and chunk, the six B
// fragments per tile, and the first two B fragments of the NEXT tile are prefetched under the 24
// MFMAs of the current one.
template <int NB32>
__device__ __forceinline__ void fused_mfma_role(const FusedArgs& p, const ScStage& st,
                                                unsigned short* planes, float* raw, int tid, int wave,
                                                float* rec, int o_lo) {
    constexpr int MAXS = 2 * NB32 + 1;
    const int lane = tid & 63;
    const int NB = p.NB;
    // this wave's tiles: nA in tile row rA from column cA, then nB in row rB from column cB (fu_assign_rows: rows w and
    // R-1-w of the launch's R tile rows -- the whole triangle and the staircases of the > 128-channel launches alike)
    // (one byte per field, one word per wave; scalar selects on the wave number)
    const unsigned sg = (unsigned)__builtin_amdgcn_readfirstlane((int)(wave == 0 ? p.seg0 : (wave == 1 ? p.seg1 : (wave == 2 ? p.seg2 : p.seg3))));
    const unsigned sg_n = (unsigned)__builtin_amdgcn_readfirstlane((int)((p.seg_n >> (8 * wave)) & 0xffu));
    const int rA_ = sg & 0xf, cA_ = (sg >> 4) & 0xf, rB_ = (sg >> 8) & 0xf, cB_ = (sg >> 12) & 0xf;
    const int nA_ = sg_n & 0xf, nB_ = sg_n >> 4;
    const int total = nA_ + nB_;
    (void)NB;
    f32x4 re[MAXS], im[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) { re[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; im[s] = re[s]; }
    const int n_chunks = (st.n_obs - o_lo + FU_OC - 1) / FU_OC;     // st.n_obs: end of this part
    constexpr int plane_elems = FU_PLANE, buf_elems = NB32 * 32 * FU_CSTRIDE;
    const unsigned short* frag00 = planes + (lane & 15) * FU_CSTRIDE + (lane >> 4) * 8;
    float* out = rec + (int64_t)p.csm_plane * p.n_tiles * SC_TILE_ELEMS;
#define FU_LD(ptr, k) fu_ld8((ptr) + (k) * plane_elems)
    // Staging is shared: the four CSM waves (VALU idle under their MFMA stream) stage observation
    // quads 0-3 AFTER their products, abs waves 0-3 stage quads 4-7 BEFORE theirs.
    const bool loads = !(p.debug_skip & 8);
    const bool csm_stages = p.abs_plane >= 0;       // CSM only: the eight other waves have nothing else to do
    const bool do_csm = p.csm_plane >= 0 && (p.debug_skip & 1) == 0;      // plane passes: staging only
    if (csm_stages) fu_stage_first<NB32>(st, p.map, raw, planes, wave, o_lo, n_chunks, loads);
    FU_BARRIER();                 // chunk 0 staged
    FU_T0();
    for (int ch = 0; ch < n_chunks; ++ch) {
        const unsigned short* frag0 = frag00 + (ch & 1) * buf_elems;
        if (do_csm && total > 0) {
            // opaque per-chunk copies: otherwise ~2 loop-invariant address VGPRs per tile stay live
            // across the chunk loop and spill at the 168-register budget
            int rA = rA_, rB = rB_, nA = nA_, cA = cA_, cB = cB_;
            asm volatile("" : "+s"(rA), "+s"(rB), "+s"(nA), "+s"(cA), "+s"(cB));
            bf16x8 arh, arm, arl, aih, aim, ail, nrh, nrm, nrl;     // A fragments of the current row (and -Re)
            bf16x8 brh[2], bih[2];                                  // first B fragments, prefetched one tile
            {                                                       // ahead into the other register set
                const unsigned short* fb = frag0 + cA * 16 * FU_CSTRIDE;   // first tile (rA, cA)
                brh[0] = FU_LD(fb, 0); bih[0] = FU_LD(fb, 3);
            }
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                if (s < total) {
                    const bool in_a = s < nA;
                    const int row = in_a ? rA : rB;
                    const int col = in_a ? cA + s : cB + (s - nA);
                    if (s == 0 || s == nA) {
                        const unsigned short* fa = frag0 + row * 16 * FU_CSTRIDE;
                        arh = FU_LD(fa, 0); arm = FU_LD(fa, 1); arl = FU_LD(fa, 2);
                        aih = FU_LD(fa, 3); aim = FU_LD(fa, 4); ail = FU_LD(fa, 5);
                        nrh = neg8(arh); nrm = neg8(arm); nrl = neg8(arl);
                    }
                    const unsigned short* fb = frag0 + col * 16 * FU_CSTRIDE;
                    const bf16x8 cbrm = FU_LD(fb, 1), cbim = FU_LD(fb, 4), cbrl = FU_LD(fb, 2), cbil = FU_LD(fb, 5);
                    const bf16x8& cbrh = brh[s & 1];
                    const bf16x8& cbih = bih[s & 1];
                    if (s + 1 < total) {      // prefetch the next tile's first fragments
                        const bool na = (s + 1) < nA;
                        const int ncol = na ? cA + s + 1 : cB + (s + 1 - nA);
                        const unsigned short* fn = frag0 + ncol * 16 * FU_CSTRIDE;
                        brh[(s + 1) & 1] = FU_LD(fn, 0); bih[(s + 1) & 1] = FU_LD(fn, 3);
                    }
                    // six leading terms of (h+m+l)(h+m+l): hh hm mh mm hl lh
                    // Re += ar*br + ai*bi ; Im += ai*br + (-ar)*bi   (two chains, interleaved; the sign flip
                    // of the row's Re fragments costs 12 VALU instructions per row and chunk)
                    FU_MFMA(arh, cbrh, re[s]);  FU_MFMA(aih, cbrh, im[s]);
                    FU_MFMA(aih, cbih, re[s]);  FU_MFMA(nrh, cbih, im[s]);
                    FU_MFMA(arh, cbrm, re[s]);  FU_MFMA(aih, cbrm, im[s]);
                    FU_MFMA(aih, cbim, re[s]);  FU_MFMA(nrh, cbim, im[s]);
                    FU_MFMA(arm, cbrh, re[s]);  FU_MFMA(aim, cbrh, im[s]);
                    FU_MFMA(aim, cbih, re[s]);  FU_MFMA(nrm, cbih, im[s]);
                    FU_MFMA(arm, cbrm, re[s]);  FU_MFMA(aim, cbrm, im[s]);
                    FU_MFMA(aim, cbim, re[s]);  FU_MFMA(nrm, cbim, im[s]);
                    FU_MFMA(arh, cbrl, re[s]);  FU_MFMA(aih, cbrl, im[s]);
                    FU_MFMA(aih, cbil, re[s]);  FU_MFMA(nrh, cbil, im[s]);
                    FU_MFMA(arl, cbrh, re[s]);  FU_MFMA(ail, cbrh, im[s]);
                    FU_MFMA(ail, cbih, re[s]);  FU_MFMA(nrl, cbih, im[s]);
                }
            }
        }
        FU_TICK(1);
        // stage chunk ch + 1 into the other buffer, then put the loads of chunk ch + 2 in flight
        if (csm_stages) fu_stage_next<NB32>(st, p.map, raw, planes, wave, o_lo, ch, n_chunks, loads);
        FU_TICK(0);
        // Two-level summation: every FU_FLUSH chunks (512 observations) the f32 accumulators of a tile
        // are folded into the output record (owned by this wave, L2-resident) and cleared, so no f32
        // chain is longer than 16 chunk-sums + n_obs/512 partials: at n_obs = 7000 the power error
        // drops from 8e-6 (one 219-long chain) to < 1e-6 relative.  The tiles take turns (tile s folds
        // when ch + 1 + s is a multiple of FU_FLUSH) so that no chunk carries all the folds.
        {
            const bool last = ch + 1 == n_chunks;
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                const int f_s = FU_FLUSH - 1 - s;              // chunk of this tile's first scheduled fold
                const bool due = ((ch + 1 + s) % FU_FLUSH) == 0;
                if (s < total && (due || last) && p.csm_plane >= 0) {
                    const bool first = ch <= f_s;
                    const bool in_a = s < nA_;
                    const int row = in_a ? rA_ : rB_;
                    const int col = in_a ? cA_ + s : cB_ + (s - nA_);
                    // uniform (scalar) tile base + ONE 32-bit unsigned per-lane offset, re-materialised here:
                    // SGPR-base addressing, no 64-bit address VGPRs kept alive (and spilled) across the chunk
                    // loop -- a spill reload in front of these stores would wait on vmcnt, i.e. on the row
                    // loads that were just put in flight
                    float* o_re = out + (int64_t)sc_tile_index(fu_gt(p.map, row), fu_gt(p.map, col), p.map.NBr) * SC_TILE_ELEMS;
                    float* o_im = o_re + (int64_t)p.n_tiles * SC_TILE_ELEMS;
                    const unsigned fl = (unsigned)fu_lane();
                    const unsigned base_idx = (fl >> 4) * 64u + (fl & 15u);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned idx = base_idx + 16u * r;
                        // later folds are fire-and-forget L2 atomics issued by the record's only writer
                        // (same order every run): no global round trip inside the chunk loop
                        if (first) {
                            o_re[idx] = re[s][r];
                            o_im[idx] = im[s][r];
                        } else {
                            unsafeAtomicAdd(o_re + idx, re[s][r]);
                            unsafeAtomicAdd(o_im + idx, im[s][r]);
                        }
                    }
                    re[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; im[s] = re[s];
                }
            }
        }
        FU_TICK(2);
        FU_BARRIER();             // chunk ch consumed by both roles, chunk ch + 1 staged
        FU_TICK(3);
    }
#undef FU_LD
    const int wps = 8 / p.n_sets;
    for (int half = wps >> 1; half >= 1; half >>= 1) { __syncthreads(); __syncthreads(); }
}

// "abs" role: per observation row and 32x32 channel block ONE v_mfma_f32_32x32x16_bf16 with C = 0
// yields d = Im(x_i conj x_j) for the 1024 pairs of the block (K = 16 slots: lanes 0-31 carry the
// six cross terms of Im(x_i) Re(x_j), lanes 32-63 those of Re(x_i) (-Im(x_j))), then acc += |d|.
//   A operand (row channel i):  P(v) = [h h | m h | l m | 0 0]      v = Im x_i (lanes 0-31) / Re x_i
//   B operand (col channel j):  Q(v) = [h m | h l | h m | 0 0]      v = Re x_j (lanes 0-31) / -Im x_j
// slot products: h.h  h.m  m.h  h.l  l.h  m.m  (the six leading terms of (h+m+l)(h+m+l)).
struct FuFragA { unsigned d0, d1, d2; };
struct FuFragB { unsigned d0, d1; };
// h, m, l: the dwords of the three planes that hold observation row k of this lane's channel
// (two bf16 per dword: k even -> low half, k odd -> high half)
template <int ODD>
__device__ __forceinline__ FuFragA fu_frag_a(unsigned h, unsigned m, unsigned l) {
    constexpr unsigned SEL = ODD ? 0x07060302u : 0x05040100u;    // (lo, hi) = (b.half, a.half)
    FuFragA f;
    f.d0 = perm_b32(h, h, SEL);        // (h, h)
    f.d1 = perm_b32(h, m, SEL);        // (m, h)
    f.d2 = perm_b32(m, l, SEL);        // (l, m)
    return f;
}
template <int ODD>
__device__ __forceinline__ FuFragB fu_frag_b(unsigned h, unsigned m, unsigned l) {
    constexpr unsigned SEL = ODD ? 0x07060302u : 0x05040100u;
    FuFragB f;
    f.d0 = perm_b32(m, h, SEL);       // (h, m)
    f.d1 = perm_b32(l, h, SEL);       // (h, l)
    return f;
}

template <int NB32, int COL_LO, int ROW_HI, int SET, int OP>
__device__ __forceinline__ void fused_valu_body(const FusedArgs& p, const ScStage& st, unsigned short* planes,
                                                float* raw, int tid, int vw, int rsub, int wps, float* rec,
                                                int o_lo) {
    using Tab = FuTab<NB32, COL_LO, ROW_HI, SET>;
    constexpr int NBLK = Tab::NBLK;
    const int lane = tid & 63;
    f32x16 acc[NBLK > 0 ? NBLK : 1];
#pragma unroll
    for (int s = 0; s < NBLK; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
    const int n_chunks = (st.n_obs - o_lo + FU_OC - 1) / FU_OC;
    constexpr int plane_elems = FU_PLANE, buf_elems = NB32 * 32 * FU_CSTRIDE;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool loads = !(p.debug_skip & 8);
    // with the |Im| plane: abs waves 0-3 stage quads 4-7 (the CSM waves take 0-3); CSM only: all eight stage
    const bool all_stage = p.abs_plane < 0;
    const int my_quad = all_stage ? vw : 4 + vw;
    constexpr bool UNIT = OP == FU_OP_UNIT;
    if (vw < 4 || all_stage) fu_stage_first<NB32, UNIT>(st, p.map, raw, planes, my_quad, o_lo, n_chunks, loads);
    FU_BARRIER();                 // chunk 0 staged
    FU_T0();
    for (int ch = 0; ch < n_chunks; ++ch) {
        if (vw < 4 || all_stage) fu_stage_next<NB32, UNIT>(st, p.map, raw, planes, my_quad, o_lo, ch, n_chunks, loads);
        FU_TICK(0);
        const unsigned short* pb = planes + (ch & 1) * buf_elems;
        // per-lane plane triples: A reads Im (lanes 0-31) / Re (32-63), B reads Re (lanes 0-31) / Im (32-63)
        const int cl = fu_lane(), ci32 = cl & 31, chf = cl >> 5;
        const unsigned negmask = chf ? 0x80008000u : 0u;
        const int offA = (chf ? 0 : 3 * plane_elems) + ci32 * FU_CSTRIDE;
        const int offB = (chf ? 3 * plane_elems : 0) + ci32 * FU_CSTRIDE;
        // zero rows past n_obs contribute |0| = 0: no bound needed for this plane
        for (int oq2 = (((p.debug_skip & 2) || p.abs_plane < 0) ? 16 : 2 * rsub); oq2 < 16; oq2 += ((oq2 & 1) ? 2 * wps - 1 : 1)) {
            // two observation rows (one dword per plane) of this lane's channels in every needed block
            unsigned NA[NB32][3], NBq[NB32][3];
#pragma unroll
            for (int b = 0; b < NB32; ++b)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (Tab::tab.use_i[b])
                        NA[b][pl] = *reinterpret_cast<const unsigned*>(
                            pb + offA + pl * plane_elems + b * 32 * FU_CSTRIDE + oq2 * 2);
                    // (-Im x_j for lanes 32-63: the sign is flipped here, once per dword of two rows -- 3 to 6 xors per
                    // row -- and not on the fragments built from it, 2 per block and row)
                    if (Tab::tab.use_j[b])
                        NBq[b][pl] = *reinterpret_cast<const unsigned*>(
                            pb + offB + pl * plane_elems + b * 32 * FU_CSTRIDE + oq2 * 2) ^ negmask;
                }
#pragma unroll
            for (int k1 = 0; k1 < 2; ++k1) {
                // operand fragments are built one block ahead of the MFMA that uses them
                auto frag_a = [&](int b) {
                    return k1 ? fu_frag_a<1>(NA[b][0], NA[b][1], NA[b][2]) : fu_frag_a<0>(NA[b][0], NA[b][1], NA[b][2]);
                };
                auto frag_b = [&](int b) {
                    return k1 ? fu_frag_b<1>(NBq[b][0], NBq[b][1], NBq[b][2]) : fu_frag_b<0>(NBq[b][0], NBq[b][1], NBq[b][2]);
                };
                FuFragA fa = frag_a(Tab::tab.bi[0]);
                FuFragB fb = frag_b(Tab::tab.bj[0]);
                f32x16 dprev;
#pragma unroll
                for (int s = 0; s < NBLK; ++s) {
                    const u32x4 ua = {fa.d0, fa.d1, fa.d2, 0u};
                    const u32x4 ub = {fb.d0, fb.d1, fb.d0, 0u};
                    const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), zero, 0, 0, 0);
                    // keep the software pipeline: the |d| accumulation of block s-1 must be issued AFTER
                    // the MFMA of block s (otherwise hipcc sinks each MFMA next to its consumer and the
                    // wave sits out the 64-cycle MFMA latency 40 times per chunk)
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < NBLK) {
                        if (Tab::tab.bi[s + 1] != Tab::tab.bi[s]) fa = frag_a(Tab::tab.bi[s + 1]);
                        if (Tab::tab.bj[s + 1] != Tab::tab.bj[s]) fb = frag_b(Tab::tab.bj[s + 1]);
                    }
                    if (s > 0) {
                        fu_accumulate16<OP, (NBLK <= 4)>(acc[s - 1], dprev);
                    }
                    dprev = d;
                }
                __builtin_amdgcn_sched_barrier(0);
                fu_accumulate16<OP, (NBLK <= 4)>(acc[NBLK - 1], dprev);
            }
        }
        FU_TICK(1);
        FU_BARRIER();             // chunk ch consumed by both roles, chunk ch + 1 staged
        FU_TICK(3);
    }
    if constexpr (OP == FU_OP_SIGN) {          // integer sums -> float (exact: |sum| <= n_obs < 2^24)
#pragma unroll
        for (int s = 0; s < NBLK; ++s)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][e] = (float)__float_as_int(acc[s][e]);
    }
    // tree-sum the row-split partials of a set through LDS (planes region, 20 KB per writer)
    float* red = reinterpret_cast<float*>(planes);
    for (int half = wps >> 1; half >= 1; half >>= 1) {
        if (rsub >= half && rsub < 2 * half) {
            float* dst = red + (size_t)(SET * (wps >> 1) + (rsub - half)) * (FU_MAXB * 16 * 64);
#pragma unroll
            for (int s = 0; s < NBLK; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[(s * 16 + e) * 64 + lane] = acc[s][e];
        }
        __syncthreads();
        if (rsub < half) {
            const float* src = red + (size_t)(SET * (wps >> 1) + rsub) * (FU_MAXB * 16 * 64);
#pragma unroll
            for (int s = 0; s < NBLK; ++s)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s][e] += src[(s * 16 + e) * 64 + lane];
        }
        __syncthreads();
    }
    if (rsub == 0 && p.abs_plane >= 0) {
        const int i32 = lane & 31, hf = lane >> 5;
        // D layout of v_mfma_f32_32x32x16_bf16: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        float* out = rec + (int64_t)p.abs_plane * p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
        for (int s = 0; s < NBLK; ++s) {
            const int BIs = Tab::tab.bi[s], BJs = Tab::tab.bj[s];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = BIs * 32 + (e & 3) + 8 * (e >> 2) + 4 * hf, j = BJs * 32 + i32;
                const int ti = i >> 4, tj = j >> 4;
                if (ti <= tj && fu_tile_ok(p.map, ti) && fu_tile_ok(p.map, tj))
                    out[(int64_t)sc_tile_index(fu_gt(p.map, ti), fu_gt(p.map, tj), p.map.NBr) * SC_TILE_ELEMS + (i & 15) * 16 +
                        (j & 15)] = acc[s][e];
            }
        }
    }
}

template <int NB32, int COL_LO, int ROW_HI, int OP>
__device__ __forceinline__ void fused_valu_role(const FusedArgs& p, const ScStage& st,
                                                unsigned short* planes, float* raw, int tid, int vw, float* rec,
                                                int o_lo) {
    constexpr int NSETS = fu_nsets(NB32, COL_LO, ROW_HI);
    constexpr int wps = 8 / NSETS;                        // VALU waves per block set (8 or 4)
    const int set = vw / wps, rsub = vw % wps;
    if constexpr (NSETS == 1) {
        fused_valu_body<NB32, COL_LO, ROW_HI, 0, OP>(p, st, planes, raw, tid, vw, rsub, wps, rec, o_lo);
    } else {
        if (set == 0) fused_valu_body<NB32, COL_LO, ROW_HI, 0, OP>(p, st, planes, raw, tid, vw, rsub, wps, rec, o_lo);
        else fused_valu_body<NB32, COL_LO, ROW_HI, 1, OP>(p, st, planes, raw, tid, vw, rsub, wps, rec, o_lo);
    }
}

// NB32 staged 32-channel blocks; the launch's products are the blocks (bi <= bj, bj >= COL_LO, bi < ROW_HI) of them
template <int NB32, int COL_LO, int ROW_HI, int OP>
__global__ void __launch_bounds__(FU_THREADS) fused_csm_absim_kernel(FusedArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // n_split workgroups per bin (consecutive workgroups land on consecutive XCDs); part k sums its
    // share of the observation chunks into its own record: part 0 into the caller's, the others into
    // the workspace, folded in afterwards by fused_combine_kernel in a fixed order.
    const int bin = blockIdx.x / p.n_split, part = blockIdx.x - bin * p.n_split;
    const int g = bin / p.F, f = bin - g * p.F;
    ScStage st = p.st;
    st.base = p.st.base + (int64_t)f * st.ax.sF + sc_group_offset(st.ax, g);
    const int nc = (p.st.n_obs + FU_OC - 1) / FU_OC;
    const int o_lo = (int)((int64_t)part * nc / p.n_split) * FU_OC;
    const int o_hi = (int)((int64_t)(part + 1) * nc / p.n_split) * FU_OC;
    st.n_obs = o_hi < p.st.n_obs ? o_hi : p.st.n_obs;
    float* rec = (part == 0 ? p.accum : p.ws + (int64_t)(part - 1) * p.n_bins * p.floats_per_bin) +
                 (int64_t)bin * p.floats_per_bin;
    // LDS: the two plane buffers at offset 0 (also the scratch of the final tree reduction), then the f32
    // landing rows of the direct HBM->LDS loads
    unsigned short* planes = reinterpret_cast<unsigned short*>(smem);
    constexpr size_t plane_bytes = (size_t)2 * NB32 * 32 * FU_CSTRIDE * 2;
    constexpr size_t red_bytes = (size_t)4 * FU_MAXB * 16 * 64 * sizeof(float);
    float* raw = reinterpret_cast<float*>(smem + (plane_bytes > red_bytes ? plane_bytes : red_bytes));
    if (wave < 4) {
        if (p.debug_skip & 32) __builtin_amdgcn_s_setprio(2);
        fused_mfma_role<NB32>(p, st, planes, raw, tid, wave, rec, o_lo);
    } else {
        if (p.debug_skip & 16) __builtin_amdgcn_s_setprio(2);
        fused_valu_role<NB32, COL_LO, ROW_HI, OP>(p, st, planes, raw, tid, wave - 4, rec, o_lo);
    }
}

// accum[bin][plane] += ws[0][bin][plane] + ws[1][bin][plane] + ... for the CSM (re, im) and (if present) |Im| planes
__global__ void __launch_bounds__(256) fused_combine_kernel(FusedArgs p) {
    const int64_t plane = (int64_t)p.n_tiles * SC_TILE_ELEMS;      // floats per plane (multiple of 256)
    const int64_t per_bin = (p.abs_plane >= 0 ? 3 : 2) * plane / 4;   // float4 items per bin
    const int64_t total = per_bin * p.n_bins;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t bin = i / per_bin, e = (i - bin * per_bin) * 4;
        const int64_t off = bin * p.floats_per_bin +
                            (e < 2 * plane ? (int64_t)p.csm_plane * plane + e : (int64_t)p.abs_plane * plane + (e - 2 * plane));
        float4 a = *reinterpret_cast<const float4*>(p.accum + off);
        for (int k = 0; k + 1 < p.n_split; ++k) {
            const float4 b = *reinterpret_cast<const float4*>(p.ws + (int64_t)k * p.n_bins * p.floats_per_bin + off);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *reinterpret_cast<float4*>(p.accum + off) = a;
    }
}

// the same fold for an arbitrary list of planes (small-channel kernel)
__global__ void __launch_bounds__(256) planes_combine_kernel(FusedArgs p) {
    const int64_t plane = (int64_t)p.n_tiles * SC_TILE_ELEMS;
    const int64_t per_bin = p.n_fold * plane / 4;                  // float4 items per bin
    const int64_t total = per_bin * p.n_bins;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t bin = i / per_bin, e = (i - bin * per_bin) * 4;
        const int which = (int)(e / plane);
        const int64_t off = bin * p.floats_per_bin + (int64_t)p.fold[which] * plane + (e - which * plane);
        float4 a = *reinterpret_cast<const float4*>(p.accum + off);
        for (int k = 0; k + 1 < p.n_split; ++k) {
            const float4 b = *reinterpret_cast<const float4*>(p.ws + (int64_t)k * p.n_bins * p.floats_per_bin + off);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *reinterpret_cast<float4*>(p.accum + off) = a;
    }
}

int sc_internal_fused_combine(const FusedArgs& a, int op, hipStream_t stream) {
    if (a.n_split > 1) {
        if (op == FU_OP_ABS || op == FU_OP_UNIT) hipLaunchKernelGGL(fused_combine_kernel, dim3(2048), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(planes_combine_kernel, dim3(2048), dim3(256), 0, stream, a);     // one plane: a.fold
        SC_CHECK_HIP(hipGetLastError());
    }
    return SC_OK;
}

template <int NB32, int COL_LO, int ROW_HI, int OP>
static int launch_fused_op(const FusedArgs& a, bool combine, hipStream_t stream) {
    size_t shmem = (size_t)2 * a.st.CP * FU_CSTRIDE * 2;
    const size_t red = (size_t)4 * FU_MAXB * 16 * 64 * sizeof(float);
    if (shmem < red) shmem = red;
    shmem += (size_t)FU_OC * FU_RAW_ROW * sizeof(float);
    auto k = fused_csm_absim_kernel<NB32, COL_LO, ROW_HI, OP>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3((unsigned)(a.n_bins * a.n_split)), dim3(FU_THREADS), shmem, stream, a);
    SC_CHECK_HIP(hipGetLastError());
    return combine ? sc_internal_fused_combine(a, OP, stream) : SC_OK;
}

// One pass of the matrix-core kernel: op = FU_OP_ABS is the headline launch (CSM planes, and |Im s| if a.abs_plane >= 0);
// FU_OP_SQ / FU_OP_SIGN are plane passes (a.csm_plane = -1, a.abs_plane = the plane to fill): the abs waves accumulate
// d^2 / sign(d) of the same per-observation matrix-core products, the CSM waves only stage.
// The launch shapes that exist (staged blocks, first block column, block rows): the triangles of 1 ... 4 blocks for up
// to 128 channels, and the staircases launch_fused_all covers 129 ... 256 channels with.
#define FU_SHAPES(X) X(1, 0, 1) X(2, 0, 2) X(3, 0, 3) X(4, 0, 4) X(4, 2, 2) X(3, 1, 3) X(4, 2, 4) X(4, 1, 4) X(4, 1, 1)
static int launch_fused(const FusedArgs& a, int op, hipStream_t stream, bool combine) {
    const int shape = a.NB32 * 100 + a.shape_col_lo * 10 + a.shape_row_hi;
#define FU_CASE(NB32, COL_LO, ROW_HI)                                                                          \
    case NB32 * 100 + COL_LO * 10 + ROW_HI:                                                                    \
        if (op == FU_OP_SQ) return launch_fused_op<NB32, COL_LO, ROW_HI, FU_OP_SQ>(a, combine, stream);        \
        if (op == FU_OP_SIGN) return launch_fused_op<NB32, COL_LO, ROW_HI, FU_OP_SIGN>(a, combine, stream);    \
        if (op == FU_OP_UNIT) return launch_fused_op<NB32, COL_LO, ROW_HI, FU_OP_UNIT>(a, combine, stream);    \
        return launch_fused_op<NB32, COL_LO, ROW_HI, FU_OP_ABS>(a, combine, stream);
    switch (shape) {
        FU_SHAPES(FU_CASE)
    default:
        sc_set_error("fused kernel: no launch shape (%d staged blocks, column %d, %d rows)", a.NB32, a.shape_col_lo, a.shape_row_hi);
        return SC_EINVAL;
    }
#undef FU_CASE
}

// ---- up to 48 channels (58 for the planes with no matrix-core form): f32 VALU kernel -------------------------
// The MFMA kernel above stages 32-row chunks of 32-channel blocks whatever C is, so its cost per observation row
// does not fall with C: at 32 channels it runs 1.0 TB/s of input, at 8 channels 0.3 TB/s, where the arithmetic
// (C (C+1) / 2 pairs x 6 flops-ish per row) would leave the stream HBM-bound.  Here a thread owns a 2 x 2 block of
// channel pairs and every S-th observation: two 16-byte LDS reads and 24 VALU operations per row for four
// cross-spectra (re, im) and their |Im| sums, exact f32 products, two-level f32 sums (512 rows per first-level
// chain).  S = 448 / #blocks slices share a workgroup (7 waves) and are summed in a fixed order at the end.
// At the cfg3 input volume (6.5 GB) it takes 1.1-1.3 ms for 2-8 channels (5-6 TB/s: the read stream), 1.8 ms
// for 16, 2.9 ms for 32 (VALU-bound from ~16 channels on) against 41 / 20 / 10 / 6.3 ms on the MFMA kernel.
// Rows are staged HBM -> registers -> LDS in chunks (double buffered, one barrier per chunk).  Same records,
// same (bin, part) split and the same combine kernel as the MFMA path.
#define SM_THREADS 448
#define SM_CHUNK_BYTES (24 * 1024)

template <bool ABS, bool SQ, bool SGN>
struct SmallSlots { static constexpr int NQ = 2 + (ABS ? 1 : 0) + (SQ ? 1 : 0) + (SGN ? 1 : 0); };     // quantities summed

template <bool ABS, bool SQ, bool SGN, bool NORM>
__global__ void __launch_bounds__(SM_THREADS) small_csm_absim_kernel(FusedArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int bin = blockIdx.x / p.n_split, part = blockIdx.x - bin * p.n_split;
    const int g = bin / p.F, f = bin - g * p.F;
    ScStage st = p.st;
    st.base = p.st.base + (int64_t)f * st.ax.sF + sc_group_offset(st.ax, g);
    const int nc32 = (p.st.n_obs + FU_OC - 1) / FU_OC;
    const int o_lo = (int)((int64_t)part * nc32 / p.n_split) * FU_OC;
    int o_hi = (int)((int64_t)(part + 1) * nc32 / p.n_split) * FU_OC;
    if (o_hi > p.st.n_obs) o_hi = p.st.n_obs;
    float* rec = (part == 0 ? p.accum : p.ws + (int64_t)(part - 1) * p.n_bins * p.floats_per_bin) +
                 (int64_t)bin * p.floats_per_bin;

    const int C = st.C, B = C >> 1, nbk = B * (B + 1) / 2;     // C even: 2 x 2 blocks of pairs (bi <= bj)
    const int S = SM_THREADS / nbk;                             // observation slices (>= 1: nbk <= 300)
    const int s = tid / nbk, b = tid - s * nbk;
    const bool active = s < S;
    int bi = 0, bj = 0;
    { int rem = b, len = B; while (rem >= len) { rem -= len; ++bi; --len; } bj = bi + rem; }
    const int RS = 2 * C;                                       // floats per staged row
    const int half = C >> 1;                                    // float4 per row
    int OC = SM_CHUNK_BYTES / (RS * 4);
    OC -= OC % S;                                               // every slice gets the same number of rows per chunk
    if (OC > 16 * S) OC = 16 * S;
    const int per_thread = OC / S;
    float* buf0 = reinterpret_cast<float*>(smem);
    float* buf1 = buf0 + (size_t)OC * RS;
    const int total4 = OC * half;
    constexpr int EMAX = (SM_CHUNK_BYTES / 16 + SM_THREADS - 1) / SM_THREADS;
    float4 stage[EMAX];
    auto load = [&](int o0) {
#pragma unroll
        for (int i = 0; i < EMAX; ++i) {
            const int e = tid + i * SM_THREADS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < total4) {
                const int row = e / half, q = e - row * half, o = o0 + row;
                if (o < o_hi) {
                    v = *reinterpret_cast<const float4*>(st.base + sc_stage_obs_offset(st, o) + 2 * q);
                    if constexpr (NORM) {          // x / |x|: the CSM of these is sum s / |s| (0 / 0 -> NaN like the reference)
                        const float ia = rsqrtf(v.x * v.x + v.y * v.y), ib = rsqrtf(v.z * v.z + v.w * v.w);
                        v = make_float4(v.x * ia, v.y * ia, v.z * ib, v.w * ib);
                    }
                }
            }
            stage[i] = v;
        }
    };
    auto store = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < EMAX; ++i) {
            const int e = tid + i * SM_THREADS;
            if (e < total4) {
                const int row = e / half, q = e - row * half;
                *reinterpret_cast<float4*>(dst + row * RS + 4 * q) = stage[i];
            }
        }
    };

    float re[4] = {0.f, 0.f, 0.f, 0.f}, im[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    float re2[4] = {0.f, 0.f, 0.f, 0.f}, im2[4] = {0.f, 0.f, 0.f, 0.f}, ab2[4] = {0.f, 0.f, 0.f, 0.f};
    float sq[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f}, sq2[4] = {0.f, 0.f, 0.f, 0.f}, sg2[4] = {0.f, 0.f, 0.f, 0.f};
    const int n_chunks = (o_hi - o_lo + OC - 1) / OC;
    const int fold_every = (512 + per_thread - 1) / per_thread;        // chunks per first-level chain
    if (n_chunks > 0) { load(o_lo); store(buf0); }
    __syncthreads();
    for (int ch = 0; ch < n_chunks; ++ch) {
        const float* cur = (ch & 1) ? buf1 : buf0;
        float* nxt = (ch & 1) ? buf0 : buf1;
        const bool more = ch + 1 < n_chunks;
        if (more) load(o_lo + (ch + 1) * OC);
        if (active) {
            const float* ra = cur + 4 * bi;
            const float* rb = cur + 4 * bj;
            for (int k = 0; k < per_thread; ++k) {
                const int row = s + S * k;
                const float4 xa = *reinterpret_cast<const float4*>(ra + row * RS);     // x_{2bi}, x_{2bi+1}
                const float4 xb = *reinterpret_cast<const float4*>(rb + row * RS);     // x_{2bj}, x_{2bj+1}
                const float ar[2] = {xa.x, xa.z}, ai[2] = {xa.y, xa.w}, br[2] = {xb.x, xb.z}, bm[2] = {xb.y, xb.w};
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const int e = 2 * u + v;
                        re[e] = fmaf(ar[u], br[v], fmaf(ai[u], bm[v], re[e]));
                        const float d = fmaf(ai[u], br[v], -(ar[u] * bm[v]));
                        im[e] += d;
                        if constexpr (ABS) ab[e] += fabsf(d);
                        if constexpr (SQ) sq[e] = fmaf(d, d, sq[e]);
                        if constexpr (SGN) sg[e] += (d > 0.f ? 1.f : 0.f) - (d < 0.f ? 1.f : 0.f);     // sign(0) = 0
                    }
            }
        }
        if ((ch + 1) % fold_every == 0 || !more) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                re2[e] += re[e]; im2[e] += im[e]; ab2[e] += ab[e]; sq2[e] += sq[e]; sg2[e] += sg[e];
                re[e] = 0.f; im[e] = 0.f; ab[e] = 0.f; sq[e] = 0.f; sg[e] = 0.f;
            }
        }
        if (more) store(nxt);
        __syncthreads();
    }
    // slices -> one total per (block, quantity), summed in slice order; then the record image (zero padded tiles,
    // both triangles of the diagonal tiles like an MFMA tile) is assembled in LDS and copied out coalesced
    // Only the quantities of this instantiation get a slot (re, im, then |im|, im^2, sign(im) as present): with all five
    // the tail needed 85 KB from 50 channels on (10 tiles) and one workgroup per CU instead of two -- CSM + |Im| + Im^2
    // took 8.2 ms at 50 channels against 5.1 ms at 48.
    constexpr int NQ = SmallSlots<ABS, SQ, SGN>::NQ;
    constexpr int Q_AB = 2, Q_SQ = 2 + (ABS ? 1 : 0), Q_SG = 2 + (ABS ? 1 : 0) + (SQ ? 1 : 0);
    float* red = reinterpret_cast<float*>(smem);                         // [NQ * 4][SM_THREADS]
    float* image = red + NQ * 4 * SM_THREADS;                            // [NQ][n_tiles][256]
    const int plane_f = p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[(e) * SM_THREADS + tid] = active ? re2[e] : 0.f;
        red[(4 + e) * SM_THREADS + tid] = active ? im2[e] : 0.f;
        if constexpr (ABS) red[(4 * Q_AB + e) * SM_THREADS + tid] = active ? ab2[e] : 0.f;
        if constexpr (SQ) red[(4 * Q_SQ + e) * SM_THREADS + tid] = active ? sq2[e] : 0.f;
        if constexpr (SGN) red[(4 * Q_SG + e) * SM_THREADS + tid] = active ? sg2[e] : 0.f;
    }
    for (int i = tid; i < NQ * plane_f; i += SM_THREADS) image[i] = 0.f;
    __syncthreads();
    for (int item = tid; item < NQ * 4 * nbk; item += SM_THREADS) {
        const int q = item / nbk, bb = item - q * nbk;                  // q = quantity * 4 + element
        float acc = 0.f;
        for (int ss = 0; ss < S; ++ss) acc += red[q * SM_THREADS + ss * nbk + bb];
        int ti = 0, tj = 0;
        { int rem = bb, len = B; while (rem >= len) { rem -= len; ++ti; --len; } tj = ti + rem; }
        const int qty = q >> 2, e = q & 3, i = 2 * ti + (e >> 1), j = 2 * tj + (e & 1);
        if (i > j) continue;                                             // lower half of a diagonal 2 x 2 block
        const bool odd = qty == 1 || (SGN && qty == Q_SG);               // Im s and sign(Im s) change sign under i <-> j
        float* pl = image + qty * plane_f + sc_tile_index(i >> 4, j >> 4, p.NB) * SC_TILE_ELEMS;
        pl[(i & 15) * 16 + (j & 15)] = (odd && i == j) ? 0.f : acc;
        if ((i >> 4) == (j >> 4) && i != j) pl[(j & 15) * 16 + (i & 15)] = odd ? -acc : acc;
    }
    __syncthreads();
    if (p.csm_plane >= 0)
        for (int i = tid; i < 2 * plane_f; i += SM_THREADS) rec[(int64_t)p.csm_plane * plane_f + i] = image[i];
    if constexpr (ABS)
        for (int i = tid; i < plane_f; i += SM_THREADS) rec[(int64_t)p.abs_plane * plane_f + i] = image[Q_AB * plane_f + i];
    if constexpr (SQ)
        for (int i = tid; i < plane_f; i += SM_THREADS) rec[(int64_t)p.sq_plane * plane_f + i] = image[Q_SQ * plane_f + i];
    if constexpr (SGN)
        for (int i = tid; i < plane_f; i += SM_THREADS) rec[(int64_t)p.sign_plane * plane_f + i] = image[Q_SG * plane_f + i];
}

template <bool ABS, bool SQ, bool SGN, bool NORM>
static void launch_small_inst(const FusedArgs& a, hipStream_t stream) {
    constexpr int NQ = SmallSlots<ABS, SQ, SGN>::NQ;
    size_t shmem = 2 * (size_t)SM_CHUNK_BYTES;
    const size_t tail = (size_t)(4 * NQ * SM_THREADS + NQ * a.n_tiles * SC_TILE_ELEMS) * sizeof(float);
    if (shmem < tail) shmem = tail;
    auto k = small_csm_absim_kernel<ABS, SQ, SGN, NORM>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3((unsigned)(a.n_bins * a.n_split)), dim3(SM_THREADS), shmem, stream, a);
}

static int launch_small(const FusedArgs& a_in, bool normalize, hipStream_t stream) {
    FusedArgs a = a_in;
    a.n_fold = 0;
    if (a.csm_plane >= 0) { a.fold[a.n_fold++] = a.csm_plane; a.fold[a.n_fold++] = a.csm_plane + 1; }
    if (a.abs_plane >= 0) a.fold[a.n_fold++] = a.abs_plane;
    if (a.sq_plane >= 0) a.fold[a.n_fold++] = a.sq_plane;
    if (a.sign_plane >= 0) a.fold[a.n_fold++] = a.sign_plane;
    if (normalize) launch_small_inst<false, false, false, true>(a, stream);
    else if (a.sign_plane >= 0) launch_small_inst<false, false, true, false>(a, stream);
    else if (a.sq_plane >= 0) launch_small_inst<true, true, false, false>(a, stream);
    else if (a.abs_plane >= 0) launch_small_inst<true, false, false, false>(a, stream);
    else launch_small_inst<false, false, false, false>(a, stream);
    SC_CHECK_HIP(hipGetLastError());
    if (a.n_split > 1) {
        hipLaunchKernelGGL(planes_combine_kernel, dim3(2048), dim3(256), 0, stream, a);
        SC_CHECK_HIP(hipGetLastError());
    }
    return SC_OK;
}

// d_X may be NULL when only the shape is known: alignment is then assumed.
static bool fused_ok(const void* d_X, const ScAxes& ax) {
    if (ax.C < 1 || ax.C > 256 || (ax.C & 1)) return false;
    if ((ax.sW | ax.sR | ax.sK | ax.sF) & 1) return false;
    return d_X == nullptr || (((uintptr_t)d_X) % 16 == 0);
}

// The f32 VALU kernel below the measured crossover (same input volume as cfg3: 4.7 vs 5.1 ms at 48 channels with a
// per-observation non-linear plane, 3.0 vs 3.9 ms at 40 channels without; the MFMA kernel wins from 56 / 44 on).
static bool small_ok(const ScAxes& ax, bool nonlinear_plane) { return ax.C <= (nonlinear_plane ? 48 : 42); }
// (Im s)^2 rides along with CSM + |Im s| on this kernel up to 58 channels (as far as its 448 threads reach), sign(Im s)
// runs on it up to 44: above, a plane pass of the matrix-core kernel is faster.  (Round 3, with two workgroups per CU at
// every size -- the LDS tail holds only the quantities in use --, same input volume as cfg3: CSM + |Im| + Im^2 in one pass
// 5.2 / 5.1 / 5.4 ms at 50 / 52 / 58 channels against 4.7 + ~3.3 in two; sign 5.3 ms at 44 channels here, 5.7 at 48 there.)
static bool small_ok_sq(const ScAxes& ax) { return ax.C <= 58; }
static bool small_ok_sign(const ScAxes& ax) { return ax.C <= 44; }

extern "C" int sc_fused_supported(int64_t n_signals) {
    return (n_signals >= 2 && n_signals <= 256 && (n_signals % 2) == 0) ? 1 : 0;
}

// Workgroups per bin.  One workgroup fills a CU (LDS), so n_bins workgroups run in ceil(n_bins / n_cu)
// rounds and the last round may be nearly empty (903 bins on 256 CUs: 4 rounds for 3.53 rounds of
// work).  Splitting every bin's observations over S workgroups shortens the rounds; pick the S
// with the fewest (rounds / S), keeping >= 16 chunks per part (S <= 24: few bins with many observations).
int sc_internal_fused_pick_split(int n_bins, int n_obs) {
    const char* e = sc_switch(SC_SW_FUSED_SPLIT);
    const int nc = (n_obs + FU_OC - 1) / FU_OC;
    static int cu_of_device[64] = {0};          // compute units per device, queried once
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (cu_of_device[dev] == 0) {
            int v = 0;
            cu_of_device[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        }
        n_cu = cu_of_device[dev];
    }
    int best = 1;
    double best_cost = 1e30;
    for (int S = 1; S <= 24; ++S) {
        if (S > 1 && nc / S < 16) break;
        // rounds of workgroups x (chunks of a part + ~3 chunks' worth of prologue, record write and drain) x 2 % per extra
        // partial record the epilogue / combine reads.  Fitted on 125 ... 1000 trials of cfg3 (round 4: with the flat 1.5 % per
        // part of before, 250 trials took three parts and ran 5 % slower than one)
        const double rounds = (double)(((int64_t)n_bins * S + n_cu - 1) / n_cu);
        const double cost = rounds * ((double)nc / S + 3.0) * (1.0 + 0.02 * (S - 1));
        if (cost < best_cost - 1e-9) { best_cost = cost; best = S; }
    }
    if (e && atoi(e) >= 1 && atoi(e) <= 24 && (atoi(e) == 1 || nc / atoi(e) >= 1)) best = atoi(e);
    return best;
}

// The CSM waves' share of a launch: R = the tile rows that have tiles, row r with the columns max(r, col_lo) ... NB-1.
// Five or more rows: wave w takes rows w and R-1-w (long with short: the triangle's 9 tiles per wave at 128 channels);
// three or four: one row per wave; two rows: two waves per row, half the columns each; one row: a quarter each.
void sc_internal_fu_assign_rows(FusedArgs* a) {
    const int NB = a->NB, col_lo = a->map.col_lo;
    const int R = a->map.row_hi < NB ? a->map.row_hi : NB;
    auto c0 = [&](int r) { return r > col_lo ? r : col_lo; };
    unsigned packed[4];
    a->seg_n = 0u;
    for (int w = 0; w < 4; ++w) {
        int sg[6] = {0, 0, 0, 0, 0, 0};
        if (R >= 5) {
            const int rA = w, rB = R - 1 - w;
            if (rA <= rB) { sg[0] = rA; sg[1] = c0(rA); sg[2] = NB - c0(rA); }
            if (rB > rA) { sg[3] = rB; sg[4] = c0(rB); sg[5] = NB - c0(rB); }
        } else if (R >= 3) {
            if (w < R) { sg[0] = w; sg[1] = c0(w); sg[2] = NB - c0(w); }
        } else {
            const int per_row = R == 2 ? 2 : 4, r = R == 2 ? (w & 1) : 0, k = R == 2 ? (w >> 1) : w;
            const int n = NB - c0(r), lo = n * k / per_row, hi = n * (k + 1) / per_row;
            sg[0] = r; sg[1] = c0(r) + lo; sg[2] = hi - lo;
        }
        packed[w] = (unsigned)sg[0] | (unsigned)sg[1] << 4 | (unsigned)sg[3] << 8 | (unsigned)sg[4] << 12;
        a->seg_n |= ((unsigned)sg[2] | (unsigned)sg[5] << 4) << (8 * w);
    }
    a->seg0 = packed[0]; a->seg1 = packed[1]; a->seg2 = packed[2]; a->seg3 = packed[3];
}

// One launch that stages the nb (<= 4) 32-channel blocks `blocks` (ascending block numbers of the record's channels) and
// owns the products (bi <= bj, bj >= col_lo, bi < row_hi) of them.
static FusedArgs fu_args_blocks(const FusedArgs& full, const int* blocks, int nb, int col_lo, int row_hi) {
    FusedArgs a = full;
    const int C = full.st.C, c_lo = blocks[0] * 32;
    a.st.base = full.st.base + c_lo;
    int staged = 0, n_last = 0;
    a.map.off32 = a.map.n32 = a.map.t32 = 0u;
    for (int b = 0; b < nb; ++b) {
        const int c = blocks[b] * 32;
        n_last = C - c < 32 ? C - c : 32;
        a.map.off32 |= (unsigned)(blocks[b] - blocks[0]) << (8 * b);
        a.map.n32 |= (unsigned)n_last << (8 * b);
        a.map.t32 |= (unsigned)(blocks[b] * 2) << (8 * b);
        staged += n_last;
    }
    a.st.C = staged;
    a.NB32 = nb;
    a.NB = 2 * (nb - 1) + (n_last + 15) / 16;      // 16-channel tiles that exist (the last block may be partial)
    a.shape_col_lo = col_lo;
    a.shape_row_hi = row_hi;
    a.n_blocks32 = fu_nblocks(nb, col_lo, row_hi);
    a.n_sets = fu_nsets(nb, col_lo, row_hi);
    a.st.CP = nb * 32;
    a.st.RS = sc_row_stride(a.st.CP);
    a.map.NBr = sc_n_blocks(C);
    a.map.col_lo = 2 * col_lo;
    a.map.row_hi = 2 * row_hi;
    sc_internal_fu_assign_rows(&a);
    return a;
}
static int launch_fused(const FusedArgs& a, int op, hipStream_t stream, bool combine);
// Every tile of the record once.  Up to 128 channels: one launch, the triangle of its 1 ... 4 blocks.  Above, a launch
// can stage four of the n = 5 ... 8 blocks at a time, and the n (n + 1) / 2 block products are dealt over launches so
// that few blocks are staged twice (each staging reads its channels from HBM again):
//   n = 5   triangle {0,1,2};  {0,1} x {3,4};  {2} x {3,4} + triangle {3,4}                        15 products, 10 staged
//   n = 6   triangle {0,1,2,3};  {0,1} x {4,5} + triangle {4,5};  {2,3} x {4,5}                     21 products, 12 staged
//   n = 7   triangle {0,1,2,3};  {0} x {4,5,6} + triangle {4,5,6};  {1}, {2}, {3} x {4,5,6}         28 products, 20 staged
//   n = 8   triangle {0..3};  triangle {4..7};  {0,1}, {2,3} x {4,5}, {6,7}                         36 products, 24 staged
// (round 2 ran every count in 129 ... 255 as n = 8: 160 channels cost what 256 do).  The split-bin partial records are
// folded once at the end.
static int launch_fused_all(const FusedArgs& full, int op, hipStream_t s) {
    const int C = full.st.C, n = (C + 31) / 32;
    struct Plan { int nb, blocks[4], col_lo, row_hi; };
    static const Plan tri[4] = {{1, {0}, 0, 1}, {2, {0, 1}, 0, 2}, {3, {0, 1, 2}, 0, 3}, {4, {0, 1, 2, 3}, 0, 4}};
    static const Plan p5[] = {{3, {0, 1, 2}, 0, 3}, {4, {0, 1, 3, 4}, 2, 2}, {3, {2, 3, 4}, 1, 3}};
    static const Plan p6[] = {{4, {0, 1, 2, 3}, 0, 4}, {4, {0, 1, 4, 5}, 2, 4}, {4, {2, 3, 4, 5}, 2, 2}};
    static const Plan p7[] = {{4, {0, 1, 2, 3}, 0, 4}, {4, {0, 4, 5, 6}, 1, 4}, {4, {1, 4, 5, 6}, 1, 1},
                              {4, {2, 4, 5, 6}, 1, 1}, {4, {3, 4, 5, 6}, 1, 1}};
    static const Plan p8[] = {{4, {0, 1, 2, 3}, 0, 4}, {4, {4, 5, 6, 7}, 0, 4}, {4, {0, 1, 4, 5}, 2, 2},
                              {4, {0, 1, 6, 7}, 2, 2}, {4, {2, 3, 4, 5}, 2, 2}, {4, {2, 3, 6, 7}, 2, 2}};
    const Plan* plan = n <= 4 ? &tri[n - 1] : n == 5 ? p5 : n == 6 ? p6 : n == 7 ? p7 : p8;
    const int n_launch = n <= 4 ? 1 : n == 5 ? 3 : n == 6 ? 3 : n == 7 ? 5 : 6;
    int rc = SC_OK;
    for (int l = 0; l < n_launch && rc == SC_OK; ++l)
        rc = launch_fused(fu_args_blocks(full, plan[l].blocks, plan[l].nb, plan[l].col_lo, plan[l].row_hi), op, s, false);
    if (rc == SC_OK) rc = sc_internal_fused_combine(full, op, s);
    return rc;
}

enum { FU_MODE_CSM = 0, FU_MODE_UNIT = 1, FU_MODE_SIGN = 2 };
static int fused_setup(const void* d_X, const sc_spectra_desc* desc, uint32_t planes, int mode, FusedArgs* a, ScAxes* ax) {
    SC_REQUIRE(desc, "NULL argument");
    if (mode == FU_MODE_UNIT) SC_REQUIRE(planes & SC_PLANE_UNIT, "planes must contain SC_PLANE_UNIT");
    else if (mode == FU_MODE_SIGN) SC_REQUIRE(planes & SC_PLANE_SIGN_IM, "planes must contain SC_PLANE_SIGN_IM");
    else SC_REQUIRE(planes & SC_PLANE_CSM, "planes must contain SC_PLANE_CSM (SC_PLANE_ABS_IM is optional)");
    sc_make_axes(desc, ax);
    SC_REQUIRE(ax->C >= 1 && ax->F >= 1 && ax->n_obs >= 1 && ax->n_groups >= 1, "empty dimension");
    if (!fused_ok(d_X, *ax)) {
        sc_set_error("fused CSM+|Im| kernel needs an even n_signals <= 256 and 16-byte aligned rows (got C=%d)", ax->C);
        return SC_EUNSUPPORTED;
    }
    a->NB = sc_n_blocks(ax->C);
    a->n_tiles = sc_n_tiles(a->NB);
    a->NB32 = (ax->C + 31) / 32;
    a->n_blocks32 = a->NB32 * (a->NB32 + 1) / 2;
    a->n_sets = 1;                            // (per launch: fu_args_blocks)
    a->n_bins = ax->n_groups * ax->F;
    a->F = ax->F;
    a->floats_per_bin = (int64_t)sc_plane_count(planes) * a->n_tiles * SC_TILE_ELEMS;
    a->csm_plane = sc_plane_offset(planes, SC_PLANE_CSM);
    a->abs_plane = (planes & SC_PLANE_ABS_IM) ? sc_plane_offset(planes, SC_PLANE_ABS_IM) : -1;   // -1: CSM only
    a->sq_plane = -1;
    a->sign_plane = -1;
    if (mode == FU_MODE_UNIT) {   // sum s / |s| = the CSM of x / |x|: the same kernels, pointed at the unit-phasor planes
        a->csm_plane = sc_plane_offset(planes, SC_PLANE_UNIT);
        a->abs_plane = -1;
    } else if (mode == FU_MODE_SIGN) {
        a->csm_plane = -1;
        a->abs_plane = -1;
        a->sign_plane = sc_plane_offset(planes, SC_PLANE_SIGN_IM);
    } else if ((planes & SC_PLANE_ABS_IM) && (planes & SC_PLANE_IM_SQ)) {
        // rides along on the small-channel kernel; a plane pass of the matrix-core kernel above its range
        a->sq_plane = sc_plane_offset(planes, SC_PLANE_IM_SQ);
    }
    a->nl_op = FU_OP_ABS;
    a->n_split = 1;
    a->ws = nullptr;
    return SC_OK;
}

extern "C" int64_t sc_fused_workspace_bytes(const sc_spectra_desc* desc, uint32_t planes) {
    FusedArgs a;
    ScAxes ax;
    const int mode = (planes & SC_PLANE_CSM) ? FU_MODE_CSM : (planes & SC_PLANE_UNIT) ? FU_MODE_UNIT : FU_MODE_SIGN;
    if (fused_setup(nullptr, desc, planes, mode, &a, &ax) != SC_OK) return 0;
    const int S = sc_internal_fused_pick_split(a.n_bins, ax.n_obs);
    return (int64_t)(S - 1) * a.n_bins * a.floats_per_bin * (int64_t)sizeof(float);
}

static int fused_run(const void* d_X, const sc_spectra_desc* desc, uint32_t planes, int mode, float* d_accum,
                     void* d_workspace, int64_t workspace_bytes, void* d_scratch, int64_t scratch_bytes, void* stream) {
    ScTimed timed_("fused_stage_b", stream);
    SC_REQUIRE(d_X && desc && d_accum, "NULL argument");
    FusedArgs a;
    ScAxes ax;
    const bool unit = mode == FU_MODE_UNIT;
    const int rc = fused_setup(d_X, desc, planes, mode, &a, &ax);
    if (rc != SC_OK) return rc;
    a.accum = d_accum;
    a.st.base = (const float2*)d_X;
    a.st.ax = ax;
    a.st.obs_stride = sc_stage_linear_stride(ax);
    a.st.C = ax.C;
    a.st.CP = a.NB32 * 32;                    // VALU blocks need 32-channel padding
    a.st.RS = sc_row_stride(a.st.CP);
    a.st.n_obs = ax.n_obs;
    {
        const char* dbg = sc_switch(SC_SW_FUSED_DEBUG);
        a.debug_skip = dbg ? atoi(dbg) : 0;
    }
    // as many parts per bin as the workspace allows (none: one workgroup per bin)
    int S = sc_internal_fused_pick_split(a.n_bins, ax.n_obs);
    const int64_t part_bytes = (int64_t)a.n_bins * a.floats_per_bin * (int64_t)sizeof(float);
    if (!d_workspace) S = 1;
    while (S > 1 && (int64_t)(S - 1) * part_bytes > workspace_bytes) --S;
    SC_REQUIRE(S == 1 || ((uintptr_t)d_workspace % 16) == 0, "workspace must be 16-byte aligned");
    a.n_split = S;
    a.ws = (float*)d_workspace;
    hipStream_t s = (hipStream_t)stream;
    const char* no_small = sc_switch(SC_SW_FUSED_NO_SMALL);       // diagnostic: every shape through the matrix-core kernel
    if (!(no_small && atoi(no_small)) &&
        (small_ok(ax, a.abs_plane >= 0) || (a.sq_plane >= 0 && small_ok_sq(ax)) || (mode == FU_MODE_SIGN && small_ok_sign(ax))))
        return launch_small(a, unit, s);
    if (mode == FU_MODE_SIGN) {
        // plane pass: sign(d) of the per-observation matrix-core products, summed as integers by the abs waves
        FusedArgs b = a;
        b.csm_plane = -1;
        b.abs_plane = a.sign_plane;
        b.nl_op = FU_OP_SIGN;
        b.fold[0] = b.abs_plane;
        b.n_fold = 1;
        return launch_fused_all(b, FU_OP_SIGN, s);
    }
    // unit phasors (PLV / PPC): the staging waves normalise the rows on the way into LDS (FU_OP_UNIT) -- no normalised
    // copy of the spectra, no scratch (d_scratch / scratch_bytes are accepted and ignored)
    (void)d_scratch; (void)scratch_bytes;
    const int rc_main = launch_fused_all(a, unit ? FU_OP_UNIT : FU_OP_ABS, s);
    if (rc_main != SC_OK || a.sq_plane < 0) return rc_main;
    // debiased wPLI: sum (Im s)^2 as a second pass of the same kernel (the abs waves hold 80 accumulator registers
    // per plane; two planes do not fit next to the matrix-core role's)
    FusedArgs b = a;
    b.csm_plane = -1;
    b.abs_plane = a.sq_plane;
    b.nl_op = FU_OP_SQ;
    b.fold[0] = b.abs_plane;
    b.n_fold = 1;
    return launch_fused_all(b, FU_OP_SQ, s);
}

extern "C" int sc_fused_csm_absim_ws_f32(const void* d_X, const sc_spectra_desc* desc, uint32_t planes,
                                         float* d_accum, void* d_workspace, int64_t workspace_bytes, void* stream) {
    return fused_run(d_X, desc, planes, FU_MODE_CSM, d_accum, d_workspace, workspace_bytes, nullptr, 0, stream);
}

// The planes of `planes` that the one-pass entry points fill for this shape (even n_signals <= 128): CSM, |Im s| (with
// CSM), s/|s|, (Im s)^2 (with CSM and |Im s|) and sign(Im s).  Everything else is sc_nonlinear_accumulate_f32's.
extern "C" uint32_t sc_fused_planes_covered(const sc_spectra_desc* desc, uint32_t planes) {
    ScAxes ax;
    if (!desc || sc_make_axes(desc, &ax) != SC_OK || !fused_ok(nullptr, ax)) return 0;
    uint32_t got = planes & (SC_PLANE_CSM | SC_PLANE_UNIT);
    if ((planes & SC_PLANE_CSM) && (planes & SC_PLANE_ABS_IM)) got |= SC_PLANE_ABS_IM;
    if ((got & SC_PLANE_ABS_IM) && (planes & SC_PLANE_IM_SQ)) got |= SC_PLANE_IM_SQ;
    got |= planes & SC_PLANE_SIGN_IM;
    return got;
}

// SC_PLANE_SIGN_IM of the record (phase_lag_index, debiased_squared_phase_lag_index: connectivity.py:983-1079): the
// small-channel kernel up to 44 channels, above it a plane pass of the matrix-core kernel (the abs waves sum
// sign(d) of the per-observation products as integers).
extern "C" int sc_fused_sign_ws_f32(const void* d_X, const sc_spectra_desc* desc, uint32_t planes, float* d_accum,
                                    void* d_workspace, int64_t workspace_bytes, void* stream) {
    return fused_run(d_X, desc, planes, FU_MODE_SIGN, d_accum, d_workspace, workspace_bytes, nullptr, 0, stream);
}

// Scratch sc_fused_unit_ws_f32 needs for this shape: none since the rows are normalised at staging (kept for ABI v2).
extern "C" int64_t sc_fused_unit_scratch_bytes(const sc_spectra_desc* desc) {
    (void)desc;
    return 0;
}

// SC_PLANE_UNIT of the record: sum over observations of s / |s| = x_i conj(x_j) / (|x_i| |x_j|), i.e. the cross-
// spectral matrix of the unit phasors x / |x| (phase_locking_value, pairwise_phase_consistency: connectivity.py:
// 897-981).  Same shapes, workspace and split as sc_fused_csm_absim_ws_f32.
extern "C" int sc_fused_unit_ws_f32(const void* d_X, const sc_spectra_desc* desc, uint32_t planes, float* d_accum,
                                    void* d_workspace, int64_t workspace_bytes, void* d_scratch, int64_t scratch_bytes,
                                    void* stream) {
    return fused_run(d_X, desc, planes, FU_MODE_UNIT, d_accum, d_workspace, workspace_bytes, d_scratch, scratch_bytes, stream);
}

extern "C" int sc_fused_csm_absim_f32(const void* d_X, const sc_spectra_desc* desc, uint32_t planes,
                                      float* d_accum, void* stream) {
    return sc_fused_csm_absim_ws_f32(d_X, desc, planes, d_accum, nullptr, 0, stream);
}
