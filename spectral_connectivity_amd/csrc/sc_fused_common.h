// sc_fused_common.h -- what the two one-pass stage-B kernels share (sc_fused.hip: complex64 spectra, split into bf16 pieces
// while staging; sc_fused2.hip: spectra that stage A already wrote as bf16 pieces): launch arguments, the staged-block
// map, the 32x32 block tables of the |Im s| role and its accumulate step, and the host helpers of the split-bin scheme.
#pragma once
#include "sc_stage.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FU_OC 32            // observation rows per chunk (= K of the bf16 MFMA)
#define FU_THREADS 768
#define FU_MAXB 5
#define FU_FLUSH 16         // chunks between folds of the MFMA accumulators into the output record

// Which channels a workgroup stages and where its tiles live in the record.  The 128 staged channel slots are four
// blocks of 32: local block b holds n32[b] channels (even; every block ahead of the last staged one is full) starting
// off32[b] elements from st.base, and its two 16-channel tiles are tiles t32[b], t32[b] + 1 of the record (NBr = tile
// rows of the record).  One contiguous range is every launch up to 128 channels; 129 ... 256 channels are covered by
// several launches whose staged blocks come from up to two ranges (see launch_fused_all).  Which of the staged blocks'
// products a launch owns is a staircase of the local upper triangle: tile (r, c), r <= c, is active when c >= col_lo and
// r < row_hi (in 16-channel tiles; the whole triangle: col_lo = 0, row_hi = NB).
// The per-block values are packed one byte each (block b in bits 8 b ... 8 b + 7: one bit-field extract for a per-lane
// b, where an indexed array in the kernel arguments would go through scratch): off32 in units of 32 channels.
struct FuMap {
    unsigned off32, n32, t32;
    int NBr;
    int col_lo, row_hi;
};
__host__ __device__ inline int fu_byte(unsigned packed, int b) { return (int)((packed >> (8 * b)) & 0xffu); }
__host__ __device__ inline int fu_gt(const FuMap& m, int b) { return fu_byte(m.t32, b >> 1) + (b & 1); }
__host__ __device__ inline bool fu_tile_ok(const FuMap& m, int b) { return (b & 1) * 16 < fu_byte(m.n32, b >> 1); }

struct FusedArgs {
    ScStage st;
    FuMap map;
    float* accum;
    int64_t floats_per_bin;
    int n_bins, F, NB, n_tiles, NB32, n_blocks32, n_sets;
    int shape_col_lo, shape_row_hi;   // the launch's 32x32 blocks: bi <= bj, bj >= shape_col_lo, bi < shape_row_hi
    int csm_plane, abs_plane;
    int sq_plane, sign_plane;   // small-channel kernel only: sum (Im s)^2, sum sign(Im s); -1 = absent
    int fold[6], n_fold;        // small-channel kernel only: the record planes a launch writes (folded over the parts)
    int nl_op;           // what the abs waves accumulate from the per-observation d = Im(x_i conj x_j) into record plane
                         // `abs_plane`: FU_OP_ABS |d| (with the CSM planes, one pass), FU_OP_SQ d^2, FU_OP_SIGN sign(d)
                         // (plane passes: csm_plane = -1, the four CSM waves only stage)
    unsigned seg0, seg1, seg2, seg3, seg_n;   // tiles of CSM wave w (fu_assign_rows): segw = row A | first column A << 4 |
                         // row B << 8 | first column B << 12; byte w of seg_n = count A | count B << 4
    int n_split;         // workgroups per bin: part k sums the chunks [k NC / n_split, (k+1) NC / n_split)
    float* ws;           // partial records of parts 1 .. n_split-1: [n_split-1][n_bins][floats_per_bin]
    int debug_skip;      // profiling aid (env SC_FUSED_DEBUG, bit mask; results are WRONG when set):
                         // 1 = CSM waves skip their MFMAs, 2 = abs waves skip theirs, 8 = no HBM loads after chunk 0
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

// Lane id re-materialised on the spot (never CSE'd or hoisted): everything derived from it has a short
// live range, so the register allocator does not carry -- and spill -- per-lane constants of one phase
// across the other.  (A spill reload bumps vmcnt and would make the wave sit out the HBM->LDS loads.)
__device__ __forceinline__ int fu_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
enum { FU_OP_ABS = 0, FU_OP_SQ = 1, FU_OP_SIGN = 2, FU_OP_UNIT = 3 };     // UNIT: CSM role only, rows normalised at staging

// ---- VALU role ---------------------------------------------------------------------------------
// Upper-triangular 32x32 blocks, row-major: t -> (BI, BJ).  Tables are compile-time (template on
// the number of 32-channel blocks and the set) so that operand fragments shared by several blocks
// of a set (same BI or same BJ) are loaded from LDS once per observation row.
// A launch's 32x32 blocks: (bi, bj), bi <= bj, bj >= COL_LO, bi < ROW_HI of the NB32 staged blocks, row-major, dealt
// over one or two sets of at most FU_MAXB (a set = the blocks one abs wave accumulates).
__host__ __device__ constexpr int fu_nblocks(int nb32, int col_lo, int row_hi) {
    int n = 0;
    for (int bi = 0; bi < nb32 && bi < row_hi; ++bi)
        for (int bj = bi > col_lo ? bi : col_lo; bj < nb32; ++bj) ++n;
    return n;
}
// Up to four blocks: one set, its rows dealt over all eight abs waves; more: two sets of four waves, the first with
// floor(n / 2) blocks (the sets of the 128-channel triangle, 5 + 5, are what fits the 168-register budget of a
// 12-wave workgroup: five blocks in ONE set of a three-block launch spilled 28 registers, 5 + 4 in that order 6).
__host__ __device__ constexpr int fu_nsets(int nb32, int col_lo, int row_hi) {
    return fu_nblocks(nb32, col_lo, row_hi) > 4 ? 2 : 1;
}
__host__ __device__ constexpr int fu_set_first(int nb32, int col_lo, int row_hi, int set) {
    return set == 0 ? 0 : fu_nblocks(nb32, col_lo, row_hi) / 2;
}
__host__ __device__ constexpr int fu_set_count(int nb32, int col_lo, int row_hi, int set) {
    return fu_nsets(nb32, col_lo, row_hi) == 1 ? fu_nblocks(nb32, col_lo, row_hi)
         : (set == 0 ? fu_nblocks(nb32, col_lo, row_hi) / 2 : fu_nblocks(nb32, col_lo, row_hi) - fu_nblocks(nb32, col_lo, row_hi) / 2);
}
// t-th block of the shape: returns bi * 4 + bj
__host__ __device__ constexpr int fu_block(int nb32, int col_lo, int row_hi, int t) {
    for (int bi = 0; bi < nb32 && bi < row_hi; ++bi)
        for (int bj = bi > col_lo ? bi : col_lo; bj < nb32; ++bj) {
            if (t == 0) return bi * 4 + bj;
            --t;
        }
    return 0;
}

template <int NB32, int COL_LO, int ROW_HI, int SET>
struct FuTab {
    static constexpr int T0 = fu_set_first(NB32, COL_LO, ROW_HI, SET);
    static constexpr int NBLK = fu_set_count(NB32, COL_LO, ROW_HI, SET);   // blocks of this set
    static_assert(NBLK <= FU_MAXB, "a set holds at most FU_MAXB blocks");
    struct Arr { int bi[FU_MAXB]; int bj[FU_MAXB]; bool use_i[4]; bool use_j[4]; };
    static constexpr Arr make() {
        Arr a{};
        for (int s = 0; s < FU_MAXB; ++s) { a.bi[s] = 0; a.bj[s] = 0; }
        for (int b = 0; b < 4; ++b) { a.use_i[b] = false; a.use_j[b] = false; }
        for (int s = 0; s < NBLK; ++s) {
            const int blk = fu_block(NB32, COL_LO, ROW_HI, T0 + s);
            a.bi[s] = blk / 4;
            a.bj[s] = blk % 4;
            a.use_i[a.bi[s]] = true;
            a.use_j[a.bj[s]] = true;
        }
        return a;
    }
    static constexpr Arr tab = make();
};


// acc <- acc (+) f(d) for one output register of a 32x32 block: |d| (wPLI weights), d^2 (debiased wPLI), sign(d) (PLI)
template <int OP>
__device__ __forceinline__ float fu_accumulate(float acc, float d) {
    if constexpr (OP == FU_OP_ABS || OP == FU_OP_UNIT) return acc + fabsf(d);
    else if constexpr (OP == FU_OP_SQ) return fmaf(d, d, acc);
    else {
        // sign(d) in {-1, 0, 1} summed as an INTEGER in the accumulator's bits: the bit pattern of a float orders like a
        // signed integer with +0 = 0, so clamping it to [-1, 1] is the sign (one v_med3_i32 + one v_add_u32 per value; the
        // MFMA's C input is +0, so an exact zero comes out as +0).  Converted to float once, after the last chunk.
        const int b = __float_as_int(d);
        const int sg = b < -1 ? -1 : (b > 1 ? 1 : b);
        const int a = __float_as_int(acc) + sg;
        return __int_as_float(a);
    }
}
// The sixteen results of one 32x32 block.  sign(d): written out by the compiler, the sixteen clamps are scheduled ahead of
// their adds and the 168-register budget of a 12-wave workgroup spills, so fifteen of them are two-instruction asm blocks
// with one temporary each.  But the wait states between an MFMA write and a VALU read are the COMPILER's job, and it
// pads only in front of instructions it can see (an asm block straight after a lone MFMA read stale registers: wrong sign
// sums at <= 32 channels): element 0 goes first as ordinary code -- the compiler waits for the MFMA there -- and a
// scheduling barrier keeps the asm blocks behind it.
template <int OP, bool PACKED = false>
__device__ __forceinline__ void fu_accumulate16(f32x16& acc, const f32x16& d) {
    if constexpr (OP == FU_OP_SQ && PACKED) {
        // two squares per instruction (v_pk_fma_f32 on the register pairs of the MFMA result): 8 issue slots per block and
        // row instead of 16 -- |d| and sign(d) have no packed form (VOP3P carries no abs modifier).  Sets of up to four
        // blocks only: with five, the aligned pairs do not fit the register budget (9 spilled pairs per row, 3.7 -> 9 ms)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            f32x2 a = {acc[e], acc[e + 1]};
            const f32x2 v = {d[e], d[e + 1]};
            a = __builtin_elementwise_fma(v, v, a);
            acc[e] = a[0]; acc[e + 1] = a[1];
        }
    } else if constexpr (OP != FU_OP_SIGN) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = fu_accumulate<OP>(acc[e], d[e]);
    } else {
        acc[0] = fu_accumulate<OP>(acc[0], d[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 1; e < 16; ++e) {
            int a = __float_as_int(acc[e]), t;
            asm("v_med3_i32 %1, %2, -1, 1\n\tv_add_u32 %0, %0, %1" : "+v"(a), "=&v"(t) : "v"(d[e]));
            acc[e] = __int_as_float(a);
        }
    }
}


// host helpers defined in sc_fused.hip
int sc_internal_fused_pick_split(int n_bins, int n_obs);              // workgroups per bin
void sc_internal_fu_assign_rows(FusedArgs* a);                         // tiles of the four CSM waves
int sc_internal_fused_combine(const FusedArgs& a, int op, hipStream_t stream);   // fold the parts' records (n_split > 1)
