// sc_fused2.hip -- stage B on spectra that arrive as f16 pieces ("planes format"), round 4.
//
// The one-pass kernel of sc_fused.hip spends a third of its vector-issue slots on work that is not the product: splitting
// every f32 coefficient into three bf16 pieces while staging (208 of 2185 VALU instructions per SIMD and 32-row chunk),
// re-assembling the operands of the per-observation |Im s| products with byte permutes (456; v_perm_b32 costs 4.3 cycles
// against 2.1-2.6 for an add: profiles/r04_issue_rates.txt) and flipping signs (~100); and its matrix pipe runs SIX cross
// terms per product.  Here:
//   * every real number is stored as TWO f16 pieces, x * 2^e = h + m (h = f16(x 2^e), m = f16(x 2^e - h): 22 significant bits,
//     error <= 2^-23 |x| + 2^-25 in scaled units), with one power-of-two scale per channel that keeps every coefficient
//     inside the f16 range (sc_planes_scales_*: from a bound on the spectra, so nothing can overflow).  That is 8 bytes per
//     complex coefficient -- the complex64 volume -- and products need THREE cross terms (h h, h m, m h; the dropped m m is
//     2^-22 relative, with random sign over the observations) instead of six: half the matrix-pipe time of the cross-
//     spectral matrix.  The scales are powers of two, taken out again exactly when a record is written;
//   * an observation row is [channel tile of 32][plane Re h, Re m, Im h, Im m][32 channels] f16, and lands in LDS
//     observation-major and plane-major by direct HBM -> LDS loads (global_load_lds_dwordx4, no VGPRs, no VALU);
//   * BOTH roles read their matrix-core operands out of that one layout with the transposing LDS load
//     ds_read_b64_tr_b16 (lane 4 r + q of a 16-lane group supplies row r, 8-byte chunk q; lane j receives column j of the
//     four rows: profiles/r04_tr_load.txt): the CSM waves take rows = observations (K = 32 observations of one plane,
//     v_mfma_f32_16x16x32_f16), the |Im s| waves rows = PLANES of one observation (K = the four cross terms of one
//     observation, v_mfma_f32_32x32x8_f16) -- which plane sits in which K slot is just the address a lane passes;
//   * the negated real part both roles need (Im s = Im x_i Re x_j - Re x_i Im x_j) is a third plane pair in LDS, made by
//     the waves that loaded the real planes (one xor per 8 coefficients and chunk).
// LDS layout of a chunk of 32 observations (bytes):   addr(p, o, c) = p * 8512 + (o & 7) * 1056 + (o >> 3) * 256 + 2 c
//   p = plane 0 .. 5 (Re h m, Im h m, -Re h m), o = observation of the chunk, c = staged channel 0 .. 127
// A direct load instruction fills one 1024-byte group (plane p, observations o7, o7 + 8, o7 + 16, o7 + 24); the 32 bytes of
// padding behind every group and the 64 behind every plane shift the banks so that every transposing load of either role
// is conflict-free (3.0 cycles per instruction measured, against 8-16 for unpadded rows) while every address stays an
// affine function of (plane, observation, channel block): ONE address register per role and buffer, everything else
// immediate offsets.
// Accuracy: the representation error 2^-23 is half an f32 ulp; the dropped m m term is <= 2^-22 |x_i| |x_j| per observation
// (typically a third of that) -- below the rounding of the f32 transform that produced the coefficients, whose error in a
// weak bin is relative to the strongest one.  (First form of this file, three bf16 pieces and six terms: 12 bytes per
// coefficient, stage B 5.1 -> 4.6 ms but stage A +0.6 ms: profiles/r04_fused2_bf16x3.txt.)
#include <stdlib.h>
#include "sc_fused_common.h"

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

#define F2_GROUP 1056                 // 4 observation rows of 256 B + 32 B
#define F2_PLANE (8 * F2_GROUP + 64)  // 8512
#define F2_NPL 6                      // Re h m, Im h m, -Re h m
#define F2_HPL 4                      // planes of a row in HBM
#define F2_BUF (F2_NPL * F2_PLANE)    // 51072 B per chunk buffer
#define F2_NBUF 3                        // chunk n + 2 is loaded while chunk n is multiplied
#define F2_SCALE_OFF (F2_NBUF * F2_BUF)          // float[128]: 1 / scale of the staged channels (1 for absent ones)
#define F2_LDS (F2_SCALE_OFF + 512)
#define F2_ROW_TILE 256               // bytes of one 32-channel tile of an observation row in HBM (4 planes x 64 B)

extern "C" int64_t sc_planes_scales_work_bytes(int64_t n_rows, int64_t width);
extern "C" int64_t sc_planes_row_bytes(int64_t n_signals) { return (int64_t)F2_ROW_TILE * ((n_signals + 31) / 32); }

__device__ __forceinline__ unsigned f2_pack(float lo, float hi) {        // two f16 (round to nearest) in one dword
    const h16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float f2_lo(unsigned p) { return (float)__builtin_bit_cast(h16x2, p)[0]; }
__device__ __forceinline__ float f2_hi(unsigned p) { return (float)__builtin_bit_cast(h16x2, p)[1]; }
// two scaled reals (channels c, c + 1 of one component) -> their h and m dwords
__device__ __forceinline__ void f2_split2(float x0, float x1, unsigned& h, unsigned& m) {
    h = f2_pack(x0, x1);
    m = f2_pack(x0 - f2_lo(h), x1 - f2_hi(h));
}

// ---- scales -------------------------------------------------------------------------------------------------------------------
// scale[c] = 2^e with bound_c * 2^e <= 2^15 (65504 is the largest f16), from a per-channel bound on |Re X|, |Im X|:
//   from the time series: |X_k(f)| <= max_n |x - trend| * sum_n |h_k[n]| <= 8 max|x| * taper_abs_sum  (8: mean / line removed from
//   bounded data stays within a few times its range);   from complex64 spectra: max(|Re X|, |Im X|) itself.
// Zero / non-finite bounds give scale 1 (an all-zero channel stays zero; non-finite samples make non-finite coefficients in either
// format).  d_scale: float[C] scale, float[C] 1 / scale behind it.
// x: n_rows rows of `width` floats (width = C * comps); column col belongs to channel col / comps.  A block takes a slab of
// rows; with width % 4 == 0 a thread reads 16 bytes of 4 rows per step (1-2 KB in flight per wave-instruction), the rows
// of a column group meet in LDS, one atomic per column and block.
// |v| of a FINITE value, 0 otherwise: a channel with an infinite or NaN sample somewhere keeps the scale its finite samples ask
// for (the windows that hold the bad sample come out NaN either way, the others must not overflow the f16 range)
__device__ __forceinline__ float planes_mag(float v) {
    const float a = fabsf(v);
    return a <= 3.4028234e38f ? a : 0.f;
}
// Per column: the largest finite |x| (and, with `part`, the largest x and the largest -x separately: the range), plus -- for the
// quality check of the format -- per-block sums of (x - pivot) and (x - pivot)^2 around the block's own pivot (its first row: a DC
// offset 1e5 times the signal must not cancel the variance away in float32), [gridDim.x][width][3] = {pivot, s1, s2}; the scale
// kernel merges the blocks in float64 in a fixed order.
__device__ __forceinline__ float planes_fin(float v) { return fabsf(v) <= 3.4028234e38f ? v : 0.f; }
__global__ void __launch_bounds__(256) planes_absmax_kernel(const float* x, int64_t n_rows, int width, int comps, unsigned* mx, float* part) {
    __shared__ unsigned red[2048];                     // [width] max |x| or (with part) [width] max x, [width] max -x  (offset patterns, see below)
    __shared__ float sred[2048];                       // [rows_per_step][width][2]
    const int64_t rows_per_block = (n_rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    const int tid = threadIdx.x;
    // ordered bit pattern of a float (monotone for every finite value): max over patterns = max over values
    auto ord = [](float v) -> unsigned { const unsigned u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    if ((width & 3) == 0 && width <= 1024 && (((uintptr_t)x) & 15) == 0 && r0 < r1) {
        const int q = width >> 2;                      // float4 columns
        const int rows_per_step = 256 / q;             // >= 1 for width <= 1024
        const int cq = tid % q, ro = tid / q;
        for (int i = tid; i < 2 * width; i += 256) red[i] = 0u;
        __syncthreads();
        if (ro < rows_per_step) {
            float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (part) {
                pv = *reinterpret_cast<const float4*>(x + r0 * width + 4 * cq);
                pv.x = planes_fin(pv.x); pv.y = planes_fin(pv.y); pv.z = planes_fin(pv.z); pv.w = planes_fin(pv.w);
            }
            float4 hi = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f), lo = hi;      // max x, max -x
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f), s1 = m, s2 = m;
            auto take = [&](const float4 v) {
                if (part) {
                    const float a0 = planes_fin(v.x), a1 = planes_fin(v.y), a2 = planes_fin(v.z), a3 = planes_fin(v.w);
                    const bool f0 = fabsf(v.x) <= 3.4028234e38f, f1 = fabsf(v.y) <= 3.4028234e38f, f2 = fabsf(v.z) <= 3.4028234e38f,
                               f3 = fabsf(v.w) <= 3.4028234e38f;
                    hi.x = fmaxf(hi.x, f0 ? a0 : hi.x); hi.y = fmaxf(hi.y, f1 ? a1 : hi.y); hi.z = fmaxf(hi.z, f2 ? a2 : hi.z); hi.w = fmaxf(hi.w, f3 ? a3 : hi.w);
                    lo.x = fmaxf(lo.x, f0 ? -a0 : lo.x); lo.y = fmaxf(lo.y, f1 ? -a1 : lo.y); lo.z = fmaxf(lo.z, f2 ? -a2 : lo.z); lo.w = fmaxf(lo.w, f3 ? -a3 : lo.w);
                    const float d0 = f0 ? a0 - pv.x : 0.f, d1 = f1 ? a1 - pv.y : 0.f, d2 = f2 ? a2 - pv.z : 0.f, d3 = f3 ? a3 - pv.w : 0.f;
                    s1.x += d0; s1.y += d1; s1.z += d2; s1.w += d3;
                    s2.x = fmaf(d0, d0, s2.x); s2.y = fmaf(d1, d1, s2.y); s2.z = fmaf(d2, d2, s2.z); s2.w = fmaf(d3, d3, s2.w);
                } else {
                    m.x = fmaxf(m.x, planes_mag(v.x)); m.y = fmaxf(m.y, planes_mag(v.y));
                    m.z = fmaxf(m.z, planes_mag(v.z)); m.w = fmaxf(m.w, planes_mag(v.w));
                }
            };
            int64_t r = r0 + ro;
            for (; r + 3 * rows_per_step < r1; r += 4 * rows_per_step) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (r + (int64_t)u * rows_per_step) * width + 4 * cq);
#pragma unroll
                for (int u = 0; u < 4; ++u) take(v[u]);
            }
            for (; r < r1; r += rows_per_step) take(*reinterpret_cast<const float4*>(x + r * width + 4 * cq));
            if (part) {
                atomicMax(red + 4 * cq, ord(hi.x)); atomicMax(red + 4 * cq + 1, ord(hi.y)); atomicMax(red + 4 * cq + 2, ord(hi.z)); atomicMax(red + 4 * cq + 3, ord(hi.w));
                atomicMax(red + width + 4 * cq, ord(lo.x)); atomicMax(red + width + 4 * cq + 1, ord(lo.y));
                atomicMax(red + width + 4 * cq + 2, ord(lo.z)); atomicMax(red + width + 4 * cq + 3, ord(lo.w));
                float* sr = sred + (ro * width + 4 * cq) * 2;          // rows_per_step * width * 2 <= 2048 floats
                sr[0] = s1.x; sr[1] = s2.x; sr[2] = s1.y; sr[3] = s2.y; sr[4] = s1.z; sr[5] = s2.z; sr[6] = s1.w; sr[7] = s2.w;
            } else {
                atomicMax(red + 4 * cq, __float_as_uint(m.x)); atomicMax(red + 4 * cq + 1, __float_as_uint(m.y));
                atomicMax(red + 4 * cq + 2, __float_as_uint(m.z)); atomicMax(red + 4 * cq + 3, __float_as_uint(m.w));
            }
        }
        __syncthreads();
        for (int col = tid; col < width; col += 256) {
            if (!part) {
                if (red[col]) atomicMax(mx + col / comps, red[col]);        // non-negative floats order like unsigned
                continue;
            }
            if (red[col]) atomicMax(mx + col, red[col]);                    // (comps == 1 on this path) ordered patterns; 0 = no finite sample
            if (red[width + col]) atomicMax(mx + width + col, red[width + col]);
            float t1 = 0.f, t2 = 0.f;
            for (int k = 0; k < rows_per_step; ++k) { t1 += sred[(k * width + col) * 2]; t2 += sred[(k * width + col) * 2 + 1]; }     // fixed order
            float* o = part + ((int64_t)blockIdx.x * width + col) * 3;
            o[0] = planes_fin(x[r0 * width + col]); o[1] = t1; o[2] = t2;
        }
        return;
    }
    for (int col = tid; col < width; col += 256) {
        float m = 0.f, hi = -3.4e38f, lo = -3.4e38f, t1 = 0.f, t2 = 0.f;
        const float pv = (part && r0 < r1) ? planes_fin(x[r0 * width + col]) : 0.f;
        bool any = false;
        for (int64_t r = r0; r < r1; ++r) {
            const float v = x[r * width + col];
            m = fmaxf(m, planes_mag(v));
            if (fabsf(v) <= 3.4028234e38f) { any = true; hi = fmaxf(hi, v); lo = fmaxf(lo, -v); const float d = v - pv; t1 += d; t2 = fmaf(d, d, t2); }
        }
        if (!part) { if (m > 0.f) atomicMax(mx + col / comps, __float_as_uint(m)); continue; }
        if (any) { atomicMax(mx + col, ord(hi)); atomicMax(mx + width + col, ord(lo)); }
        float* o = part + ((int64_t)blockIdx.x * width + col) * 3;
        o[0] = pv; o[1] = t1; o[2] = t2;
    }
}
// Largest-magnitude form (scales from spectra, and the series form without the quality check): one thread per channel.
__global__ void planes_scale_kernel(const unsigned* mx, int C, float factor, float* scale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float b = __uint_as_float(mx[c]) * factor;
    float s = 1.f;
    if (b > 0.f && b < 3.0e38f) {
        int e;
        (void)frexpf(b, &e);                       // b = f * 2^e, f in [0.5, 1): b <= 2^e
        int k = 15 - e;
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
        s = ldexpf(1.f, k);
    }
    scale[c] = s;
    scale[C + c] = 1.f / s;
}
// Series form with the quality check, one wave per channel.  mx: [C] ordered pattern of max x, [C] of max -x, then one word that
// receives min over the channels of (typical sample magnitude of the channel's quietest block x scale) as the bit pattern of a
// non-negative float.
//   detrend on:  |x - trend| <= 4 (max x - min x): the scale ignores a DC offset; typical magnitude = a block's standard deviation
//   detrend off: |x| <= max |x|; typical magnitude = a block's root mean square
// rows_per_block: rows of x in every block of the statistics kernel (the last one may hold fewer)
__global__ void __launch_bounds__(64) planes_scale_quality_kernel(unsigned* mx, int C, float taper_abs_sum, int detrend, float* scale,
                                                                  const float* part, int n_blocks, int64_t rows_per_block, int64_t n_rows) {
    const int c = blockIdx.x, lane = threadIdx.x;
    auto unord = [](unsigned p) -> float { return __uint_as_float((p & 0x80000000u) ? (p & 0x7fffffffu) : ~p); };
    const unsigned ph = mx[c], pl = mx[C + c];
    const bool any = ph != 0u && pl != 0u;
    const float hi = any ? unord(ph) : 0.f, lo = any ? -unord(pl) : 0.f;          // max x, min x
    const float amp = detrend ? 4.f * (hi - lo) : fmaxf(fabsf(hi), fabsf(lo));
    const float b = amp * taper_abs_sum;
    float s = 1.f;
    if (b > 0.f && b < 3.0e38f) {
        int e;
        (void)frexpf(b, &e);
        int k = 15 - e;
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
        s = ldexpf(1.f, k);
    }
    if (lane == 0) { scale[c] = s; scale[C + c] = 1.f / s; }
    // The typical magnitude is taken from the QUIETEST block of rows (a slab of >= 128 consecutive (time, trial) rows), not from the
    // whole series: an artefact inflates the overall variance with itself and would hide the very stretches it starves of precision;
    // blocks without any variation (flat / zero-padded stretches: their coefficients are exact zeros after the detrend) do not count.
    float quiet = 3.4e38f;
    for (int blk = lane; blk < n_blocks; blk += 64) {
        const int64_t rb = (int64_t)blk * rows_per_block;
        const double nb = (double)((rb + rows_per_block < n_rows ? rb + rows_per_block : n_rows) - rb);
        if (nb < 2.0) continue;
        const float* o = part + ((int64_t)blk * C + c) * 3;
        const double s1 = (double)o[1], s2 = (double)o[2];
        const double var = (s2 - s1 * s1 / nb) / nb, mean = (double)o[0] + s1 / nb;
        if (!(var > 1e-12 * (mean * mean + 1e-300))) continue;           // (no variation beyond the rounding of a constant)
        const float typ = (float)sqrt(detrend ? var : var + mean * mean);
        quiet = fminf(quiet, typ);
    }
    for (int off = 32; off >= 1; off >>= 1) quiet = fminf(quiet, __shfl_down(quiet, off, 64));
    if (lane == 0 && quiet < 3.0e38f && b > 0.f && b < 3.0e38f) atomicMin(mx + 2 * C, __float_as_uint(quiet * s));
}
static int64_t planes_stat_blocks(int64_t n_rows) {
    int64_t blocks = n_rows / 128;
    return blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
}
// bytes of device scratch the quality form needs: max x / max -x per channel, the quality word, the per-block (pivot, s1, s2)
extern "C" int64_t sc_planes_scales_work_bytes(int64_t n_rows, int64_t width) {
    return 4 * ((2 * width + 1 + 3) / 4 * 4) + 4 * 3 * planes_stat_blocks(n_rows) * width;
}
static int planes_scales(const float* d_x, int64_t n_rows, int C, int comps, float factor, float* d_scale, void* d_work, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    unsigned* mx = (unsigned*)d_work;
    SC_CHECK_HIP(hipMemsetAsync(mx, 0, sizeof(unsigned) * C, s));
    hipLaunchKernelGGL(planes_absmax_kernel, dim3((unsigned)planes_stat_blocks(n_rows)), dim3(256), 0, s, d_x, n_rows, C * comps, comps, mx, (float*)nullptr);
    hipLaunchKernelGGL(planes_scale_kernel, dim3((C + 63) / 64), dim3(64), 0, s, mx, C, factor, d_scale);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
// d_work: 4 * n_signals bytes of scratch
extern "C" int sc_planes_scales_from_series_f32(const float* d_x, int64_t T, int64_t R, int64_t C, double taper_abs_sum,
                                                float* d_scale, void* d_work, void* stream) {
    ScTimed timed_("planes_scales", stream);
    SC_REQUIRE(d_x && d_scale && d_work && T >= 1 && R >= 1 && C >= 1 && taper_abs_sum > 0.0, "bad argument");
    return planes_scales(d_x, T * R, (int)C, 1, (float)(8.0 * taper_abs_sum), d_scale, d_work, stream);
}
// The same scan with the quality check of the format (sc_hip.h): scales that ignore a DC offset when stage A detrends, and
// *d_quality (device) = min over the channels of (typical sample magnitude) * scale; d_work: sc_planes_scales_work_bytes(T * R, C)
extern "C" int sc_planes_scales_quality_f32(const float* d_x, int64_t T, int64_t R, int64_t C, int detrend_type, double taper_abs_sum,
                                            float* d_scale, void* d_work, int64_t work_bytes, float* d_quality, void* stream) {
    ScTimed timed_("planes_scales", stream);
    SC_REQUIRE(d_x && d_scale && d_work && d_quality && T >= 1 && R >= 1 && C >= 1 && taper_abs_sum > 0.0, "bad argument");
    SC_REQUIRE(detrend_type >= 0 && detrend_type <= 2, "unknown detrend_type");
    SC_REQUIRE(work_bytes >= sc_planes_scales_work_bytes(T * R, C), "work buffer smaller than sc_planes_scales_work_bytes");
    hipStream_t s = (hipStream_t)stream;
    unsigned* mx = (unsigned*)d_work;
    const int64_t n_rows = T * R, blocks = planes_stat_blocks(n_rows), head = (2 * C + 1 + 3) / 4 * 4;
    float* part = (float*)d_work + head;
    SC_CHECK_HIP(hipMemsetAsync(mx, 0, sizeof(unsigned) * 2 * C, s));
    SC_CHECK_HIP(hipMemsetAsync(mx + 2 * C, 0x7f, sizeof(unsigned), s));            // 0x7f7f7f7f = 3.4e38: "no channel yet"
    hipLaunchKernelGGL(planes_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d_x, n_rows, (int)C, 1, mx, part);
    hipLaunchKernelGGL(planes_scale_quality_kernel, dim3((unsigned)C), dim3(64), 0, s, mx, (int)C, (float)taper_abs_sum,
                       detrend_type != SC_DETREND_NONE ? 1 : 0, d_scale, part, (int)blocks, (n_rows + blocks - 1) / blocks, n_rows);
    SC_CHECK_HIP(hipMemcpyAsync(d_quality, mx + 2 * C, sizeof(float), hipMemcpyDeviceToDevice, s));
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
extern "C" int sc_planes_scales_from_spectra_f32(const void* d_X, int64_t n_rows, int64_t C, float* d_scale, void* d_work, void* stream) {
    SC_REQUIRE(d_X && d_scale && d_work && n_rows >= 1 && C >= 1, "bad argument");
    return planes_scales((const float*)d_X, n_rows, (int)C, 2, 1.0f, d_scale, d_work, stream);          // dense rows of C complex64
}

// ---- conversions between complex64 spectra and the planes format (uploaded coefficients, consumers of complex64) -------
struct PlanesConvArgs {
    const float2* X;
    unsigned char* P;
    const float* scale;       // [C] scale, [C] 1 / scale
    int64_t sF, sW, sR, sK;
    int F, W, R, K, C, nct;
    int64_t n_rows;
};
// one thread per channel pair of a row
__global__ void __launch_bounds__(256) planes_from_spectra_kernel(PlanesConvArgs a) {
    const int pairs_per_row = a.nct * 16;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_rows * pairs_per_row) return;
    const int64_t row = i / pairs_per_row;
    const int pr = (int)(i - row * pairs_per_row), c = 2 * pr;
    int64_t t = row;
    const int k = (int)(t % a.K); t /= a.K;
    const int r = (int)(t % a.R); t /= a.R;
    const int w = (int)(t % a.W); t /= a.W;
    const int f = (int)t;
    const float2* src = a.X + (int64_t)f * a.sF + (int64_t)w * a.sW + (int64_t)r * a.sR + (int64_t)k * a.sK + c;
    float2 v0 = make_float2(0.f, 0.f), v1 = v0;
    if (c < a.C) { const float s0 = a.scale[c]; v0 = src[0]; v0.x *= s0; v0.y *= s0; }
    if (c + 1 < a.C) { const float s1 = a.scale[c + 1]; v1 = src[1]; v1.x *= s1; v1.y *= s1; }
    unsigned* dst = reinterpret_cast<unsigned*>(a.P + row * (int64_t)a.nct * F2_ROW_TILE + (pr >> 4) * F2_ROW_TILE) + (pr & 15);
    unsigned rh, rm, ih, im;
    f2_split2(v0.x, v1.x, rh, rm);
    f2_split2(v0.y, v1.y, ih, im);
    dst[0] = rh; dst[16] = rm; dst[32] = ih; dst[48] = im;
}
__global__ void __launch_bounds__(256) spectra_from_planes_kernel(PlanesConvArgs a) {
    const int pairs_per_row = a.nct * 16;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_rows * pairs_per_row) return;
    const int64_t row = i / pairs_per_row;
    const int pr = (int)(i - row * pairs_per_row), c = 2 * pr;
    if (c >= a.C) return;
    int64_t t = row;
    const int k = (int)(t % a.K); t /= a.K;
    const int r = (int)(t % a.R); t /= a.R;
    const int w = (int)(t % a.W); t /= a.W;
    const int f = (int)t;
    const unsigned* src = reinterpret_cast<const unsigned*>(a.P + row * (int64_t)a.nct * F2_ROW_TILE + (pr >> 4) * F2_ROW_TILE) + (pr & 15);
    const unsigned rh = src[0], rm = src[16], ih = src[32], im = src[48];
    const float i0 = a.scale[a.C + c], i1 = (c + 1 < a.C) ? a.scale[a.C + c + 1] : 1.f;
    const float2 v0 = make_float2((f2_lo(rh) + f2_lo(rm)) * i0, (f2_lo(ih) + f2_lo(im)) * i0);      // h + m is exact in f32
    const float2 v1 = make_float2((f2_hi(rh) + f2_hi(rm)) * i1, (f2_hi(ih) + f2_hi(im)) * i1);
    float2* dst = const_cast<float2*>(a.X) + (int64_t)f * a.sF + (int64_t)w * a.sW + (int64_t)r * a.sR + (int64_t)k * a.sK + c;
    dst[0] = v0;
    if (c + 1 < a.C) dst[1] = v1;
}
static int planes_conv(const void* d_X, const sc_spectra_desc* d, const float* d_scale, void* d_P, bool to_planes, void* stream) {
    SC_REQUIRE(d_X && d && d_P && d_scale, "NULL argument");
    SC_REQUIRE(d->n_freq >= 1 && d->n_windows >= 1 && d->n_trials >= 1 && d->n_tapers >= 1 && d->n_signals >= 1, "empty dimension");
    PlanesConvArgs a;
    a.X = (const float2*)d_X; a.P = (unsigned char*)d_P; a.scale = d_scale;
    a.sF = d->stride_freq; a.sW = d->stride_window; a.sR = d->stride_trial; a.sK = d->stride_taper;
    a.F = (int)d->n_freq; a.W = (int)d->n_windows; a.R = (int)d->n_trials; a.K = (int)d->n_tapers; a.C = (int)d->n_signals;
    a.nct = (a.C + 31) / 32;
    a.n_rows = (int64_t)a.F * a.W * a.R * a.K;
    const int64_t items = a.n_rows * a.nct * 16;
    const unsigned blocks = (unsigned)((items + 255) / 256);
    if (to_planes) hipLaunchKernelGGL(planes_from_spectra_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(spectra_from_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
extern "C" int sc_planes_from_spectra_f32(const void* d_X, const sc_spectra_desc* desc, const float* d_scale, void* d_P, void* stream) {
    ScTimed timed_("planes_from_spectra", stream);
    return planes_conv(d_X, desc, d_scale, d_P, true, stream);
}
extern "C" int sc_spectra_from_planes_f32(const void* d_P, const sc_spectra_desc* desc, const float* d_scale, void* d_X, void* stream) {
    ScTimed timed_("spectra_from_planes", stream);
    return planes_conv(d_X, desc, d_scale, const_cast<void*>(d_P), false, stream);
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------
struct Fused2Args {
    FusedArgs f;                  // record, map, tiles of the CSM waves, split; f.st.n_obs / f.st.ax as in sc_fused.hip (strides in ROWS)
    const unsigned char* P;       // planes-format spectra, dense rows [F][W][R][K]
    const float* inv_scale;       // [C] 1 / scale of the record's channels
    int64_t row_bytes;            // bytes per observation row (256 per 32-channel tile)
    int64_t obs_rows;             // rows between consecutive observations of a bin (linear: checked on the host)
    int loader_csm;               // 1: the CSM waves issue the HBM -> LDS loads (8 each), 0: the |Im s| waves (4 each)
    int terms4;                   // 1: the CSM products keep the m m term (few observations per bin: its error does not average out)
    int fold_obs;                 // observations a tile's f32 accumulators take in before they are folded into the record
};

__device__ __forceinline__ h16x4 f2_tr(lds_u8* base, int off) {
    return __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off)));
}
// operand fragment of v_mfma_f32_16x16x32_f16 (8 K slots per lane): plane p, observations
// 8 (2 (g >> 1) + t) + 4 (g & 1) + r  for the two loads t = 0, 1 (g = lane >> 4, r = (lane >> 2) & 3, baked into `base`)
__device__ __forceinline__ h16x8 f2_frag(lds_u8* base, int p) {
    const h16x4 a = f2_tr(base, p * F2_PLANE), b = f2_tr(base, p * F2_PLANE + 256);
    const h16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return v;
}

#define F2_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
// Every wave has waited for its own HBM -> LDS loads (vmcnt) before it gets here; the barrier itself publishes LDS writes
// only (a __syncthreads() would also wait for the fold atomics the CSM waves have just sent to L2: ~1 us per chunk).
#define F2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Loads of one chunk (32 instructions of 1 KB): the eight |Im s| waves (4 .. 11; no other memory traffic of theirs is in
// flight, so their vmcnt counts these loads exactly) fill plane (w - 4) % 4 of the observation groups o7 = 4 ((w - 4) / 4) .. + 3.
// Lane l of an instruction carries observation o7 + 8 (l >> 4), 16-byte piece l & 15 of the 256-byte row (channel tile
// (l & 15) >> 2).
struct F2Loader {
    const unsigned char* src;     // first row of this (bin, part), + the plane offset of this wave (wave-uniform)
    unsigned voff;                // per-lane byte offset inside a chunk (the only per-lane state kept across the chunk loop)
    int lds_off;                  // LDS offset of this wave's first group (plane, o7) inside a buffer (wave-uniform)
    int o7_0, plane;              // wave-uniform; plane < 0: this wave loads nothing
    int n_inst;                   // observation groups (instructions) per chunk: 4 (eight loader waves) or 8 (four)
};
// NHALF = 2 (one or two staged 32-channel blocks): a chunk is 64 observations -- observation 32 + o of the chunk sits in the channel
// slots 64 .. 127 of observation o's row (the half of every 256-byte row slot that 64 channels leave empty), so twice the bytes are
// in flight per buffer and every barrier covers twice the work.  Both roles then run their body twice per chunk, the second time
// 128 bytes further into the rows.  (Measured before: at 64 channels the loads cost 0.8 of 3.1 ms -- one workgroup per CU with two
// 16 KB chunks in flight cannot cover the HBM latency.)
template <int NHALF>
__device__ __forceinline__ F2Loader f2_loader(const Fused2Args& a, int wave, const unsigned char* part_base) {
    F2Loader L;
    const int lane = fu_lane();
    const int piece = lane & 15, ct = NHALF == 2 ? ((piece >> 2) & 1) : (piece >> 2), half = NHALF == 2 ? (piece >> 3) : 0;
    if (a.loader_csm) {          // the four CSM waves load (they have the slack: half the matrix work of the first form)
        L.plane = wave < 4 ? wave : -1;
        L.o7_0 = 0;
        L.n_inst = 8;
    } else {
        L.plane = wave >= 4 ? ((wave - 4) & 3) : -1;
        L.o7_0 = wave >= 4 ? 4 * ((wave - 4) >> 2) : 0;
        L.n_inst = 4;
    }
    L.voff = (unsigned)((8 * (lane >> 4) + FU_OC * half) * a.obs_rows * a.row_bytes + fu_byte(a.f.map.off32, ct) * F2_ROW_TILE + (piece & 3) * 16);
    L.src = part_base + (L.plane < 0 ? 0 : L.plane) * 64;
    // LDS planes: Re h m -> 0 1, Im h m -> 2 3 (-Re h m -> 4 5 are made in f2_finish)
    L.lds_off = (L.plane < 0 ? 0 : L.plane) * F2_PLANE + L.o7_0 * F2_GROUP;
    return L;
}
// o0 = first observation of the chunk (relative to the part's first), n_left = observations left in the part from o0
template <int NHALF>
__device__ __forceinline__ void f2_issue(const Fused2Args& a, const F2Loader& L, lds_u8* buf, int o0, int n_left) {
    if (L.plane < 0) return;
    const int lane = fu_lane();                                   // (re-materialised: short live ranges, see fu_lane)
    const int ct = NHALF == 2 ? ((lane >> 2) & 1) : ((lane & 15) >> 2);
    const bool lane_ok = ct < a.f.NB32 && fu_byte(a.f.map.n32, ct) > 0;      // this lane's channel tile is staged
    const int o_first = L.o7_0 + 8 * (lane >> 4) + (NHALF == 2 ? FU_OC * ((lane >> 3) & 1) : 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= L.n_inst) break;
        lds_u8* dst = buf + L.lds_off + i * F2_GROUP;             // wave-uniform
        const unsigned char* src = L.src + (int64_t)(o0 + L.o7_0 + i) * a.obs_rows * a.row_bytes;   // wave-uniform
        // Spelled in asm: behind the builtin the compiler's wait-count pass puts s_waitcnt vmcnt(0) in front of the next LDS
        // read of ANY address (it cannot tell the other buffer from this one), i.e. the wave would sit out the loads it has
        // just issued -- measured: load time and product time simply added up.  The waves wait for their own loads
        // explicitly (vmcnt) before f2_finish and the barrier that publishes the buffer.
        if (lane_ok && o_first + i < n_left)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"((unsigned)reinterpret_cast<uintptr_t>(dst)), "v"(L.voff), "s"(src) : "memory", "m0");
    }
}
// After the wave's loads have landed (vmcnt): rows past the end of the part become zeros, and the waves that loaded
// real-part planes write the negated copies (planes 4, 5) -- one xor per 8 coefficients.
template <int NHALF>
__device__ __forceinline__ void f2_finish(const F2Loader& L, lds_u8* buf, int n_left) {
    if (L.plane < 0) return;
    const int lane = fu_lane();
    if (n_left < FU_OC * NHALF) {                                  // wave-uniform: the last chunk of a part only
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i >= L.n_inst) break;
            const int o = L.o7_0 + i + 8 * (lane >> 4) + (NHALF == 2 ? FU_OC * ((lane >> 3) & 1) : 0);
            if (o >= n_left)
                *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(buf + L.lds_off + i * F2_GROUP + 16 * lane) = (u32x4){0u, 0u, 0u, 0u};
        }
    }
    if (L.plane < 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i >= L.n_inst) break;
            lds_u8* g = buf + L.lds_off + i * F2_GROUP + 16 * lane;
            u32x4 v = *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(g);
            v[0] ^= 0x80008000u; v[1] ^= 0x80008000u; v[2] ^= 0x80008000u; v[3] ^= 0x80008000u;
            *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(g + 4 * F2_PLANE) = v;
        }
    }
}

// instructions f2_issue really sends for a chunk with n_left observations (an instruction whose lanes are all past the end
// is skipped): what vmcnt has to leave outstanding when the chunk BEHIND the awaited one is in flight too
__device__ __forceinline__ int f2_issued(const F2Loader& L, int n_left) {
    if (L.plane < 0) return 0;
    const int k = n_left - L.o7_0;
    return k < 0 ? 0 : (k > L.n_inst ? L.n_inst : k);
}
__device__ __forceinline__ void f2_wait_loads(int outstanding) {
    switch (outstanding) {
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

static_assert(2 * 4 + 1 <= FU_FLUSH, "one fold slot per tile of a wave");
// Matrix-core role: wave w owns the tiles fu_assign_rows gave it (tile rows w and R - 1 - w of the triangle).  Per tile and
// chunk 12 v_mfma_f32_16x16x32_f16: the cross terms h h, h m, m h x {Re Re, Im Im, Im Re, (-Re) Im} (16 with the m m term);
// every operand fragment is two transposing LDS loads, no VALU.
template <int NB32>
__device__ __forceinline__ void f2_mfma_role(const Fused2Args& a, lds_u8* lds, const unsigned char* lds_generic, const F2Loader& L,
                                             int wave, float* rec, int n_part) {
    const FusedArgs& p = a.f;
    constexpr int MAXS = 2 * NB32 + 1;
    constexpr int NHALF = NB32 <= 2 ? 2 : 1, CH = FU_OC * NHALF;
    const int FLUSH = a.fold_obs / CH;           // chunks between the folds of a tile's accumulators into the record (fused2_run)
    const unsigned sg = (unsigned)__builtin_amdgcn_readfirstlane((int)(wave == 0 ? p.seg0 : (wave == 1 ? p.seg1 : (wave == 2 ? p.seg2 : p.seg3))));
    const unsigned sg_n = (unsigned)__builtin_amdgcn_readfirstlane((int)((p.seg_n >> (8 * wave)) & 0xffu));
    const int rA_ = sg & 0xf, cA_ = (sg >> 4) & 0xf, rB_ = (sg >> 8) & 0xf, cB_ = (sg >> 12) & 0xf;
    const int nA_ = sg_n & 0xf, nB_ = sg_n >> 4;
    const int total = nA_ + nB_;
    f32x4 re[MAXS], im[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) { re[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; im[s] = re[s]; }
    const int n_chunks = (n_part + CH - 1) / CH;
    float* out = rec + (int64_t)p.csm_plane * p.n_tiles * SC_TILE_ELEMS;
    const bool do_csm = p.csm_plane >= 0 && (p.debug_skip & 1) == 0;
    const bool loads = !(p.debug_skip & 8);
    const bool t4 = a.terms4 != 0;
    for (int ch = 0; ch < n_chunks; ++ch) {
        lds_u8* cur = lds + (loads ? (ch % F2_NBUF) : 0) * F2_BUF;
        const bool more = ch + 1 < n_chunks && loads, more2 = ch + 2 < n_chunks && loads;
        if (more2) f2_issue<NHALF>(a, L, lds + ((ch + 2) % F2_NBUF) * F2_BUF, (ch + 2) * CH, n_part - (ch + 2) * CH);
        if (do_csm && total > 0) {
            int rA = rA_, rB = rB_, nA = nA_, cA = cA_, cB = cB_;
            asm volatile("" : "+s"(rA), "+s"(rB), "+s"(nA), "+s"(cA), "+s"(cB));
            const int lane = fu_lane(), g = lane >> 4;
            lds_u8* b00 = cur + ((4 * (g & 1) + ((lane >> 2) & 3)) * F2_GROUP + 2 * (g >> 1) * 256 + (lane & 3) * 8);
            h16x8 arh, arm, aih, aim, nrh, nrm;
#pragma unroll
            for (int hv = 0; hv < NHALF; ++hv) {          // (NHALF == 2: observations 32 .. 63 of the chunk, 128 bytes into the rows)
            lds_u8* b0 = b00 + hv * 128;
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                if (s < total) {
                    const bool in_a = s < nA;
                    const int row = in_a ? rA : rB;
                    const int col = in_a ? cA + s : cB + (s - nA);
                    if (s == 0 || s == nA) {
                        lds_u8* fa = b0 + row * 32;
                        arh = f2_frag(fa, 0); arm = f2_frag(fa, 1); aih = f2_frag(fa, 2); aim = f2_frag(fa, 3);
                        nrh = f2_frag(fa, 4); nrm = f2_frag(fa, 5);
                    }
                    lds_u8* fb = b0 + col * 32;
                    const h16x8 brh = f2_frag(fb, 0), bih = f2_frag(fb, 2), brm = f2_frag(fb, 1), bim = f2_frag(fb, 3);
                    // Re += ar*br + ai*bi ; Im += ai*br + (-ar)*bi   for the piece pairs h h, h m, m h (, m m)
                    F2_MFMA(arh, brh, re[s]);  F2_MFMA(aih, brh, im[s]);
                    F2_MFMA(aih, bih, re[s]);  F2_MFMA(nrh, bih, im[s]);
                    F2_MFMA(arh, brm, re[s]);  F2_MFMA(aih, brm, im[s]);
                    F2_MFMA(aih, bim, re[s]);  F2_MFMA(nrh, bim, im[s]);
                    F2_MFMA(arm, brh, re[s]);  F2_MFMA(aim, brh, im[s]);
                    F2_MFMA(aim, bih, re[s]);  F2_MFMA(nrm, bih, im[s]);
                    if (t4) {
                        F2_MFMA(arm, brm, re[s]);  F2_MFMA(aim, brm, im[s]);
                        F2_MFMA(aim, bim, re[s]);  F2_MFMA(nrm, bim, im[s]);
                    }
                }
            }
            }
        }
        if (more && L.plane >= 0) {
            // Chunk ch + 1 has landed once only the loads of chunk ch + 2 are outstanding.  (The fold atomics of the chunk
            // before sit between the two in this wave's queue: loads return in order, so "at most the loads of ch + 2
            // outstanding" still means every load of ch + 1 is back; the atomics are a chunk old by now.)
            f2_wait_loads(more2 ? f2_issued(L, n_part - (ch + 2) * CH) : 0);
            f2_finish<NHALF>(L, lds + ((ch + 1) % F2_NBUF) * F2_BUF, n_part - (ch + 1) * CH);
        }
        // two-level summation as in sc_fused.hip: a tile's accumulators are folded into the record every FLUSH chunks (512
        // observations: fused2_run), the tiles taking turns; the channel scales (powers of two) come out here, exactly
        {
            const bool last = ch + 1 == n_chunks;
#pragma unroll
            for (int s = 0; s < MAXS; ++s) {
                const int f_s = FLUSH - 1 - s;
                const bool due = ((ch + 1 + s) % FLUSH) == 0 && !(p.debug_skip & 64);
                if (s < total && (due || last) && p.csm_plane >= 0) {
                    const bool first = ch <= f_s || (p.debug_skip & 64);
                    const bool in_a = s < nA_;
                    const int row = in_a ? rA_ : rB_;
                    const int col = in_a ? cA_ + s : cB_ + (s - nA_);
                    float* o_re = out + (int64_t)sc_tile_index(fu_gt(p.map, row), fu_gt(p.map, col), p.map.NBr) * SC_TILE_ELEMS;
                    float* o_im = o_re + (int64_t)p.n_tiles * SC_TILE_ELEMS;
                    const unsigned fl = (unsigned)fu_lane();
                    const unsigned base_idx = (fl >> 4) * 64u + (fl & 15u);
                    // this lane's four entries: rows 4 (lane >> 4) + r, column lane & 15 of the tile
                    const float* isc = reinterpret_cast<const float*>((const unsigned char*)lds_generic + F2_SCALE_OFF);
                    const float sj = isc[col * 16 + (int)(fl & 15u)];
                    const f32x4 si = *reinterpret_cast<const f32x4*>(isc + row * 16 + 4 * (int)(fl >> 4));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned idx = base_idx + 16u * r;
                        const float sc = sj * si[r];
                        if (first) {
                            o_re[idx] = re[s][r] * sc;
                            o_im[idx] = im[s][r] * sc;
                        } else {
                            unsafeAtomicAdd(o_re + idx, re[s][r] * sc);
                            unsafeAtomicAdd(o_im + idx, im[s][r] * sc);
                        }
                    }
                    re[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; im[s] = re[s];
                }
            }
        }
        F2_BARRIER();
    }
    const int wps = 8 / p.n_sets;
    for (int half = wps >> 1; half >= 1; half >>= 1) { __syncthreads(); __syncthreads(); }
}

// |Im s| role: per observation row and 32x32 channel block ONE v_mfma_f32_32x32x8_f16 with C = 0 (K = 8 slots: lanes
// 0-31 carry the four cross terms h.h h.m m.h m.m of Im x_i Re x_j, lanes 32-63 those of (-Re x_i) Im x_j), then
// acc += |d|.  An operand is ONE transposing load whose four rows are planes of one observation:
//   A (row channel i):  planes [h h m m] of Im (lanes 0-31) / -Re (lanes 32-63)
//   B (col channel j):  planes [h m h m] of Re (lanes 0-31) /  Im (lanes 32-63)
template <int NB32, int COL_LO, int ROW_HI, int SET, int OP, int RPW>
__device__ __forceinline__ void f2_valu_body(const Fused2Args& a, lds_u8* lds, unsigned char* smem, const F2Loader& L, int tid,
                                             int rsub, int wps, float* rec, int n_part) {
    const FusedArgs& p = a.f;
    using Tab = FuTab<NB32, COL_LO, ROW_HI, SET>;
    constexpr int NBLK = Tab::NBLK;
    // packed squares (v_pk_fma_f32 on aligned register pairs) only for single-set shapes: with two sets the pairs do not fit
    constexpr bool PK = NBLK <= 4 && fu_nsets(NB32, COL_LO, ROW_HI) == 1;
    const int lane = tid & 63;
    f32x16 acc[NBLK > 0 ? NBLK : 1];
#pragma unroll
    for (int s = 0; s < NBLK; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
    constexpr int NHALF = NB32 <= 2 ? 2 : 1, CH = FU_OC * NHALF;
    const int n_chunks = (n_part + CH - 1) / CH;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool loads = !(p.debug_skip & 8);
    const bool compute = !(p.debug_skip & 2) && p.abs_plane >= 0;
    constexpr int rpw = RPW;                       // observation rows of an 8-row group this wave takes (1 or 2)
    for (int ch = 0; ch < n_chunks; ++ch) {
        lds_u8* cur = lds + (loads ? (ch % F2_NBUF) : 0) * F2_BUF;
        lds_u8* nxt = lds + ((ch + 1) % F2_NBUF) * F2_BUF;
        const bool more = ch + 1 < n_chunks && loads, more2 = ch + 2 < n_chunks && loads;
        // chunk ch + 2 into the buffer chunk ch - 1 was multiplied from (every wave is past that chunk's barrier)
        if (more2) f2_issue<NHALF>(a, L, lds + ((ch + 2) % F2_NBUF) * F2_BUF, (ch + 2) * CH, n_part - (ch + 2) * CH);
        if (compute) {
            const int cl = fu_lane(), hf = cl >> 5, r = (cl >> 2) & 3;
            const int pA = hf ? (r < 2 ? 4 : 5) : (r < 2 ? 2 : 3);
            const int pB = hf ? ((r & 1) ? 3 : 2) : ((r & 1) ? 1 : 0);
            const int common = rsub * rpw * F2_GROUP + ((cl >> 4) & 1) * 32 + (cl & 3) * 8;
#pragma unroll
            for (int hv = 0; hv < NHALF; ++hv) {          // (NHALF == 2: observations 32 .. 63 of the chunk, 128 bytes into the rows)
            lds_u8* bA = cur + (pA * F2_PLANE + common + hv * 128);
            lds_u8* bB = cur + (pB * F2_PLANE + common + hv * 128);
            // zero rows past the end of the part contribute |0| = 0: no bound needed
            // The wave's rows t = 0 .. 4 rpw - 1 (observation 8 (t / rpw) + rsub rpw + t % rpw): the operand fragments of row
            // t + 1 are requested before row t is multiplied, so no row starts with a wait for the LDS.
            constexpr int NR = 4 * RPW;
            h16x4 FA[2][4], FB[2][4];
            auto fetch = [&](int t, int w) {
                const int ro = (t % RPW) * F2_GROUP + (t / RPW) * 256;
#pragma unroll
                for (int b = 0; b < NB32; ++b) {
                    if (Tab::tab.use_i[b]) FA[w][b] = f2_tr(bA, ro + b * 64);
                    if (Tab::tab.use_j[b]) FB[w][b] = f2_tr(bB, ro + b * 64);
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const int w = t & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < NR) fetch(t + 1, w ^ 1);
                f32x16 dprev;
#pragma unroll
                for (int s = 0; s < NBLK; ++s) {
                    const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x8f16(FA[w][Tab::tab.bi[s]], FB[w][Tab::tab.bj[s]], zero, 0, 0, 0);
                    // software pipeline: the accumulation of block s - 1 is issued AFTER the MFMA of block s
                    __builtin_amdgcn_sched_barrier(0);
                    if (s > 0) fu_accumulate16<OP, PK>(acc[s - 1], dprev);
                    dprev = d;
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (OP == FU_OP_SIGN && NBLK == 1) {
                    // A lone block per row: the asm form of the sign accumulate (fu_accumulate16) would sit straight behind its
                    // own MFMA, where only instructions the compiler can see get the wait states they need -- measured: wrong
                    // sign sums from the second row of a wave on (<= 32 channels, > 8 observations).  Plain code here: one block
                    // leaves the registers for it.
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[0][e] = fu_accumulate<OP>(acc[0][e], dprev[e]);
                } else {
                    fu_accumulate16<OP, PK>(acc[NBLK - 1], dprev);
                }
            }
            }
        }
        if (more) {       // chunk ch + 1 has landed once only the loads of chunk ch + 2 are outstanding
            f2_wait_loads(more2 ? f2_issued(L, n_part - (ch + 2) * CH) : 0);
            f2_finish<NHALF>(L, nxt, n_part - (ch + 1) * CH);
        }
        F2_BARRIER();
    }
    if constexpr (OP == FU_OP_SIGN) {
#pragma unroll
        for (int s = 0; s < NBLK; ++s)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][e] = (float)__float_as_int(acc[s][e]);
    }
    // tree-sum the row-split partials of a set through LDS (20 KB per writer), as in sc_fused.hip
    float* red = reinterpret_cast<float*>(smem);
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
        // D layout of the 32x32 MFMAs: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        float* out = rec + (int64_t)p.abs_plane * p.n_tiles * SC_TILE_ELEMS;
#pragma unroll
        for (int s = 0; s < NBLK; ++s) {
            const int BIs = Tab::tab.bi[s], BJs = Tab::tab.bj[s];
            const int j = BJs * 32 + i32, tj = j >> 4;
            const float* isc = reinterpret_cast<const float*>(smem + F2_SCALE_OFF);
            const float sj = (OP == FU_OP_SIGN) ? 1.f : isc[j];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = BIs * 32 + (e & 3) + 8 * (e >> 2) + 4 * hf;
                const int ti = i >> 4;
                if (ti <= tj && fu_tile_ok(p.map, ti) && fu_tile_ok(p.map, tj)) {
                    float sc = (OP == FU_OP_SIGN) ? 1.f : sj * isc[i];
                    if (OP == FU_OP_SQ) sc *= sc;
                    out[(int64_t)sc_tile_index(fu_gt(p.map, ti), fu_gt(p.map, tj), p.map.NBr) * SC_TILE_ELEMS + (i & 15) * 16 +
                        (j & 15)] = acc[s][e] * sc;
                }
            }
        }
    }
}

// Sustained shader clock of the last launch: wave 0 of three workgroups (first, middle, last) reads the shader-cycle counter
// (s_memtime) and the constant 100 MHz counter (s_memrealtime) when it starts and when it ends; cycles / real time over a
// workgroup's life (~0.35 ms at cfg3) is the clock the part sustains under this kernel's load (sc_debug_fused2_clock).
__device__ unsigned long long f2_clock_buf[3 * 4];
__device__ __forceinline__ void f2_clock_stamp(int slot, int which) {
    unsigned long long c, r;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r));
    f2_clock_buf[slot * 4 + 2 * which] = c;
    f2_clock_buf[slot * 4 + 2 * which + 1] = r;
}
extern "C" int sc_debug_fused2_clock(double* ghz) {
    unsigned long long h[12];
    SC_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(f2_clock_buf), sizeof h));
    double sum = 0.0;
    int n = 0;
    for (int s = 0; s < 3; ++s) {
        const double cyc = (double)(h[s * 4 + 2] - h[s * 4]), ticks = (double)(h[s * 4 + 3] - h[s * 4 + 1]);
        if (ticks > 0 && cyc > 0) { sum += cyc / (ticks / 100.0e6) * 1e-9; ++n; }
    }
    if (ghz) *ghz = n ? sum / n : 0.0;
    return SC_OK;
}

template <int NB32, int COL_LO, int ROW_HI, int OP>
__global__ void __launch_bounds__(FU_THREADS) fused2_kernel(Fused2Args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const FusedArgs& p = a.f;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int clock_slot = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1));
    if (clock_slot >= 0 && tid == 0) f2_clock_stamp(clock_slot, 0);
    const int bin = blockIdx.x / p.n_split, part = blockIdx.x - bin * p.n_split;
    const int g = bin / p.F, f = bin - g * p.F;
    const int nc = (p.st.n_obs + FU_OC - 1) / FU_OC;
    const int o_lo = (int)((int64_t)part * nc / p.n_split) * FU_OC;
    int o_hi = (int)((int64_t)(part + 1) * nc / p.n_split) * FU_OC;
    if (o_hi > p.st.n_obs) o_hi = p.st.n_obs;
    const int n_part = o_hi - o_lo;
    float* rec = (part == 0 ? p.accum : p.ws + (int64_t)(part - 1) * p.n_bins * p.floats_per_bin) + (int64_t)bin * p.floats_per_bin;
    const unsigned char* part_base = a.P + ((int64_t)f * p.st.ax.sF + sc_group_offset(p.st.ax, g) + (int64_t)o_lo * a.obs_rows) * a.row_bytes;
    lds_u8* lds = (lds_u8*)smem;
    constexpr int NHALF = NB32 <= 2 ? 2 : 1, CH = FU_OC * NHALF;
    const F2Loader L = f2_loader<NHALF>(a, wave, part_base);
    // slots of channel tiles that are not staged stay zero for good
    for (int i = tid * 16; i < F2_SCALE_OFF; i += FU_THREADS * 16)
        *reinterpret_cast<u32x4*>(smem + i) = (u32x4){0u, 0u, 0u, 0u};
    if (tid < 128) {        // staged channel slot -> reciprocal scale of the record's channel (folds read it from LDS)
        const int gch = fu_byte(p.map.off32, tid >> 5) * 32 + (tid & 31);
        reinterpret_cast<float*>(smem + F2_SCALE_OFF)[tid] = ((tid >> 5) < p.NB32 && (tid & 31) < fu_byte(p.map.n32, tid >> 5) && gch < p.st.C)
                                                                 ? a.inv_scale[gch] : 1.f;
    }
    __syncthreads();
    f2_issue<NHALF>(a, L, lds, 0, n_part);
    const bool second = n_part > CH && !(p.debug_skip & 8);
    if (second) f2_issue<NHALF>(a, L, lds + F2_BUF, CH, n_part - CH);
    f2_wait_loads(second ? f2_issued(L, n_part - CH) : 0);
    f2_finish<NHALF>(L, lds, n_part);
    F2_BARRIER();
    if (wave < 4) {
        if (p.debug_skip & 32) __builtin_amdgcn_s_setprio(2);
        f2_mfma_role<NB32>(a, lds, smem, L, wave, rec, n_part);
        if (clock_slot >= 0 && tid == 0) f2_clock_stamp(clock_slot, 1);
    } else {
        if (p.debug_skip & 16) __builtin_amdgcn_s_setprio(2);
        constexpr int NSETS = fu_nsets(NB32, COL_LO, ROW_HI);
        constexpr int wps = 8 / NSETS;
        const int vw = wave - 4, set = vw / wps, rsub = vw % wps;
        if constexpr (NSETS == 1) {
            f2_valu_body<NB32, COL_LO, ROW_HI, 0, OP, 8 / wps>(a, lds, smem, L, tid, rsub, wps, rec, n_part);
        } else {
            if (set == 0) f2_valu_body<NB32, COL_LO, ROW_HI, 0, OP, 8 / wps>(a, lds, smem, L, tid, rsub, wps, rec, n_part);
            else f2_valu_body<NB32, COL_LO, ROW_HI, 1, OP, 8 / wps>(a, lds, smem, L, tid, rsub, wps, rec, n_part);
        }
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
// What the planes-format kernels fill: the CSM planes, |Im s| with them, (Im s)^2 with both (a second pass of the same kernel:
// the |Im s| waves square the per-observation products instead), and sign(Im s) on its own (a pass that sums signs as
// integers) -- up to 1024 signals (129 ... 256: the launches of sc_fused.hip's launch_fused_all, each staging four 32-channel
// blocks; beyond: the general plan of fused2_launch_all), observations of a bin one linear run of rows.
#define F2_MAX_SIGNALS 1024      // 32 blocks of 32 channels (the block map packs a block's 16-tile row into 8 bits: 127 blocks at most)
static bool fused2_families_ok(uint32_t fam) {
    return fam == SC_PLANE_CSM || fam == (SC_PLANE_CSM | SC_PLANE_ABS_IM) ||
           fam == (SC_PLANE_CSM | SC_PLANE_ABS_IM | SC_PLANE_IM_SQ) || fam == SC_PLANE_SIGN_IM;
}
static int fused2_setup(const sc_spectra_desc* desc, uint32_t planes, Fused2Args* out, ScAxes* ax_out) {
    SC_REQUIRE(desc, "NULL argument");
    sc_spectra_desc d = *desc;                 // rows of the planes buffer are dense [F][W][R][K]: strides in rows
    d.stride_taper = 1; d.stride_trial = d.n_tapers; d.stride_window = d.n_trials * d.n_tapers;
    d.stride_freq = d.n_windows * d.n_trials * d.n_tapers;
    ScAxes ax;
    sc_make_axes(&d, &ax);
    SC_REQUIRE(ax.C >= 1 && ax.F >= 1 && ax.n_obs >= 1 && ax.n_groups >= 1, "empty dimension");
    const uint32_t fam = planes & ~(uint32_t)SC_RECORD_F64;
    if (!fused2_families_ok(fam) || (planes & SC_RECORD_F64) || ax.C > F2_MAX_SIGNALS || sc_stage_linear_stride(ax) <= 0) {
        sc_set_error("planes-format stage B takes CSM (+ |Im s| (+ (Im s)^2)) or sign(Im s) records of up to 1024 signals whose "
                     "observations are one linear run (got planes 0x%x, %d signals)", planes, ax.C);
        return SC_EUNSUPPORTED;
    }
    Fused2Args& a = *out;
    FusedArgs& f = a.f;
    f.st.base = nullptr; f.st.ax = ax; f.st.obs_stride = sc_stage_linear_stride(ax); f.st.C = ax.C; f.st.n_obs = ax.n_obs;
    f.NB32 = (ax.C + 31) / 32;                  // of the whole record; a launch stages up to four (fused2_args_blocks)
    f.st.CP = f.NB32 * 32; f.st.RS = 0;
    f.NB = sc_n_blocks(ax.C);
    f.n_tiles = sc_n_tiles(f.NB);
    f.n_bins = ax.n_groups * ax.F;
    f.F = ax.F;
    f.floats_per_bin = (int64_t)sc_plane_count(planes) * f.n_tiles * SC_TILE_ELEMS;
    f.csm_plane = (planes & SC_PLANE_CSM) ? sc_plane_offset(planes, SC_PLANE_CSM) : -1;
    f.abs_plane = (planes & SC_PLANE_ABS_IM) ? sc_plane_offset(planes, SC_PLANE_ABS_IM) : -1;
    f.sq_plane = (planes & SC_PLANE_IM_SQ) ? sc_plane_offset(planes, SC_PLANE_IM_SQ) : -1;
    f.sign_plane = (planes & SC_PLANE_SIGN_IM) ? sc_plane_offset(planes, SC_PLANE_SIGN_IM) : -1;
    f.n_fold = 0;
    f.nl_op = FU_OP_ABS;
    f.n_split = 1; f.ws = nullptr; f.debug_skip = 0;
    f.map.NBr = f.NB;
    a.row_bytes = sc_planes_row_bytes(ax.C);
    a.obs_rows = f.st.obs_stride;
    a.terms4 = ax.n_obs < 256 ? 1 : 0;
    a.fold_obs = 512;
    a.loader_csm = 1;
    *ax_out = ax;
    return SC_OK;
}

// One launch that stages the nb (<= 4) 32-channel blocks `blocks` (ascending block numbers of the record's channels) and owns
// the products (bi <= bj, bj >= col_lo, bi < row_hi) of them (same scheme as fu_args_blocks of sc_fused.hip; the loader takes a
// block's rows from tile off32 of the observation row).
static Fused2Args fused2_args_blocks(const Fused2Args& full, const int* blocks, int nb, int col_lo, int row_hi) {
    Fused2Args a = full;
    FusedArgs& f = a.f;
    const int C = full.f.st.C;
    int n_last = 0;
    f.map.off32 = f.map.n32 = f.map.t32 = 0u;
    for (int b = 0; b < nb; ++b) {
        const int c = blocks[b] * 32;
        n_last = C - c < 32 ? C - c : 32;
        f.map.off32 |= (unsigned)blocks[b] << (8 * b);
        f.map.n32 |= (unsigned)n_last << (8 * b);
        f.map.t32 |= (unsigned)(blocks[b] * 2) << (8 * b);
    }
    f.NB32 = nb;
    f.NB = 2 * (nb - 1) + (n_last + 15) / 16;      // 16-channel tiles that exist among the staged blocks
    f.shape_col_lo = col_lo;
    f.shape_row_hi = row_hi;
    f.n_blocks32 = fu_nblocks(nb, col_lo, row_hi);
    f.n_sets = fu_nsets(nb, col_lo, row_hi);
    f.st.CP = nb * 32;
    f.map.col_lo = 2 * col_lo;
    f.map.row_hi = 2 * row_hi;
    sc_internal_fu_assign_rows(&f);
    // measured at the cfg3 volume: 128 channels 3.71 vs 3.82 ms with the CSM waves loading, 64 channels 3.23 vs 2.98
    a.loader_csm = nb >= 3 ? 1 : 0;
    return a;
}

#define F2_SHAPES(X) X(1, 0, 1) X(2, 0, 2) X(3, 0, 3) X(4, 0, 4) X(4, 2, 2) X(3, 1, 3) X(4, 2, 4) X(4, 1, 4) X(4, 1, 1) X(3, 2, 2)
template <int NB32, int COL_LO, int ROW_HI, int OP>
static void fused2_launch_op(const Fused2Args& a, hipStream_t s) {
    auto k = fused2_kernel<NB32, COL_LO, ROW_HI, OP>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F2_LDS);
    hipLaunchKernelGGL(k, dim3((unsigned)(a.f.n_bins * a.f.n_split)), dim3(FU_THREADS), F2_LDS, s, a);
}
static int fused2_launch(const Fused2Args& a, int op, hipStream_t s) {
    const int shape = a.f.NB32 * 100 + a.f.shape_col_lo * 10 + a.f.shape_row_hi;
#define F2_CASE(NB32, COL_LO, ROW_HI)                                                              \
    case NB32 * 100 + COL_LO * 10 + ROW_HI:                                                        \
        if (op == FU_OP_SQ) fused2_launch_op<NB32, COL_LO, ROW_HI, FU_OP_SQ>(a, s);                \
        else if (op == FU_OP_SIGN) fused2_launch_op<NB32, COL_LO, ROW_HI, FU_OP_SIGN>(a, s);       \
        else fused2_launch_op<NB32, COL_LO, ROW_HI, FU_OP_ABS>(a, s);                              \
        break;
    switch (shape) {
        F2_SHAPES(F2_CASE)
    default:
        sc_set_error("planes-format stage B: no launch shape (%d staged blocks, column %d, %d rows)", a.f.NB32, a.f.shape_col_lo, a.f.shape_row_hi);
        return SC_EINVAL;
    }
#undef F2_CASE
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
// every tile of the record once (the launch plans of sc_fused.hip: launch_fused_all)
static int fused2_launch_all(const Fused2Args& full, int op, hipStream_t s) {
    const int C = full.f.st.C, n = (C + 31) / 32;
    struct Plan { int nb, blocks[4], col_lo, row_hi; };
    static const Plan tri[4] = {{1, {0}, 0, 1}, {2, {0, 1}, 0, 2}, {3, {0, 1, 2}, 0, 3}, {4, {0, 1, 2, 3}, 0, 4}};
    static const Plan p5[] = {{3, {0, 1, 2}, 0, 3}, {4, {0, 1, 3, 4}, 2, 2}, {3, {2, 3, 4}, 1, 3}};
    static const Plan p6[] = {{4, {0, 1, 2, 3}, 0, 4}, {4, {0, 1, 4, 5}, 2, 4}, {4, {2, 3, 4, 5}, 2, 2}};
    static const Plan p7[] = {{4, {0, 1, 2, 3}, 0, 4}, {4, {0, 4, 5, 6}, 1, 4}, {4, {1, 4, 5, 6}, 1, 1},
                              {4, {2, 4, 5, 6}, 1, 1}, {4, {3, 4, 5, 6}, 1, 1}};
    static const Plan p8[] = {{4, {0, 1, 2, 3}, 0, 4}, {4, {4, 5, 6, 7}, 0, 4}, {4, {0, 1, 4, 5}, 2, 2},
                              {4, {0, 1, 6, 7}, 2, 2}, {4, {2, 3, 4, 5}, 2, 2}, {4, {2, 3, 6, 7}, 2, 2}};
    int rc = SC_OK;
    if (n > 8) {
        // More than 256 signals (round 6; before: the host tiled the channels in blocks of 128 and paid a gathered 256-channel triangle per
        // block pair, on the complex64 kernels).  Groups of four consecutive blocks take their triangle; two halves (pairs of blocks) of
        // DIFFERENT groups take their 64 x 64 rectangle; with an odd block count the last block is a group (or the third block of one)
        // of its own and meets every pair outside its group as a 64 x 32 rectangle.  Every 32 x 32 block product exactly once.
        const int n_pairs = n / 2, lone = (n & 1) ? n - 1 : -1;
        for (int g0 = 0; g0 < n && rc == SC_OK; g0 += 4) {
            const int nbg = n - g0 < 4 ? n - g0 : 4;
            Plan t = {nbg, {g0, g0 + 1, g0 + 2, g0 + 3}, 0, nbg};
            rc = fused2_launch(fused2_args_blocks(full, t.blocks, t.nb, t.col_lo, t.row_hi), op, s);
        }
        for (int hi = 0; hi < n_pairs && rc == SC_OK; ++hi) {
            for (int hj = hi + 1; hj < n_pairs && rc == SC_OK; ++hj) {
                if (hi / 2 == hj / 2) continue;                    // the two halves of one group: inside its triangle
                const int b[4] = {2 * hi, 2 * hi + 1, 2 * hj, 2 * hj + 1};
                rc = fused2_launch(fused2_args_blocks(full, b, 4, 2, 2), op, s);
            }
            if (lone >= 0 && hi / 2 != lone / 4 && rc == SC_OK) {
                const int b[3] = {2 * hi, 2 * hi + 1, lone};
                rc = fused2_launch(fused2_args_blocks(full, b, 3, 2, 2), op, s);
            }
        }
        return rc;
    }
    const Plan* plan = n <= 4 ? &tri[n - 1] : n == 5 ? p5 : n == 6 ? p6 : n == 7 ? p7 : p8;
    const int n_launch = n <= 4 ? 1 : n == 5 ? 3 : n == 6 ? 3 : n == 7 ? 5 : 6;
    for (int l = 0; l < n_launch && rc == SC_OK; ++l)
        rc = fused2_launch(fused2_args_blocks(full, plan[l].blocks, plan[l].nb, plan[l].col_lo, plan[l].row_hi), op, s);
    return rc;
}

extern "C" int sc_fused2_supported(const sc_spectra_desc* desc, uint32_t planes) {
    Fused2Args a;
    ScAxes ax;
    const int rc = fused2_setup(desc, planes, &a, &ax);
    return rc == SC_OK ? 1 : 0;
}

static int fused2_run(const void* d_P, const sc_spectra_desc* desc, const float* d_scale, uint32_t planes,
                      float* d_accum, void* d_workspace, int64_t workspace_bytes, int* n_parts, void* stream) {
    ScTimed timed_("fused_stage_b", stream);
    SC_REQUIRE(d_P && desc && d_accum && d_scale, "NULL argument");
    SC_REQUIRE(((uintptr_t)d_P % 16) == 0, "planes buffer must be 16-byte aligned");
    Fused2Args a;
    ScAxes ax;
    const int rc = fused2_setup(desc, planes, &a, &ax);
    if (rc != SC_OK) return rc;
    FusedArgs& f = a.f;
    a.P = (const unsigned char*)d_P;
    a.inv_scale = d_scale + ax.C;
    f.accum = d_accum;
    {
        const char* dbg = sc_switch(SC_SW_FUSED_DEBUG);
        f.debug_skip = dbg ? atoi(dbg) : 0;
        const char* t4 = sc_switch(SC_SW_FUSED2_TERMS);
        if (t4) a.terms4 = atoi(t4) == 4 ? 1 : 0;
    }
    int S = sc_internal_fused_pick_split(f.n_bins, ax.n_obs);
    const int64_t part_bytes = (int64_t)f.n_bins * f.floats_per_bin * (int64_t)sizeof(float);
    if (!d_workspace) S = 1;
    while (S > 1 && (int64_t)(S - 1) * part_bytes > workspace_bytes) --S;
    SC_REQUIRE(S == 1 || ((uintptr_t)d_workspace % 16) == 0, "workspace must be 16-byte aligned");
    f.n_split = S;
    f.ws = (float*)d_workspace;
    {
        // Fold interval of the two-level summation: 512 observations (16 matrix instructions per accumulator between folds).
        // Round 5 measured what longer intervals buy and cost at cfg3 (three parts of 2333 observations; profiles/r05_fused2_fold_ab.txt):
        // no intermediate fold at all is 0.04 ms (1 %) faster and takes the coherence of the strongly coupled bins from 0.39 to 1.67
        // of the full-depth bound -- the accumulator's roundings do not average out over 73 instructions.  512 stays.
        // SC_FUSED_FOLD_OBS overrides (A/B).
        a.fold_obs = 512;
        const char* fo = sc_switch(SC_SW_FUSED_FOLD_OBS);
        if (fo && atoi(fo) >= 64 * 9) a.fold_obs = atoi(fo) / 64 * 64;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool parts_ok = n_parts && ax.C <= 128 && f.sq_plane < 0 && f.sign_plane < 0;
    if (n_parts) *n_parts = 1;
    if (f.sign_plane >= 0) {
        // sign(Im s) summed as integers by the |Im s| waves; the CSM waves only load
        Fused2Args b = a;
        b.f.csm_plane = -1;
        b.f.abs_plane = f.sign_plane;
        b.f.nl_op = FU_OP_SIGN;
        b.f.fold[0] = b.f.abs_plane; b.f.n_fold = 1;
        const int rc2 = fused2_launch_all(b, FU_OP_SIGN, s);
        return rc2 != SC_OK ? rc2 : sc_internal_fused_combine(b.f, FU_OP_SIGN, s);
    }
    {
        Fused2Args m = a;
        m.f.sq_plane = -1;
        const int rc2 = fused2_launch_all(m, FU_OP_ABS, s);
        if (rc2 != SC_OK) return rc2;
        if (parts_ok) { *n_parts = f.n_split; return SC_OK; }      // the caller's epilogue sums the parts
        const int rc3 = sc_internal_fused_combine(m.f, FU_OP_ABS, s);
        if (rc3 != SC_OK || f.sq_plane < 0) return rc3;
    }
    // debiased wPLI: sum (Im s)^2 as a second pass (the |Im s| waves hold 80 accumulator registers per plane)
    Fused2Args b = a;
    b.f.csm_plane = -1;
    b.f.abs_plane = f.sq_plane;
    b.f.sq_plane = -1;
    b.f.nl_op = FU_OP_SQ;
    b.f.fold[0] = b.f.abs_plane; b.f.n_fold = 1;
    const int rc4 = fused2_launch_all(b, FU_OP_SQ, s);
    return rc4 != SC_OK ? rc4 : sc_internal_fused_combine(b.f, FU_OP_SQ, s);
}
extern "C" int sc_fused2_csm_absim_f32(const void* d_P, const sc_spectra_desc* desc, const float* d_scale, uint32_t planes,
                                       float* d_accum, void* d_workspace, int64_t workspace_bytes, void* stream) {
    return fused2_run(d_P, desc, d_scale, planes, d_accum, d_workspace, workspace_bytes, nullptr, stream);
}
extern "C" int sc_fused2_csm_absim_parts_f32(const void* d_P, const sc_spectra_desc* desc, const float* d_scale, uint32_t planes,
                                             float* d_accum, void* d_workspace, int64_t workspace_bytes, int* n_parts,
                                             void* stream) {
    SC_REQUIRE(n_parts, "NULL argument");
    return fused2_run(d_P, desc, d_scale, planes, d_accum, d_workspace, workspace_bytes, n_parts, stream);
}
