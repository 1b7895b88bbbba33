// sc_common.h -- shared host/device helpers of libsc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sc_hip.h"

void sc_set_error(const char* fmt, ...);

// Diagnostic switches (ablation / A-B tools, tests of alternative kernels): environment variables read ONCE when the library is
// loaded and again whenever the host calls sc_debug_reload_env() -- no getenv on a launch path.  sc_switch(id) is the value
// string or nullptr when the variable is unset.
enum ScSwitch {
    SC_SW_FUSED_DEBUG, SC_SW_FUSED2_TERMS, SC_SW_FUSED_SPLIT, SC_SW_FUSED_NO_SMALL, SC_SW_MTFFT_DEBUG, SC_SW_MTFFT_WIDE,
    SC_SW_MTFFT_F64, SC_SW_F64_SPLIT, SC_SW_F64_OC, SC_SW_F64_NO_FORK, SC_SW_F64_NO_BLOCK, SC_SW_WILSON_FFT, SC_SW_GLOBAL_EIG,
    SC_SW_GLOBAL_NT256, SC_SW_GRANGER_KERNEL, SC_SW_FUSED_FOLD_OBS, SC_SW_MTFFT_LONG, SC_SW_CANON_EIG, SC_SW_MTFFT_MIXED, SC_SW_MTFFT_MIXED_GEO, SC_SW_MTFFT_SLICE, SC_SW_MVAR_INVERSE, SC_SW_COUNT
};
const char* sc_switch(int id);

// sc_api.hip: Z[rows][F] -> X[f][b_off + row] (X has `batch` columns), imaginary part of rows 0 and nyquist_row zeroed
int sc_internal_rows_to_bins(const void* d_Z, void* d_X, int64_t rows, int64_t F, int64_t batch, int64_t b_off,
                             int64_t nyquist_row, hipStream_t st);

// sc_mtfft_long.hip: stage A for long power-of-two windows (two half-workgroups in anti-phase)
bool sc_internal_mtfft_long_applies(int64_t N, int64_t C, int64_t groups);
int64_t sc_internal_mtfft_long_coverage(int64_t N, int64_t C);
int sc_internal_mtfft_long(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L, int64_t step, int64_t W, int64_t N,
                           const float* d_tapers, int64_t K, int detrend_type, const void* d_twiddles, void* d_X, void* d_P,
                           const float* d_scale, hipStream_t st);

// sc_mtfft_mixed.hip: stage A for the window lengths N = 10 RM RF that are not powers of two (register-resident radix-10 passes,
// anti-phase half-workgroups, planes-format output)
bool sc_internal_mtfft_mix_has(int64_t N);
bool sc_internal_mtfft_mix_applies(int64_t N, int64_t C, int64_t groups);
int64_t sc_internal_mtfft_mix_coverage(int64_t N, int64_t C, bool planes);
int sc_internal_mtfft_mix(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L, int64_t step, int64_t W, int64_t N,
                          const float* d_tapers, int64_t K, int detrend_type, const void* d_twiddles, void* d_X, void* d_P,
                          const float* d_scale, hipStream_t st);

// sc_api.hip: process-wide cache of the batched in-place complex128 rocFFT plans of the Wilson kernels (never destroyed: see there);
// *cached == false: the table is full and the caller destroys the plan
struct rocfft_plan_t;
int sc_internal_z2z_plan(struct rocfft_plan_t** plan, int forward, size_t N, size_t batch, bool* cached);

// sc_wilson_fft.hip: A <- fft(causal(ifft(A))) in one kernel, for the lengths `supported` accepts
bool sc_internal_causal_fft_supported(int64_t N);
bool sc_internal_causal_fft_natural_supported(int64_t N);
int sc_internal_causal_fft_pair_natural(void* d_A, const int32_t* d_status, int64_t n_problems, int C, int64_t N,
                                        hipStream_t st);
int sc_internal_causal_fft_pair(void* d_A, const int32_t* d_status, int64_t n_problems, int C, int64_t N,
                                hipStream_t st);

// sc_wilson_pair.hip: pairwise Granger with the 2 x 2 Wilson iteration of a pair resident on one compute unit
bool sc_internal_granger_resident_applies(int64_t n_freq_accum, int64_t N);
int sc_internal_granger_resident(const void* d_accum, int64_t n_groups, int64_t N, int64_t C, uint32_t planes, int64_t n_obs,
                                 const int32_t* d_pairs, int64_t n_pairs, double tol, int max_iter, void* d_work, size_t work_bytes,
                                 int keep_output, double* d_out, int32_t* d_n_iter, int32_t* d_status, int32_t* h_summary,
                                 hipStream_t st);

// sc_timing.hip: brackets the launches of an entry point with two hipEvents on its stream while sc_timing_enable(1)
struct ScTimed {
    int slot;
    void* st;
    ScTimed(const char* name, void* stream);
    ~ScTimed();
    ScTimed(const ScTimed&) = delete;
    ScTimed& operator=(const ScTimed&) = delete;
};

#define SC_CHECK_HIP(expr)                                                        \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            sc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                         __FILE__, __LINE__);                                     \
            return SC_EHIP;                                                       \
        }                                                                         \
    } while (0)

#define SC_REQUIRE(cond, msg)                                                     \
    do {                                                                          \
        if (!(cond)) {                                                            \
            sc_set_error("invalid argument: %s (%s)", msg, #cond);                \
            return SC_EINVAL;                                                     \
        }                                                                         \
    } while (0)

// Accumulator records as their consumers see them: float elements (f32 engine) or double elements (f64 engine,
// SC_RECORD_F64 set in `planes`).  Pointer arithmetic in elements, loads widened to double.
struct ScRec {
    const void* p;
    int f64;
    __host__ __device__ ScRec operator+(int64_t n) const { return ScRec{(const char*)p + n * (f64 ? 8 : 4), f64}; }
    __device__ double operator[](int64_t i) const {
        return f64 ? ((const double*)p)[i] : (double)((const float*)p)[i];
    }
};
__host__ inline ScRec sc_rec(const void* d_accum, uint32_t planes) { return ScRec{d_accum, (planes & SC_RECORD_F64) ? 1 : 0}; }

#define SC_TILE 16            // channel tile edge of the accumulator layout
#define SC_TILE_ELEMS 256
#define SC_MAX_SIGNALS 256    // v1 kernels stage all channels of an observation row in LDS

// number of planes / offset (in planes) of a plane inside a bin record
__host__ __device__ inline int sc_plane_count(uint32_t planes) {
    int n = 0;
    if (planes & SC_PLANE_CSM) n += 2;
    if (planes & SC_PLANE_ABS_IM) n += 1;
    if (planes & SC_PLANE_IM_SQ) n += 1;
    if (planes & SC_PLANE_SIGN_IM) n += 1;
    if (planes & SC_PLANE_UNIT) n += 2;
    return n;
}
__host__ __device__ inline int sc_plane_offset(uint32_t planes, uint32_t which) {
    int n = 0;
    if (which == SC_PLANE_CSM) return n;
    if (planes & SC_PLANE_CSM) n += 2;
    if (which == SC_PLANE_ABS_IM) return n;
    if (planes & SC_PLANE_ABS_IM) n += 1;
    if (which == SC_PLANE_IM_SQ) return n;
    if (planes & SC_PLANE_IM_SQ) n += 1;
    if (which == SC_PLANE_SIGN_IM) return n;
    if (planes & SC_PLANE_SIGN_IM) n += 1;
    return n;  // SC_PLANE_UNIT
}
__host__ __device__ inline int sc_n_blocks(int64_t C) { return (int)((C + SC_TILE - 1) / SC_TILE); }
__host__ __device__ inline int sc_n_tiles(int nb) { return nb * (nb + 1) / 2; }
// packed index of upper tile (bi <= bj)
__host__ __device__ inline int sc_tile_index(int bi, int bj, int nb) {
    return bi * nb - bi * (bi - 1) / 2 + (bj - bi);
}

// Decoding of group / observation indices into element offsets (see sc_spectra_desc).
struct ScAxes {
    int64_t sW, sR, sK, sF;
    int W, R, K, C, F;
    int kW, kR, kK;   // kept sizes (1 if reduced)
    int rW, rR, rK;   // reduced sizes (1 if kept)
    int n_groups, n_obs;
};

inline int sc_make_axes(const sc_spectra_desc* d, ScAxes* a) {
    if (!d) return SC_EINVAL;
    a->sW = d->stride_window; a->sR = d->stride_trial; a->sK = d->stride_taper; a->sF = d->stride_freq;
    a->W = (int)d->n_windows; a->R = (int)d->n_trials; a->K = (int)d->n_tapers;
    a->C = (int)d->n_signals; a->F = (int)d->n_freq;
    a->kW = d->reduce_window ? 1 : a->W; a->rW = d->reduce_window ? a->W : 1;
    a->kR = d->reduce_trial ? 1 : a->R;  a->rR = d->reduce_trial ? a->R : 1;
    a->kK = d->reduce_taper ? 1 : a->K;  a->rK = d->reduce_taper ? a->K : 1;
    a->n_groups = a->kW * a->kR * a->kK;
    a->n_obs = a->rW * a->rR * a->rK;
    return SC_OK;
}

__device__ inline int64_t sc_group_offset(const ScAxes& a, int g) {
    int gk = g % a.kK; g /= a.kK;
    int gr = g % a.kR; g /= a.kR;
    return (int64_t)g * a.sW + (int64_t)gr * a.sR + (int64_t)gk * a.sK;
}
__device__ inline int64_t sc_obs_offset(const ScAxes& a, int o) {
    int ok = o % a.rK; o /= a.rK;
    int orr = o % a.rR; o /= a.rR;
    return (int64_t)o * a.sW + (int64_t)orr * a.sR + (int64_t)ok * a.sK;
}

// Streaming stores of the spectra (written once, read by the next kernel from HBM: 6.5 GB at cfg3 against 32 MB of L2 and
// 256 MB of MALL): the non-temporal hint keeps the lines from lingering in the write-back L2 -- stage A 1.537 -> 1.471 ms at
// 256 samples, 1.517 -> 1.406 at 128, 2.186 -> 2.141 at 1024 (A/B inside one process, profiles/r03_stage_a_ab.txt).  ONLY for
// store groups that cover whole 128-byte lines: partial lines need the L2 to merge them (4096-sample windows, 32-byte
// pieces: 0.91 -> 1.73 ms with the hint; the float64 transform's 16-byte halves: 4.1 -> 6.1 ms).
#ifdef __HIPCC__
typedef float sc_f32x4 __attribute__((ext_vector_type(4)));
typedef float sc_f32x2 __attribute__((ext_vector_type(2)));
typedef double sc_f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sc_stream_store(float2* p, float2 a, float2 b) {          // 16-byte aligned pair
    const sc_f32x4 v = {a.x, a.y, b.x, b.y};
    __builtin_nontemporal_store(v, reinterpret_cast<sc_f32x4*>(p));
}
__device__ __forceinline__ void sc_stream_store(float2* p, float2 a) {
    const sc_f32x2 v = {a.x, a.y};
    __builtin_nontemporal_store(v, reinterpret_cast<sc_f32x2*>(p));
}
__device__ __forceinline__ void sc_stream_store(double2* p, double2 a) {
    const sc_f64x2 v = {a.x, a.y};
    __builtin_nontemporal_store(v, reinterpret_cast<sc_f64x2*>(p));
}
#endif
