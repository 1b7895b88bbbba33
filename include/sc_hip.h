/* sc_hip.h -- C ABI of libsc_hip.so: the MI355X (gfx950) engine behind the
 * Multitaper / Connectivity API of Eden-Kramer-Lab/spectral_connectivity.
 *
 * This is the drop-in boundary for ONE path of the reference: time series ->
 * DPSS-tapered sliding-window FFT -> cross-spectral matrix -> expectation ->
 * coherence / PLI / wPLI / PLV / PPC (-> pairwise spectral Granger, canonical coherence).
 * It replaces the reference's array-backend plug point, the import-time switch
 * `xp = cupy | numpy` plus `fft/ifft/...` (transforms.py:405-439, connectivity.py:31-65,
 * minimum_phase_decomposition.py:14-26), for the call sites listed per function below.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes.  Every `dev_*`/`d_*` pointer is DEVICE memory
 *    owned by the caller (the Python host allocates it with torch); the library never frees or
 *    retains caller buffers past the call.  Device memory the library allocates ITSELF, all of it
 *    released before the owning call returns or with the owning handle:
 *      - `sc_fft_plan` handles (sc_fft_plan_create*): the rocFFT plan, its work buffer and a <= 64 MB
 *        transform scratch (hipMalloc; freed by sc_fft_plan_destroy; sc_fft_plan_work_bytes reports it) -- the rocFFT plan
 *        object itself (twiddle tables, run-time compiled code: KBs to a few MB) is retired, not destroyed, and the
 *        Wilson kernels keep theirs for the life of the process: unloading rocFFT's code right before the first
 *        launch of another kernel has run stale instructions on MI355X / ROCm 7.0;
 *      - stream-ordered scratch (hipMallocAsync / hipFreeAsync on the call's stream) inside
 *        sc_multitaper_fft_f32 for N = 4096 (row-major spectra before the transpose, <= 2 GB),
 *        sc_canonical_coherence_f64 (inverted group factors, n_bins * n_groups * 4 KB; groups beyond 32 channels:
 *        the blocks of the pairs in flight, <= 192 MB),
 *        sc_global_coherence_f64 above 64 signals (rotation log; above 128 signals also the packed matrices) and the
 *        rocFFT work buffers of
 *        sc_granger_pairwise_f64 / sc_wilson_factor_f64 / sc_mvar_factor_f64 for lengths their fused
 *        transform kernel does not take.
 *      - sc_accumulate_f64: one side stream per device (created on first use, kept) and two events per call (destroyed
 *        when it returns) -- it forks its per-observation plane kernels from the caller's stream and joins them back
 *        before it returns --, and a stream-ordered scratch of (S - 1) records (<= 2 GB) for the partial sums of the S
 *        workgroups that share a bin's observations (hipMallocAsync / hipFreeAsync on `stream`).
 *      - sc_global_coherence_f64 beyond 64 signals: the matrices (n_signals^2 x 16 B per resident workgroup, <= 1024 of
 *        them) and per-vector work arrays of its eigen-solver (hipMalloc / hipFree inside the call, which synchronises).
 *    Everything else (spectra, records, workspaces, outputs) is caller memory with sizes the
 *    *_bytes / sc_accum_layout queries report.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All calls are
 *    asynchronous on that stream; nothing synchronises the device.
 *  - Return value: 0 on success, negative SC_E* otherwise; sc_last_error() returns a
 *    thread-local message for the last failure.
 *  - One plan / one stream is used by one host thread at a time; different streams may be
 *    driven from different threads.
 *
 * Device layouts (row-major, last index fastest)
 *  - time series      x[T][R][C]                      float   (reference layout, transforms.py:574)
 *  - tapers           h[K][L]                         float   (= reference tapers^T / fs, see sc_taper_windows_f32)
 *  - tapered windows  y[N][W][R][K][C]                float   (time-major: lanes <-> channels)
 *  - spectra          X[F][W][R][K][C]                float2  (one-sided, F = N/2+1) -- or any
 *                     layout described by sc_spectra_desc strides (e.g. the reference's
 *                     own (W,R,K,N,C) order for uploaded coefficients)
 *  - accumulators     A[bin][plane][tile][16][16]     float (double when SC_RECORD_F64 is set in `planes`: the
 *                     float64 engine)   bin = group*F + f ; upper-triangular
 *                     16x16 channel tiles (bi <= bj), tile index = bi*NB - bi*(bi-1)/2 + (bj-bi);
 *                     UN-normalised sums over observations (so trial shards can be summed)
 *  - measures         M[bin][C][C]                    float (or float2 for complex measures)
 */
#ifndef SC_HIP_H
#define SC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 6

/* error codes */
#define SC_OK 0
#define SC_EINVAL (-1)   /* bad argument / shape                 */
#define SC_EHIP (-2)     /* HIP runtime error (launch, memcpy)   */
#define SC_EFFT (-3)     /* rocFFT error                         */
#define SC_ENOMEM (-4)   /* work-buffer allocation failed        */
#define SC_EUNSUPPORTED (-5)

/* detrend_type of sc_taper_windows_f32 (transforms.py:1798-1915) */
#define SC_DETREND_NONE 0
#define SC_DETREND_CONSTANT 1
#define SC_DETREND_LINEAR 2

/* accumulator planes (bit mask).  Plane order inside a bin record is the order below,
 * skipping planes that are not requested.  CSM always occupies two planes (re, im). */
#define SC_PLANE_CSM 0x01u      /* sum X_i conj(X_j)            : 2 planes (re, im)  connectivity.py:447-461 */
#define SC_PLANE_ABS_IM 0x02u   /* sum |Im s|                   : 1 plane            connectivity.py:1008-1014 */
#define SC_PLANE_IM_SQ 0x04u    /* sum (Im s)^2                 : 1 plane            connectivity.py:1096-1102 */
#define SC_PLANE_SIGN_IM 0x08u  /* sum sign(Im s)               : 1 plane            connectivity.py:970-980 */
#define SC_PLANE_UNIT 0x10u     /* sum s/|s|                    : 2 planes (re, im)  connectivity.py:899-903 */
/* record format flag, OR-ed into `planes` wherever accumulator records are handed to a consumer (sc_measure_*,
 * sc_granger_pairwise_f64, sc_mvar_factor_f64, sc_global_coherence_f64, sc_canonical_coherence_f64): the records
 * hold doubles (written by sc_accumulate_f64) instead of floats */
#define SC_RECORD_F64 0x100u

/* measures of sc_measure_f32 (all on non-negative frequency bins the caller accumulated) */
#define SC_M_POWER 0                 /* connectivity.py:612-630   out float [bin][C]        */
#define SC_M_CSM 1                   /* connectivity.py:463-492   out float2 [bin][C][C]    */
#define SC_M_COHERENCY 2             /* connectivity.py:632-657   out float2                */
#define SC_M_COHERENCE_MAGNITUDE 3   /* connectivity.py:675-702   out float                 */
#define SC_M_COHERENCE_PHASE 4       /* connectivity.py:659-673                             */
#define SC_M_IMAGINARY_COHERENCE 5   /* connectivity.py:704-743                             */
#define SC_M_PLV 6                   /* connectivity.py:905-931   needs SC_PLANE_UNIT       */
#define SC_M_PLI 7                   /* connectivity.py:933-980   needs SC_PLANE_SIGN_IM    */
#define SC_M_WPLI 8                  /* connectivity.py:982-1028  needs CSM + ABS_IM        */
#define SC_M_DEBIASED_PLI2 9         /* connectivity.py:1030-1058 needs SIGN_IM             */
#define SC_M_DEBIASED_WPLI2 10       /* connectivity.py:1060-1127 needs CSM+ABS_IM+IM_SQ    */
#define SC_M_PPC 11                  /* connectivity.py:1129-1159 needs UNIT                */
#define SC_M_PLV_COMPLEX 12          /* connectivity.py:897-903   out float2                */

typedef struct sc_fft_plan sc_fft_plan; /* opaque */

/* How the (window, trial, taper) axes of the spectra map to memory and which of them the
 * expectation averages (connectivity.py:67-75 EXPECTATION).  Strides are in complex
 * elements; the channel stride is 1. */
typedef struct sc_spectra_desc {
    int64_t n_freq;      /* F: number of frequency bins to process                    */
    int64_t n_windows;   /* W */
    int64_t n_trials;    /* R */
    int64_t n_tapers;    /* K */
    int64_t n_signals;   /* C */
    int64_t stride_freq; /* elements between consecutive bins                          */
    int64_t stride_window;
    int64_t stride_trial;
    int64_t stride_taper;
    int32_t reduce_window; /* 1 if the expectation averages over windows ("time")     */
    int32_t reduce_trial;
    int32_t reduce_taper;
    int32_t reserved;
} sc_spectra_desc;

/* ---- library ------------------------------------------------------------------------ */
int sc_abi_version(void);
const char* sc_last_error(void);
int sc_device_count(int* count);

/* ---- stage A: window extraction + detrend + taper multiply (custom HIP) ---------------
 * Replaces _sliding_window (transforms.py:1311-1374), detrend (:1798-1915) and the taper
 * broadcast-multiply of _multitaper_fft (:1402-1404), fused, no window copy.
 *   y[w][r][k][c][n] = (x[w*step+n][r][c] - trend) * h[k][n]   for n <  min(L, N)
 *                    = 0                                        for min(L,N) <= n < N
 * (one row of N samples per transform, n fastest: the layout rocFFT streams at full rate)
 * The caller folds the reference's sqrt(fs) taper scaling and the 1/fs of
 * transforms.py:1405 into h (h = tapers^T / fs). */
int sc_taper_windows_f32(const float* d_x, int64_t T, int64_t R, int64_t C,
                         int64_t L, int64_t step, int64_t W, int64_t N,
                         const float* d_tapers, int64_t K, int detrend_type,
                         float* d_y, void* stream);
/* float64 engine: the reference's own precision (it promotes every input to float64, transforms.py:1402-1405) */
int sc_taper_windows_f64(const double* d_x, int64_t T, int64_t R, int64_t C,
                         int64_t L, int64_t step, int64_t W, int64_t N,
                         const double* d_tapers, int64_t K, int detrend_type,
                         double* d_y, void* stream);

/* float64 time series [T][R][C] -> the float32 copy [T][R][C_out] the float32 engine transforms (C_out >= C: channels
 * beyond C are zero, the pad channel of odd counts).  remove_mean != 0 subtracts the per-(trial, channel) mean over time
 * in float64 BEFORE the cast -- every window's detrend (transforms.py:1798-1915) removes any constant anyway, so only the
 * rounding changes: a DC offset many times the signal no longer costs the float32 copy its digits. */
int sc_timeseries_to_f32(const double* d_x, int64_t T, int64_t R, int64_t C, int remove_mean, float* d_y,
                         int64_t C_out, void* stream);

/* ---- stage A: batched real-to-complex FFT (rocFFT) -----------------------------------
 * Replaces fft(projected, n=N, axis=-2) of transforms.py:1405 (scipy.fft / cupyx.scipy.fft).
 * Input  y[batch][N] float (the rows sc_taper_windows_f32 writes), output X[F][batch] float2
 * (F = N/2+1, frequency-major like every spectra tensor here).  The plan transforms the rows in
 * unit-stride chunks into a scratch it owns (64 MB, cache-resident) and a tiled transpose moves
 * each chunk into X, zeroing the imaginary part of the DC / Nyquist rows on the way;
 * sc_fft_plan_work_bytes reports the device memory the plan holds. */
int sc_fft_plan_create(sc_fft_plan** plan, int64_t N, int64_t batch);
int sc_fft_plan_work_bytes(const sc_fft_plan* plan, size_t* bytes);
int sc_fft_execute(sc_fft_plan* plan, const float* d_y, void* d_X /*float2*/, void* stream);
int sc_fft_plan_destroy(sc_fft_plan* plan);
/* double-precision plans of the float64 engine: y double rows in, X[F][batch] double2 out */
int sc_fft_plan_create_f64(sc_fft_plan** plan, int64_t N, int64_t batch);
int sc_fft_execute_f64(sc_fft_plan* plan, const double* d_y, void* d_X /*double2*/, void* stream);

/* ---- stage A, fused fast path (custom HIP) ---------------------------------------------
 * Window extraction + detrend + taper multiply + real FFT + transposed store in ONE kernel:
 * replaces _sliding_window, detrend, _multitaper_fft and the swapaxes of Multitaper.fft
 * (transforms.py:1147-1171, :1311-1405) with a single read of x and a single write of
 *   X[f][w][r][k][c],  f = 0..N/2   (one-sided; the negative bins of a real input are
 *                                    conjugate mirrors and are never materialised).
 * Supported when sc_multitaper_fft_supported(L, N) != 0: L <= N and N either a power of two in
 * 64 ... 4096 (register-resident radix-16 passes; from 256 samples on, when the request fills the chip, two half-workgroups
 * in anti-phase -- one runs the passes while the other stores: sc_mtfft_long.hip) or 2^a 3^b 5^c in 8 ... 2048 (mixed-radix Stockham
 * passes in LDS: the lengths next_fast_len, transforms.py:1024-1036, returns for the usual window
 * durations -- 200, 250, 500, 1000 ...); otherwise use sc_taper_windows_f32 + sc_fft_execute
 * (rocFFT, any length).
 * d_twiddles: float2[N] device table filled once by sc_fft_twiddles_f32(N, ...). */
int sc_multitaper_fft_supported(int64_t L, int64_t N);
int sc_fft_twiddles_f32(int64_t N, void* d_twiddles /*float2[N]*/, void* stream);
int sc_multitaper_fft_f32(const float* d_x, int64_t T, int64_t R, int64_t C,
                          int64_t L, int64_t step, int64_t W, int64_t N,
                          const float* d_tapers, int64_t K, int detrend_type,
                          const void* d_twiddles, void* d_X /*float2*/, void* stream);
/* float64 engine, the same fusion in doubles (sc_mtfft_f64.hip): complex128 spectra X[f][w][r][k][c]; powers of two
 * 64 ... 1024: register-resident radix-16 passes like the float32 kernel; the next_fast_len lengths: one wave per
 * packed channel pair, in-place mixed-radix passes in LDS; twiddles computed by the kernel.  Lengths with a compiled
 * transform: 64, 128, 256, 512, 1024 and 200, 250, 400, 500, 1000 (sc_multitaper_fft_f64_supported); every other
 * length: sc_taper_windows_f64 + sc_fft_execute_f64. */
int sc_multitaper_fft_f64_supported(int64_t L, int64_t N);
int sc_multitaper_fft_f64(const double* d_x, int64_t T, int64_t R, int64_t C,
                          int64_t L, int64_t step, int64_t W, int64_t N,
                          const double* d_tapers, int64_t K, int detrend_type,
                          void* d_X /*double2*/, void* stream);

/* ---- stage B: accumulators -----------------------------------------------------------
 * Number of floats of one bin record for `planes`, and total bins (groups * F). */
int sc_accum_layout(const sc_spectra_desc* desc, uint32_t planes,
                    int64_t* n_bins, int64_t* floats_per_bin, int64_t* n_groups,
                    int64_t* n_observations);

/* Sum over the averaged axes of the per-observation cross-spectral outer product, on the
 * matrix cores (v_mfma_f32_16x16x4_f32), upper-triangular 16x16 tiles only.
 * Replaces _complex_inner_product + mean of connectivity.py:447-492, :1799-1822 without
 * materialising the per-observation (W,R,K,N,C,C) temporary.  Writes the two CSM planes
 * of every bin record in d_accum (other planes untouched).  Sums are UN-normalised: a rank
 * that holds a shard of the trials accumulates its shard and the records are added. */
int sc_csm_accumulate_f32(const void* d_X /*float2*/, const sc_spectra_desc* desc,
                          uint32_t planes, float* d_accum, void* stream);

/* Per-observation non-linear planes (|Im s|, (Im s)^2, sign Im s, s/|s|), VALU kernel.
 * Replaces the fcn hook of _expectation_cross_spectral_matrix (connectivity.py:463-492)
 * as used by PLV/PLI/wPLI/debiased variants/PPC (:897-1159).  Writes the requested
 * non-CSM planes of every bin record. */
int sc_nonlinear_accumulate_f32(const void* d_X /*float2*/, const sc_spectra_desc* desc,
                                uint32_t planes, uint32_t which, float* d_accum, void* stream);

/* Fused stage B for the headline pair coherence + wPLI (planes = CSM | ABS_IM; CSM alone is accepted too and
 * then only the matrix-core update runs): ONE pass over the spectra computes
 * the CSM planes AND the per-observation Im(x_i conj x_j) products of the ABS_IM plane on the bf16 matrix
 * pipe (exact 3-way bf16 split of every f32 coefficient; 12-wave workgroups: one CSM wave and two |Im| waves
 * per SIMD over one double-buffered LDS staging of the spectra, rows pulled HBM -> LDS directly).  Same results as
 * sc_csm_accumulate_f32 + sc_nonlinear_accumulate_f32(SC_PLANE_ABS_IM); even n_signals <= 256 (above 128 channels
 * the record is filled by several launches over 64-channel quarters: the two 128-channel halves, then every quarter
 * of the first half against every quarter of the second). */
int sc_fused_supported(int64_t n_signals);
int sc_fused_csm_absim_f32(const void* d_X /*float2*/, const sc_spectra_desc* desc,
                           uint32_t planes, float* d_accum, void* stream);
/* Same, with a device workspace (sc_fused_workspace_bytes; may be 0) that lets several workgroups share
 * one (window, frequency) bin when the bin count does not fill the GPU's compute units evenly: each
 * sums part of the observations into its own record, folded together in a fixed order afterwards. */
int64_t sc_fused_workspace_bytes(const sc_spectra_desc* desc, uint32_t planes);
int sc_fused_csm_absim_ws_f32(const void* d_X /*float2*/, const sc_spectra_desc* desc, uint32_t planes,
                              float* d_accum, void* d_workspace, int64_t workspace_bytes, void* stream);

/* SC_PLANE_UNIT of the record, sum over observations of s/|s| (phase_locking_value,
 * pairwise_phase_consistency: connectivity.py:897-981), through the same kernels: s/|s| factorises into
 * (x_i/|x_i|) conj(x_j/|x_j|), so the sum is the cross-spectral matrix of the unit phasors x/|x|
 * (0/0 -> NaN like the reference's x/abs(x)).  The normalisation happens while the rows are staged, in both kernels:
 * no copy of the spectra, d_scratch / scratch_bytes are ignored and sc_fused_unit_scratch_bytes returns 0 (both kept
 * for ABI v2 callers).  Shapes, workspace and split as above. */
/* SC_PLANE_UNIT for shapes the one-pass kernels do not take (odd channel counts without a pad channel): a normalised copy of the
 * spectra (x/|x|, 0 -> NaN) in d_scratch (sc_unit_scratch_bytes) goes through the f32-MFMA CSM kernel. */
int64_t sc_unit_scratch_bytes(const sc_spectra_desc* desc);
int sc_unit_accumulate_f32(const void* d_X /*float2*/, const sc_spectra_desc* desc, uint32_t planes,
                           float* d_accum, void* d_scratch, int64_t scratch_bytes, void* stream);

/* Which planes of `planes` the one-pass entry points fill for this shape (even n_signals <= 256): CSM, |Im s| (with
 * CSM) and s/|s|; (Im s)^2 (with CSM and |Im s|, filled by sc_fused_csm_absim_ws_f32: in the same pass up to 52 channels,
 * as a second pass of the matrix-core kernel above) and sign(Im s) (sc_fused_sign_ws_f32: phase_lag_index,
 * connectivity.py:933-980; f32 VALU kernel up to 40 channels, above it a pass of the matrix-core kernel whose |Im| waves
 * sum sign(d) of the per-observation products).  Whatever is left is sc_nonlinear_accumulate_f32's. */
uint32_t sc_fused_planes_covered(const sc_spectra_desc* desc, uint32_t planes);
int sc_fused_sign_ws_f32(const void* d_X /*float2*/, const sc_spectra_desc* desc, uint32_t planes,
                         float* d_accum, void* d_workspace, int64_t workspace_bytes, void* stream);
int64_t sc_fused_unit_scratch_bytes(const sc_spectra_desc* desc);
int sc_fused_unit_ws_f32(const void* d_X /*float2*/, const sc_spectra_desc* desc, uint32_t planes,
                         float* d_accum, void* d_workspace, int64_t workspace_bytes, void* d_scratch,
                         int64_t scratch_bytes, void* stream);

/* ---- planes format (ABI v4): every real number stored as two f16 pieces, x * scale[c] = h + m ---------------------------
 * The one-pass stage-B kernels multiply on the 16-bit matrix pipe and split every f32 coefficient while they stage it; stage A
 * can store the pieces instead (sc_multitaper_fft_planes_f32): stage B's staging becomes plain HBM -> LDS loads and its
 * products need three cross terms instead of six.  h = f16(x scale), m = f16(x scale - h): 22 significant bits; scale[c] is
 * a power of two per channel that keeps every coefficient inside the f16 range (sc_planes_scales_*); records come out
 * unscaled (exactly: powers of two).  Layout: dense rows [F][W][R][K] (bin, window, trial, taper) of
 * sc_planes_row_bytes(C) = 256 * ceil(C / 32) bytes: [channel tile of 32][plane Re h, Re m, Im h, Im m][32 channels] f16,
 * absent channels zero -- 8 bytes per coefficient, like the complex64 spectra of _multitaper_fft (transforms.py:1377-1405).
 * d_scale: float[2 C] = the scales, then their reciprocals. */
int64_t sc_planes_row_bytes(int64_t n_signals);
/* scales from a bound on the spectra of a (T, R, C) time series: |X_k(f)| <= 8 max|x| * taper_abs_sum, taper_abs_sum =
 * max_k sum_n |tapers[k][n]| (the tapers as passed to stage A, i.e. / fs); d_work: 4 C bytes of device scratch */
int sc_planes_scales_from_series_f32(const float* d_x, int64_t T, int64_t R, int64_t C, double taper_abs_sum,
                                     float* d_scale, void* d_work, void* stream);
/* The same scan with the QUALITY CHECK of the format.  One scale per channel serves every window, so the format delivers its 22
 * bits only while a channel's typical coefficient stays well above the f16 subnormals in scaled units -- which a DC offset or an
 * artefact thousands of times the typical amplitude would prevent.  This form (a) takes the scale from the RANGE of the channel
 * when stage A detrends (detrend_type != 0: |x - trend| <= 4 (max x - min x), a DC offset no longer enters; without detrend
 * |x| <= max |x|), and (b) reports *d_quality (device float) = min over the channels of (typical sample magnitude: the standard
 * deviation -- root mean square without detrend -- of the channel's QUIETEST slab of >= 128 consecutive (time, trial) rows) * scale.  The typical coefficient in scaled units is that times
 * min_k ||tapers_k||_2; below SC_PLANES_MIN_TYPICAL the host keeps complex64 spectra (sc_multitaper_fft_f32) -- the planes
 * format is a device detail, never a precision trade.  d_work: sc_planes_scales_work_bytes(T * R, C) bytes. */
#define SC_PLANES_MIN_TYPICAL 2.5f
int64_t sc_planes_scales_work_bytes(int64_t n_rows, int64_t width);
int sc_planes_scales_quality_f32(const float* d_x, int64_t T, int64_t R, int64_t C, int detrend_type, double taper_abs_sum,
                                 float* d_scale, void* d_work, int64_t work_bytes, float* d_quality, void* stream);
/* scales from the largest |Re|, |Im| of dense complex64 rows [n_rows][C] */
int sc_planes_scales_from_spectra_f32(const void* d_X /*float2*/, int64_t n_rows, int64_t C, float* d_scale, void* d_work,
                                      void* stream);
/* Stage A straight into the planes format: sc_multitaper_fft_f32 with the spectra leaving as f16 pieces (same transform,
 * same coefficients up to the 22-bit representation; window lengths N = 64 ... 4096, powers of two, even n_signals). */
int sc_multitaper_fft_planes_supported(int64_t L, int64_t N, int64_t n_signals);
int sc_multitaper_fft_planes_f32(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L, int64_t step, int64_t W,
                                 int64_t N, const float* d_tapers, int64_t K, int detrend_type, const void* d_twiddles,
                                 const float* d_scale, void* d_P, void* stream);
/* complex64 spectra described by desc (strides in elements) <-> planes buffer (dense rows, desc's sizes) */
int sc_planes_from_spectra_f32(const void* d_X /*float2*/, const sc_spectra_desc* desc, const float* d_scale, void* d_P,
                               void* stream);
int sc_spectra_from_planes_f32(const void* d_P, const sc_spectra_desc* desc, const float* d_scale, void* d_X /*float2*/,
                               void* stream);
/* Stage B on the planes format: the CSM planes and the |Im s| plane of every bin record in one pass, exactly what
 * sc_fused_csm_absim_ws_f32 writes (replaces _expectation_cross_spectral_matrix with fcn = identity and abs(Im),
 * connectivity.py:463-526, :982-1028).  desc: sizes and reduce flags (strides ignored: the rows are dense).  Takes
 * planes == SC_PLANE_CSM, SC_PLANE_CSM | SC_PLANE_ABS_IM, the latter | SC_PLANE_IM_SQ (debiased wPLI: a second pass squares
 * the per-observation products) or SC_PLANE_SIGN_IM alone (phase_lag_index: connectivity.py:933-980); up to 1024 signals
 * (129 and more in several launches over 32-channel blocks; more than 256: round 6); observations of a bin that form one linear run of rows (every
 * expectation type but "time_tapers" with several trials); sc_fused2_supported tells.
 * Workspace as for sc_fused_csm_absim_ws_f32 (sc_fused_workspace_bytes with the same desc). */
int sc_fused2_supported(const sc_spectra_desc* desc, uint32_t planes);
/* The same without the pass that folds the split-bin partial records: *n_parts workgroups shared every bin; part 0 of the
 * records is in d_accum, part k >= 1 at (float*)d_workspace + (k - 1) * n_bins * floats_per_bin (sc_accum_layout) -- for
 * sc_measure_multi_parts, which sums them while it reads. */
int sc_fused2_csm_absim_parts_f32(const void* d_P, const sc_spectra_desc* desc, const float* d_scale, uint32_t planes,
                                  float* d_accum, void* d_workspace, int64_t workspace_bytes, int* n_parts, void* stream);
/* shader clock (GHz) the device sustained during the last sc_fused2_csm_absim_f32 launch (in-kernel cycle / real-time counters) */
int sc_debug_fused2_clock(double* ghz);
/* The diagnostic switches of the library (SC_FUSED_DEBUG, SC_FUSED_SPLIT, SC_MTFFT_DEBUG, SC_WILSON_FFT, ...: ablation tools and
 * the tests of alternative kernels) are environment variables read once when the library is loaded; a host that changes one
 * afterwards calls this to have them read again.  No reference counterpart (the reference has no native code). */
int sc_debug_reload_env(void);
/* rocFFT plans are never destroyed (a destroyed plan unloads run-time compiled code that a lazily loaded kernel of this library
 * has then executed stale: see sc_fft_plan_destroy); they live in pools keyed by geometry and are re-used.  *n_created:
 * rocfft_plan_create calls of the process; *n_pooled: plans the pools hold; *n_idle: of those, real-to-complex row plans that
 * belong to no live sc_fft_plan.  n_created == n_pooled always: what a process holds is bounded by the DISTINCT (length, batch,
 * precision) geometries it has asked for, not by how many sc_fft_plan objects it created and destroyed.  Any pointer may be NULL. */
int sc_debug_fft_plans(int64_t* n_created, int64_t* n_pooled, int64_t* n_idle);
int sc_fused2_csm_absim_f32(const void* d_P, const sc_spectra_desc* desc, const float* d_scale, uint32_t planes,
                            float* d_accum, void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- stage C: measures epilogue ------------------------------------------------------
 * Elementwise measure algebra on accumulated sums (connectivity.py:612-1159): divides by
 * n_observations AFTER any cross-GPU reduction, applies the reference's eps clamps, NaN /
 * zero diagonals and clips, mirrors the triangle into the full C x C matrix.
 * d_out: float [n_bins][C][C] (float2 for complex measures, float [n_bins][C] for power). */
int sc_measure_f32(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                   int64_t n_observations, int measure, void* d_out, void* stream);
/* The same measures written as double / double2 -- what the reference returns (float64 / complex128) -- from float
 * or (SC_RECORD_F64) double records: no widening pass over the result. */
int sc_measure_f64(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                   int64_t n_observations, int measure, void* d_out, void* stream);
/* Up to SC_MEASURE_MULTI_MAX real-valued C x C measures (coherence magnitude / phase, imaginary coherence, PLV, PPC,
 * PLI, wPLI and the debiased variants) of ONE record in one launch: the record is read once.  measures, d_outs: HOST
 * arrays of n_measures entries; d_outs[m]: float (_f32) or double (_f64) [n_bins][C][C]. */
#define SC_MEASURE_MULTI_MAX 4
int sc_measure_multi_f32(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                         int64_t n_observations, int n_measures, const int* measures, void* const* d_outs,
                         void* stream);
int sc_measure_multi_f64(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                         int64_t n_observations, int n_measures, const int* measures, void* const* d_outs,
                         void* stream);
/* The same, from a record that arrives as n_parts partial records (float, or double with SC_RECORD_F64 in `planes`): part 0
 * at d_part0, part k >= 1 at d_rest + (k - 1) * part_stride elements -- the bin blocks a rank received from the other ranks
 * in the direct exchange of the trial-sharded path (the sum over trial shards), or the per-workgroup partial
 * records of a split-bin stage B (sc_fused2_csm_absim_parts_f32).  The parts are summed in part order in the records'
 * precision while they are read: no summation pass, no second copy of the record.  wide: float64 outputs. */
int sc_measure_multi_parts(const void* d_part0, const void* d_rest, int n_parts, int64_t part_stride, int64_t n_bins,
                           int64_t n_signals, uint32_t planes, int64_t n_observations, int n_measures, const int* measures,
                           void* const* d_outs, int wide, void* stream);
/* ONE measure of sc_measure_f32 / _f64 -- power and the complex-valued ones included -- from partial records (same layout);
 * wide != 0: double / complex128 output. */
int sc_measure_parts(const void* d_part0, const void* d_rest, int n_parts, int64_t part_stride, int64_t n_bins,
                     int64_t n_signals, uint32_t planes, int64_t n_observations, int measure, void* d_out, int wide, void* stream);

/* ---- stage B of the float64 engine -----------------------------------------------------
 * Replaces the same reference code as sc_csm_accumulate_f32 / sc_nonlinear_accumulate_f32 (connectivity.py:447-526,
 * :897-1159, :1799-1822) in the reference's own arithmetic: complex128 spectra (any sc_spectra_desc layout, 16-byte
 * aligned), the cross-spectral matrix on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), the per-observation planes on
 * the fp64 VALU, double records of the same tile layout.  `which` names the planes to fill (subset of `planes`).
 * Selected by Connectivity(dtype=numpy.complex128), the reference's default dtype. */
int sc_accumulate_f64(const void* d_X /*double2*/, const sc_spectra_desc* desc, uint32_t planes, uint32_t which,
                      double* d_accum, void* stream);

/* ---- timing ------------------------------------------------------------------------------
 * hipEvent timers inside the library: after sc_timing_enable(1) every compute entry point brackets its launches with
 * two events on the stream it was given; sc_last_timing waits for them, writes (name, milliseconds) in call order
 * (at most max_entries) and forgets them.  sc_timing_enable(0) stops recording.  One timing session per process. */
typedef struct sc_timing {
    char name[48];
    float ms;
} sc_timing;
int sc_timing_enable(int on);
int sc_last_timing(sc_timing* out, int max_entries, int* n_entries);

/* ---- pairwise spectral Granger prediction (batched 2x2 Wilson factorisation, fp64) -----
 * Replaces Connectivity.pairwise_spectral_granger_prediction / subset_... and the Python
 * loop over pairs of _estimate_spectral_granger_prediction (connectivity.py:1161-1213,
 * :2282-2340) together with minimum_phase_decomposition (minimum_phase_decomposition.py:
 * 227-322), _estimate_transfer_function / _estimate_noise_covariance /
 * _remove_instantaneous_causality / _estimate_predictive_power (connectivity.py:1679-1779,
 * :1825-1848).  All (group, pair) problems advance together; FFTs along frequency are rocFFT
 * batched Z2Z plans created inside the call.
 *   d_accum     accumulator records with SC_PLANE_CSM, n_groups * n_freq_accum bins;
 *               n_freq_accum = N/2+1 (real input: negative bins are mirrored) or N.
 *   d_pairs     int32 [n_pairs][2] channel indices (i, j)
 *   d_out       double [n_groups][N/2+1][C][C]; out[.., i, j] = influence j -> i, NaN
 *               elsewhere (diagonal, pairs not requested, non-positive values)
 *   d_n_iter    int32 [n_groups*n_pairs] Wilson iterations used per problem
 *   d_status    int32 [n_groups*n_pairs]: 1 converged, 0 hit max_iter
 *   h_summary   optional HOST int32[3]: {iterations run, problems not converged, problems restarted because a lag-0
 *               covariance of their batch was not positive definite}.  The reference's batched Cholesky fails as a
 *               whole: every window of the channel pair then starts from the Cholesky factor of a random Wishart
 *               draw around a multiple of the identity (minimum_phase_decomposition.py:78-93, global NumPy generator);
 *               here every (group, pair) problem of such a pair starts from its expectation, the identity -- the
 *               start matters, the fixed point reached at a finite N depends on it (tests/golden/f12_*).
 *   flags       SC_GRANGER_KEEP_OUTPUT: do not NaN-fill d_out first (the caller walks a long pair list in chunks
 *               that share one output; any number of (group, pair) problems is accepted per call, the workspace
 *               is what grows: n_groups * n_pairs * N * 160 bytes)
 * Unlike every other entry point this one SYNCHRONISES the stream, once per four Wilson iterations (to stop
 * when every problem has converged, like the reference loop; converged problems are skipped by every kernel,
 * so the iterations queued past the last convergence are empty launches). */
#define SC_GRANGER_KEEP_OUTPUT 1
int sc_granger_workspace_bytes(int64_t n_groups, int64_t n_pairs, int64_t N, size_t* bytes);
int sc_granger_pairwise_f64(const void* d_accum, int64_t n_groups, int64_t n_freq_accum,
                            int64_t N, int64_t n_signals, uint32_t planes, int64_t n_observations,
                            const int32_t* d_pairs, int64_t n_pairs, double tolerance, int max_iterations,
                            void* d_work, size_t work_bytes, int flags, double* d_out, int32_t* d_n_iter,
                            int32_t* d_status, int32_t* h_summary, void* stream);

/* Minimum-phase (Wilson) factor of caller-supplied two-sided 2x2 Hermitian spectra: replaces
 * minimum_phase_decomposition() (minimum_phase_decomposition.py:227-322) for c <= 2 (a 1x1
 * spectrum is embedded as diag(s, 1)).  d_S: double [P][4][N] = (s00, s11, Re s01, Im s01) per bin;
 * d_G: complex128 [P][4][N] = (g00, g01, g10, g11), S = G G^H.  Workspace as for n_groups = 1,
 * n_pairs = P of sc_granger_workspace_bytes.  h_summary: HOST int32[3] as above.  Synchronises the stream (see above). */
int sc_wilson_factor_f64(const double* d_S, int64_t n_problems, int64_t N, double tolerance,
                         int max_iterations, void* d_work, size_t work_bytes, void* d_G /*complex128*/,
                         int32_t* d_n_iter, int32_t* d_status, int32_t* h_summary, void* stream);

/* ---- full C x C Wilson factor and the directed MVAR measures (fp64) ----------------------
 * Replaces Connectivity._minimum_phase_factor / _transfer_function / _noise_covariance /
 * _MVAR_Fourier_coefficients (connectivity.py:567-589), minimum_phase_decomposition() for any c
 * (minimum_phase_decomposition.py:227-322) and directed_transfer_function, directed_coherence,
 * partial_directed_coherence, generalized_partial_directed_coherence,
 * direct_directed_transfer_function (connectivity.py:1237-1426).  All windows iterate together,
 * converged windows are frozen; up to 128 signals the C x C factor of one (window, bin) lives in the registers of one
 * workgroup, 129 ... 512 run a panel-blocked inverse in global memory and products cut into 128 x 128 blocks:
 * n_signals <= sc_mvar_max_signals() (512 since round 6; records of more than 256 signals come from the planes-format
 * stage B or are assembled from channel-block pairs by the host), larger systems return SC_EUNSUPPORTED.
 * sc_mvar_factor_f64: exactly one of d_accum (accumulator records holding SC_PLANE_CSM, N or N/2+1
 * bins per window, real-input symmetry completes the rest) and d_S (complex128 [P][N][C][C], two-sided
 * Hermitian spectra) is non-NULL.  d_G: complex128 [P][N][C][C].  d_status[p]: 1 converged, 0 not
 * converged after max_iterations; h_summary = HOST int32[3] {iterations run, windows still running, windows whose
 * lag-0 covariance was not positive definite and that started from the identity (see sc_granger_pairwise_f64)}.
 * Synchronises the stream once per four iterations.  Per iteration: A = G^-1 S G^-H + I by a register-resident
 * Gauss-Jordan elimination per (window, bin) -- of [G | S] up to 64 signals; of G alone, followed by two matrix-core
 * products, for 65 ... 128 (five instead of three C^2 N complex128 arrays of workspace per window) --, the causal
 * transform pair along frequency, G <- G A+ on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
 * sc_mvar_measure_f64: d_G as above -> double [P][N/2+1][C][C] (SC_MVAR_DTF..DDTF), complex128
 * [P][N/2+1][C][C] (SC_MVAR_TRANSFER, SC_MVAR_COEFFICIENTS) or double [P][C][C] (NOISE_COVARIANCE).
 * The Tikhonov terms are the reference's: 1e-12 * mean(H0^2) over all windows, 1e-12 * mean(|H|^2)
 * over all windows and non-negative bins.  One workspace size serves both calls. */
#define SC_MVAR_DTF 0                /* connectivity.py:1237-1270 */
#define SC_MVAR_DC 1                 /* connectivity.py:1272-1309 */
#define SC_MVAR_PDC 2                /* connectivity.py:1311-1364 */
#define SC_MVAR_GPDC 3               /* connectivity.py:1366-1400 */
#define SC_MVAR_DDTF 4               /* connectivity.py:1402-1426 */
#define SC_MVAR_TRANSFER 5           /* connectivity.py:1712-1748 */
#define SC_MVAR_COEFFICIENTS 6       /* connectivity.py:581-589   */
#define SC_MVAR_NOISE_COVARIANCE 7   /* connectivity.py:1679-1709 */
int sc_mvar_max_signals(void);
int sc_mvar_workspace_bytes(int64_t n_groups, int64_t n_signals, int64_t N, size_t* bytes);
int sc_mvar_factor_f64(const void* d_accum, const void* d_S /*complex128*/, int64_t n_groups,
                       int64_t n_freq_accum, int64_t N, int64_t n_signals, uint32_t planes,
                       int64_t n_observations, double tolerance, int max_iterations, void* d_work,
                       size_t work_bytes, void* d_G /*complex128*/, int32_t* d_n_iter, int32_t* d_status,
                       int32_t* h_summary, void* stream);
int sc_mvar_measure_f64(const void* d_G /*complex128*/, int64_t n_groups, int64_t N, int64_t n_signals,
                        int which, void* d_out, void* d_work, size_t work_bytes, void* stream);

/* ---- global coherence (fp64, from the accumulated CSM) --------------------------------------
 * Replaces Connectivity.global_coherence / _estimate_global_coherence (connectivity.py:822-895,
 * :2245-2279): the leading squared singular values / n_estimates and left singular vectors of the
 * n_signals x (n_trials n_tapers) coefficient matrix per (window, two-sided bin) are the leading
 * eigenpairs of the cross-spectral matrix; parallel cyclic Jacobi in LDS, n_signals <=
 * sc_global_coherence_max_signals() (256; beyond 64 signals the eigenvectors come from a logged rotation
 * sequence applied to batches of unit vectors, any max_rank <= n_signals, and the call synchronises the stream).  d_accum: records with SC_PLANE_CSM accumulated over
 * trials and tapers, N or N/2+1 bins per window (real-input symmetry completes the rest).
 * d_values: double [n_groups][N][max_rank]; d_vectors: complex128 [n_groups][N][n_signals][max_rank],
 * unit norm, largest component real positive (the reference's phase is LAPACK's/ARPACK's).
 * ascending != 0 orders the max_rank largest values smallest-first (scipy svds, which the reference
 * takes when max_rank < n_signals - 1). */
int sc_global_coherence_max_signals(void);
int sc_global_coherence_f64(const void* d_accum, int64_t n_groups, int64_t n_freq_accum, int64_t N,
                            int64_t n_signals, uint32_t planes, int64_t n_observations, int max_rank,
                            int ascending, double* d_values, void* d_vectors /*complex128*/, void* stream);

/* ---- canonical coherence between channel groups (fp64, from the accumulated CSM) -------
 * Replaces Connectivity.canonical_coherence, _normalize_fourier_coefficients and
 * _estimate_canonical_coherence (connectivity.py:745-820, :1979-2032): per (bin, group pair)
 * the squared largest singular value of L_g^-1 S_gh L_h^-H with S_gg = L_g L_g^H.
 *   d_members   int32 [n_groups][stride] channel indices of every group, stride = 16 if
 *               max_group_size <= 16, 32 if <= 32, else 128 (max supported: sc_canonical_max_group() = 128)
 *   d_sizes     int32 [n_groups]
 *   d_out       double [n_bins][n_groups][n_groups], symmetric, NaN diagonal
 *   d_fail      int32 [1]: number of group blocks that were not positive definite
 * Groups of <= 16 channels take a stream-ordered workspace of n_bins * n_groups * 4 KB for the
 * inverted group factors, groups beyond 64 channels one of 768 KB per persistent workgroup (<= 256) for the blocks of
 * the pair in flight (hipMallocAsync / hipFreeAsync on `stream`); SC_ENOMEM if that fails.  17 ... 64 channels: a
 * workgroup per (bin, group pair) with the whole problem in LDS (round 6), no workspace. */
int sc_canonical_max_group(void);
int sc_canonical_coherence_f64(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                               int64_t n_observations, const int32_t* d_members, const int32_t* d_sizes,
                               int n_groups, int max_group_size, double* d_out, int32_t* d_fail,
                               void* stream);

/* ---- host-pointer side of the boundary: memory, copies, streams (sc_memory.hip, ABI v3) ---------------------
 * What the reference's CuPy backend does with `xp.asarray(time_series)` on the way in and `.get()` on the way out
 * (transforms.py:405-439, connectivity.py:31-65): with these a host that has only ctypes + NumPy drives the whole path
 * (spectral_connectivity_amd/numpy_host.py); a PyTorch host may keep its own allocator -- every compute entry point
 * takes raw device pointers from either.
 *   sc_device_alloc / sc_device_free   stream-ordered pool allocation (hipMallocAsync / hipFreeAsync on `stream`)
 *   sc_host_alloc / sc_host_free       page-locked host memory (copies at link rate, asynchronous on their stream)
 *   sc_host_register / _unregister     pin a buffer the host already owns (a NumPy array) for the same effect
 *   sc_memcpy_h2d / sc_memcpy_d2h      asynchronous on `stream` for page-locked host memory; from / to pageable memory
 *                                      the runtime stages the copy and returns when the host buffer is reusable
 *   sc_memset_zero, sc_stream_create (non-blocking), sc_stream_destroy, sc_stream_synchronize
 * sc_nonfinite_f32 / _f64: the constructor's NaN / infinity scan of the time series (transforms.py:746-753) on the
 * device: *d_flag |= 1 if any of the n samples is not finite (caller zeroes the flag, reads it after a sync). */
int sc_device_alloc(void** d_ptr, size_t bytes, void* stream);
int sc_device_free(void* d_ptr, void* stream);
int sc_host_alloc(void** h_ptr, size_t bytes);
int sc_host_free(void* h_ptr);
int sc_host_register(void* h_ptr, size_t bytes);
int sc_host_unregister(void* h_ptr);
int sc_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
int sc_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int sc_memset_zero(void* d_ptr, size_t bytes, void* stream);
int sc_stream_create(void** stream);
int sc_stream_destroy(void* stream);
int sc_stream_synchronize(void* stream);
int sc_nonfinite_f32(const float* d_x, int64_t n, int32_t* d_flag, void* stream);
int sc_nonfinite_f64(const double* d_x, int64_t n, int32_t* d_flag, void* stream);

/* ---- Exchange of trial-sharded records over RCCL / xGMI (SURVEY section 8(b) `sc_allreduce`, 8(e)) ------------------------
 * The reference has no multi-device path (its CuPy backend is one GPU: transforms.py:405-439, connectivity.py:31-65); trials
 * shard over one process per GPU because the expectation is a plain sum of per-observation terms (connectivity.py:67-75,
 * :489): every rank accumulates its trials into un-normalised records, the records are summed, sc_measure_* normalises by the
 * TOTAL observation count.  These calls are that sum for a host without torch.distributed (the PyTorch host's form of the same
 * steps is parallel.py; SC_EXCHANGE=library routes it through here):
 *   sc_comm_unique_id            128 bytes from ONE rank, carried to the others by the host (file, socket, MPI, torch ...)
 *   sc_comm_create               one communicator per process on the current device (ncclCommInitRank)
 *   sc_comm_exchange_blocks_f32  block j of d_send -> rank j, d_recv[i] <- rank i: all N - 1 links at once (the direct reduce-
 *                                scatter; sc_measure_multi_parts sums the received blocks in rank order while it reads them)
 *   sc_comm_allreduce_f32        in-place sum of a whole record buffer (ring form)
 *   sc_comm_gather_f32           n floats of every rank -> d_recv[rank][n] on root
 * RCCL is bound at run time (dlopen librccl.so.1): without it sc_comm_available() is 0 and the calls return SC_EUNSUPPORTED. */
typedef struct sc_comm sc_comm;
int sc_comm_available(void);
int sc_comm_unique_id(void* id128);
int sc_comm_create(const void* id128, int n_ranks, int rank, sc_comm** out);
int sc_comm_destroy(sc_comm* comm);
int sc_comm_size(const sc_comm* comm, int* n_ranks, int* rank);
int sc_comm_allreduce_f32(sc_comm* comm, float* d_buf, int64_t n, void* stream);
int sc_comm_exchange_blocks_f32(sc_comm* comm, const float* d_send, float* d_recv, int64_t block, void* stream);
int sc_comm_gather_f32(sc_comm* comm, const float* d_send, float* d_recv, int64_t n, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SC_HIP_H */
