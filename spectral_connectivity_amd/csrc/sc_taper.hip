// sc_taper.hip -- fused sliding-window extraction + detrend + DPSS taper multiply (any window length; the
// power-of-two lengths take sc_mtfft.hip instead and never materialise the tapered windows).
//
// One workgroup = one window w and 64 consecutive (trial, channel) columns of x[t][r*C+c]; lanes run along the
// channel-fastest axis for the loads, so every global load of a wave is one contiguous segment.  The 4 waves
// split the L samples of the window: pass 1 accumulates the trend sums in fp64 (a DC offset >> signal would
// otherwise leak through an fp32 mean), LDS combines the 4 partial sums.  Pass 2 re-reads the window (L2-
// resident) 64 samples at a time into an LDS tile, subtracts the trend, and writes the K tapered copies ROW-major,
//     y[w][r][k][c][n],   n fastest,
// with the lanes along n: rocFFT's unit-stride batched real transform runs at stream rate on that layout, while
// its strided plans (time-major y) spend 8x the transform's own time in gather / scatter kernels.
#include "sc_common.h"

// T = float (f32 engine) or double (f64 engine: the reference's own arithmetic, transforms.py:1377-1405)
template <typename T>
__global__ void __launch_bounds__(256)
taper_windows_kernel(const T* __restrict__ x, T* __restrict__ y,
                     const T* __restrict__ tapers, int64_t RC, int C, int K, int L, int step,
                     int W, int N, int detrend) {
    __shared__ double s_sum[4][64];
    __shared__ double s_sumt[4][64];
    __shared__ T tile[64][65];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: slice of L
    const int w = blockIdx.y;
    const int64_t rc0 = (int64_t)blockIdx.x * 64;
    const int64_t rc = rc0 + lane;
    const bool live = rc < RC;
    const int lq0 = (int)(((int64_t)L * q) / 4), lq1 = (int)(((int64_t)L * (q + 1)) / 4);
    const T* xw = x + (int64_t)w * step * RC + (live ? rc : 0);

    double a = 0.0, b = 0.0;  // trend = a * t_l + b, t_l = (l+1)/L
    if (detrend != SC_DETREND_NONE) {
        double sum = 0.0, sumt = 0.0;
        if (live) {
            for (int l = lq0; l < lq1; ++l) {
                double v = (double)xw[(int64_t)l * RC];
                sum += v;
                sumt += v * (double)(l + 1);
            }
        }
        s_sum[q][lane] = sum;
        s_sumt[q][lane] = sumt;
        __syncthreads();
        sum = s_sum[0][lane] + s_sum[1][lane] + s_sum[2][lane] + s_sum[3][lane];
        sumt = (s_sumt[0][lane] + s_sumt[1][lane] + s_sumt[2][lane] + s_sumt[3][lane]) / (double)L;
        const double n = (double)L;
        if (detrend == SC_DETREND_CONSTANT) {
            b = sum / n;
        } else {  // least-squares line on abscissa (l+1)/L  (transforms.py:1903-1909)
            const double St = (n + 1.0) * 0.5;                                   // sum t
            const double Stt = (n + 1.0) * (2.0 * n + 1.0) / (6.0 * n);          // sum t^2
            const double den = n * Stt - St * St;
            a = (den != 0.0) ? (n * sumt - St * sum) / den : 0.0;
            b = (sum - a * St) / n;
        }
    }
    const int64_t R = RC / C;
    const int nmax = L < N ? L : N;
    const double invL = 1.0 / (double)L;
    for (int l0 = 0; l0 < N; l0 += 64) {
        // 16 samples per wave into the tile, lanes along the columns
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int l = l0 + q * 16 + j;
            T v = (T)0;                                       // zero padding nmax <= n < N
            if (live && l < nmax) {
                const double t = (double)(l + 1) * invL;
                v = (T)((double)xw[(int64_t)l * RC] - (a * t + b));
            }
            tile[q * 16 + j][lane] = v;
        }
        __syncthreads();
        // 16 columns per wave out of the tile, lanes along n
        const int n = l0 + lane;
        if (n < N) {
            for (int k = 0; k < K; ++k) {
                const T h = (n < nmax) ? tapers[(int64_t)k * L + n] : (T)0;
                for (int jc = 0; jc < 16; ++jc) {
                    const int col = q * 16 + jc;
                    const int64_t rcc = rc0 + col;
                    if (rcc >= RC) break;
                    const int64_t r = rcc / C, c = rcc - r * C;
                    y[((((int64_t)w * R + r) * K + k) * C + c) * N + n] = tile[lane][col] * h;
                }
            }
        }
        __syncthreads();
    }
}

// The same with WIDE stores (round 6): the kernel above writes 256 (float) or 512 (double) contiguous bytes per wave and store -- one
// sample per lane -- into K x 16 different rows per step, and runs at 0.5 / 1.0 TB/s.  Here a step covers SPL = 64 V samples (V = 4
// floats or 2 doubles per lane: a 16-byte store), a wave's store is 1 KB of one row.  Needs N % V == 0 (rows stay 16-byte aligned).
template <typename T, int V>
__global__ void __launch_bounds__(256)
taper_windows_wide_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ tapers, int64_t RC, int C, int K, int L,
                          int step, int W, int N, int detrend) {
    constexpr int SPL = 64 * V;
    extern __shared__ __align__(16) unsigned char tw_smem[];
    T* tile = reinterpret_cast<T*>(tw_smem);                 // [SPL][65]
    __shared__ double s_sum[4][64];
    __shared__ double s_sumt[4][64];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.y;
    const int64_t rc0 = (int64_t)blockIdx.x * 64;
    const int64_t rc = rc0 + lane;
    const bool live = rc < RC;
    const int lq0 = (int)(((int64_t)L * q) / 4), lq1 = (int)(((int64_t)L * (q + 1)) / 4);
    const T* xw = x + (int64_t)w * step * RC + (live ? rc : 0);
    double a = 0.0, b = 0.0;
    if (detrend != SC_DETREND_NONE) {
        double sum = 0.0, sumt = 0.0;
        if (live) {
            for (int l = lq0; l < lq1; ++l) {
                double v = (double)xw[(int64_t)l * RC];
                sum += v;
                sumt += v * (double)(l + 1);
            }
        }
        s_sum[q][lane] = sum;
        s_sumt[q][lane] = sumt;
        __syncthreads();
        sum = s_sum[0][lane] + s_sum[1][lane] + s_sum[2][lane] + s_sum[3][lane];
        sumt = (s_sumt[0][lane] + s_sumt[1][lane] + s_sumt[2][lane] + s_sumt[3][lane]) / (double)L;
        const double n = (double)L;
        if (detrend == SC_DETREND_CONSTANT) {
            b = sum / n;
        } else {
            const double St = (n + 1.0) * 0.5;
            const double Stt = (n + 1.0) * (2.0 * n + 1.0) / (6.0 * n);
            const double den = n * Stt - St * St;
            a = (den != 0.0) ? (n * sumt - St * sum) / den : 0.0;
            b = (sum - a * St) / n;
        }
    }
    const int64_t R = RC / C;
    const int nmax = L < N ? L : N;
    const double invL = 1.0 / (double)L;
    typedef T TV __attribute__((ext_vector_type(V)));
    for (int l0 = 0; l0 < N; l0 += SPL) {
#pragma unroll 4
        for (int j = 0; j < SPL / 4; ++j) {                   // SPL / 4 samples per wave into the tile, lanes along the columns
            const int l = l0 + q * (SPL / 4) + j;
            T v = (T)0;
            if (live && l < nmax) {
                const double t = (double)(l + 1) * invL;
                v = (T)((double)xw[(int64_t)l * RC] - (a * t + b));
            }
            tile[(q * (SPL / 4) + j) * 65 + lane] = v;
        }
        __syncthreads();
        const int n = l0 + V * lane;                          // V consecutive samples per lane
        if (n < N) {
            for (int k = 0; k < K; ++k) {
                T h[V];
#pragma unroll
                for (int u = 0; u < V; ++u) h[u] = (n + u < nmax) ? tapers[(int64_t)k * L + n + u] : (T)0;
                for (int jc = 0; jc < 16; ++jc) {
                    const int col = q * 16 + jc;
                    const int64_t rcc = rc0 + col;
                    if (rcc >= RC) break;
                    const int64_t r = rcc / C, c = rcc - r * C;
                    TV o;
#pragma unroll
                    for (int u = 0; u < V; ++u) o[u] = (n + u < N) ? tile[(V * lane + u) * 65 + col] * h[u] : (T)0;
                    *reinterpret_cast<TV*>(y + ((((int64_t)w * R + r) * K + k) * C + c) * N + n) = o;
                }
            }
        }
        __syncthreads();
    }
}

template <typename Real>
static int taper_windows(const Real* d_x, int64_t T, int64_t R, int64_t C, int64_t L,
                         int64_t step, int64_t W, int64_t N, const Real* d_tapers,
                         int64_t K, int detrend_type, Real* d_y, void* stream) {
    ScTimed timed_("taper_windows", stream);
    SC_REQUIRE(d_x && d_tapers && d_y, "NULL device pointer");
    SC_REQUIRE(T >= 1 && R >= 1 && C >= 1 && L >= 1 && step >= 1 && W >= 1 && N >= 1 && K >= 1,
               "dimensions must be positive");
    SC_REQUIRE((W - 1) * step + L <= T, "windows exceed the time series");
    SC_REQUIRE(detrend_type >= 0 && detrend_type <= 2, "unknown detrend_type");
    SC_REQUIRE(W <= 65535, "too many windows for one launch");
    const int64_t RC = R * C;
    dim3 grid((unsigned)((RC + 63) / 64), (unsigned)W);
    constexpr int V = 16 / (int)sizeof(Real);
    // wide stores: 16 bytes a lane, 1 KB a wave -- from four steps a window on (float64: 300 / 384 samples 1.04 / 0.82 -> 1.26 / 0.94 ms
    // with it, 768 samples 0.54 -> 0.44, 4096 samples 1.94 -> 1.64; float32: 8192 samples 3.76 -> 2.77)
    if (N % V == 0 && N >= 4 * 64 * V && ((uintptr_t)d_y % 16) == 0) {
        const size_t lds = (size_t)64 * V * 65 * sizeof(Real);
        auto k = taper_windows_wide_kernel<Real, V>;
        SC_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, dim3(256), lds, (hipStream_t)stream, d_x, d_y,
                           d_tapers, RC, (int)C, (int)K, (int)L, (int)step, (int)W, (int)N, detrend_type);
    } else
    hipLaunchKernelGGL(taper_windows_kernel<Real>, grid, dim3(256), 0, (hipStream_t)stream, d_x, d_y,
                       d_tapers, RC, (int)C, (int)K, (int)L, (int)step, (int)W, (int)N, detrend_type);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

extern "C" int sc_taper_windows_f32(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L,
                                    int64_t step, int64_t W, int64_t N, const float* d_tapers,
                                    int64_t K, int detrend_type, float* d_y, void* stream) {
    return taper_windows<float>(d_x, T, R, C, L, step, W, N, d_tapers, K, detrend_type, d_y, stream);
}

extern "C" int sc_taper_windows_f64(const double* d_x, int64_t T, int64_t R, int64_t C, int64_t L,
                                    int64_t step, int64_t W, int64_t N, const double* d_tapers,
                                    int64_t K, int detrend_type, double* d_y, void* stream) {
    return taper_windows<double>(d_x, T, R, C, L, step, W, N, d_tapers, K, detrend_type, d_y, stream);
}

// ---- float64 time series -> the float32 copy the float32 engine transforms ------------------------------------------
// y[t][r][c] = (float)(x[t][r][c] - m[r][c]), m = the mean over time in fp64 when remove_mean (every window's own detrend
// removes any constant, so taking one out BEFORE the cast changes nothing but the rounding: a DC offset 1e5 times the signal
// -- raw EEG / MEG -- would otherwise cost the float32 copy all but two digits of the signal); channels C ... C_out - 1 of
// y are zero (the pad channel of odd channel counts).  One thread per (trial, channel): the loads of a wave are one
// contiguous run per time step; x is read twice (2 x 8 bytes per sample: 0.7 ms for 1 GB).
__global__ void __launch_bounds__(256) timeseries_to_f32_kernel(const double* __restrict__ x, float* __restrict__ y, int64_t T,
                                                                int64_t R, int64_t C, int64_t C_out, int remove_mean, int t_split) {
    const int64_t rc = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (trial, output channel)
    if (rc >= R * C_out) return;
    const int64_t r = rc / C_out, c = rc - r * C_out;
    const int64_t t0 = (int64_t)blockIdx.y * T / t_split, t1 = (int64_t)(blockIdx.y + 1) * T / t_split;
    if (c >= C) {
        for (int64_t t = t0; t < t1; ++t) y[(t * R + r) * C_out + c] = 0.f;
        return;
    }
    double m = 0.0;
    if (remove_mean) {
        for (int64_t t = 0; t < T; ++t) m += x[(t * R + r) * C + c];      // every time slice sums the whole series: same mean
        m /= (double)T;
        // one NaN / infinite sample must cost only the windows that contain it (the reference's per-window detrend, and
        // the float64 engine): a non-finite mean is not removed -- the per-window detrend takes the constant out anyway
        if (!(fabs(m) <= 1.7976931348623157e308)) m = 0.0;
    }
    for (int64_t t = t0; t < t1; ++t) y[(t * R + r) * C_out + c] = (float)(x[(t * R + r) * C + c] - m);
}

extern "C" int sc_timeseries_to_f32(const double* d_x, int64_t T, int64_t R, int64_t C, int remove_mean, float* d_y,
                                    int64_t C_out, void* stream) {
    ScTimed timed_("timeseries_to_f32", stream);
    SC_REQUIRE(d_x && d_y, "NULL device pointer");
    SC_REQUIRE(T >= 1 && R >= 1 && C >= 1 && C_out >= C, "bad dimensions");
    const int64_t blocks = (R * C_out + 255) / 256;
    SC_REQUIRE(blocks < (int64_t)1 << 31, "too many (trial, channel) series for one launch");
    // few series (a handful of trials x channels): the time axis is cut into slices so that the launch still fills the chip
    int t_split = 1;
    while (blocks * t_split < 512 && t_split * 64 <= T && t_split < 64) t_split *= 2;      // (every slice re-reads the series for the mean)
    hipLaunchKernelGGL(timeseries_to_f32_kernel, dim3((unsigned)blocks, (unsigned)t_split), dim3(256), 0, (hipStream_t)stream,
                       d_x, d_y, T, R, C, C_out, remove_mean, t_split);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
