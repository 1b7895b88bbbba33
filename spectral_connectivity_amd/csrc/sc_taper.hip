// sc_taper.hip -- fused sliding-window extraction + detrend + DPSS taper multiply.
//
// One workgroup = one window w and 64 consecutive (trial, channel) columns of x[t][r*C+c];
// lanes run along the channel-fastest axis, so every global load/store of a wave is one
// contiguous segment (x is (T,R,C) with C fastest -- no transpose, no LDS staging needed).
// The 4 waves of the group split the L samples of the window: pass 1 accumulates the trend
// sums in fp64 (a DC offset >> signal would otherwise leak through an fp32 mean), LDS
// combines the 4 partial sums, pass 2 re-reads its slice (L2-resident), subtracts the trend
// and writes the K tapered copies time-major: y[n][w][r][k][c].
#include "sc_common.h"

__global__ void __launch_bounds__(256)
taper_windows_kernel(const float* __restrict__ x, float* __restrict__ y,
                     const float* __restrict__ tapers, int64_t RC, int C, int K, int L, int step,
                     int W, int N, int detrend) {
    __shared__ double s_sum[4][64];
    __shared__ double s_sumt[4][64];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id: slice of L
    const int w = blockIdx.y;
    const int64_t rc = (int64_t)blockIdx.x * 64 + lane;
    const bool live = rc < RC;
    const int lq0 = (int)(((int64_t)L * q) / 4), lq1 = (int)(((int64_t)L * (q + 1)) / 4);
    const float* xw = x + (int64_t)w * step * RC + (live ? rc : 0);

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
    if (!live) return;
    const int r = (int)(rc / C), c = (int)(rc % C);
    const int64_t WRKC = (int64_t)W * (RC / C) * K * C;   // elements per time sample of y
    float* yb = y + ((int64_t)w * (RC / C) + r) * K * C + c;
    const int nmax = L < N ? L : N;
    const double invL = 1.0 / (double)L;
    for (int l = lq0; l < lq1 && l < nmax; ++l) {
        const double t = (double)(l + 1) * invL;
        const float v = (float)((double)xw[(int64_t)l * RC] - (a * t + b));
        float* yl = yb + (int64_t)l * WRKC;
        for (int k = 0; k < K; ++k) yl[(int64_t)k * C] = v * tapers[(int64_t)k * L + l];
    }
    // zero padding L <= n < N, split over the 4 waves like the samples
    if (N > L) {
        const int pad = N - L;
        const int p0 = L + (int)(((int64_t)pad * q) / 4), p1 = L + (int)(((int64_t)pad * (q + 1)) / 4);
        for (int n = p0; n < p1; ++n) {
            float* yl = yb + (int64_t)n * WRKC;
            for (int k = 0; k < K; ++k) yl[(int64_t)k * C] = 0.0f;
        }
    }
}

extern "C" int sc_taper_windows_f32(const float* d_x, int64_t T, int64_t R, int64_t C, int64_t L,
                                    int64_t step, int64_t W, int64_t N, const float* d_tapers,
                                    int64_t K, int detrend_type, float* d_y, void* stream) {
    SC_REQUIRE(d_x && d_tapers && d_y, "NULL device pointer");
    SC_REQUIRE(T >= 1 && R >= 1 && C >= 1 && L >= 1 && step >= 1 && W >= 1 && N >= 1 && K >= 1,
               "dimensions must be positive");
    SC_REQUIRE((W - 1) * step + L <= T, "windows exceed the time series");
    SC_REQUIRE(detrend_type >= 0 && detrend_type <= 2, "unknown detrend_type");
    SC_REQUIRE(W <= 65535, "too many windows for one launch");
    const int64_t RC = R * C;
    dim3 grid((unsigned)((RC + 63) / 64), (unsigned)W);
    hipLaunchKernelGGL(taper_windows_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_x, d_y,
                       d_tapers, RC, (int)C, (int)K, (int)L, (int)step, (int)W, (int)N, detrend_type);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
