// sc_api.hip -- library housekeeping, rocFFT plans, accumulator layout.
#include <stdarg.h>
#include <string.h>
#include <rocfft/rocfft.h>
#include "sc_common.h"

static thread_local char g_err[512] = "";

void sc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sc_abi_version(void) { return SC_ABI_VERSION; }
extern "C" const char* sc_last_error(void) { return g_err; }

extern "C" int sc_device_count(int* count) {
    SC_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return SC_OK;
}

// ------------------------------------------------------------------------------- rocFFT
struct sc_fft_plan {
    rocfft_plan plan;
    rocfft_execution_info info;
    void* work;
    size_t work_bytes;
    int64_t N, batch;
};

static int g_rocfft_ready = 0;

#define SC_CHECK_FFT(expr)                                                          \
    do {                                                                            \
        rocfft_status s_ = (expr);                                                  \
        if (s_ != rocfft_status_success) {                                          \
            sc_set_error("%s failed: rocfft_status %d (%s:%d)", #expr, (int)s_,     \
                         __FILE__, __LINE__);                                       \
            return SC_EFFT;                                                         \
        }                                                                           \
    } while (0)

extern "C" int sc_fft_plan_create(sc_fft_plan** out, int64_t N, int64_t batch) {
    SC_REQUIRE(out != nullptr, "plan out pointer is NULL");
    SC_REQUIRE(N >= 1 && batch >= 1, "N and batch must be positive");
    if (!g_rocfft_ready) { SC_CHECK_FFT(rocfft_setup()); g_rocfft_ready = 1; }
    sc_fft_plan* p = new sc_fft_plan();
    memset(p, 0, sizeof(*p));
    p->N = N; p->batch = batch;
    rocfft_plan_description desc = nullptr;
    SC_CHECK_FFT(rocfft_plan_description_create(&desc));
    // both sides "batch fastest": element stride = batch, distance between transforms = 1
    size_t stride[1] = {(size_t)batch};
    SC_CHECK_FFT(rocfft_plan_description_set_data_layout(
        desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr,
        1, stride, 1, 1, stride, 1));
    size_t lengths[1] = {(size_t)N};
    rocfft_status s = rocfft_plan_create(&p->plan, rocfft_placement_notinplace,
                                         rocfft_transform_type_real_forward, rocfft_precision_single,
                                         1, lengths, (size_t)batch, desc);
    rocfft_plan_description_destroy(desc);
    if (s != rocfft_status_success) {
        sc_set_error("rocfft_plan_create(N=%lld, batch=%lld) failed: status %d", (long long)N,
                     (long long)batch, (int)s);
        delete p;
        return SC_EFFT;
    }
    SC_CHECK_FFT(rocfft_plan_get_work_buffer_size(p->plan, &p->work_bytes));
    SC_CHECK_FFT(rocfft_execution_info_create(&p->info));
    if (p->work_bytes) {
        if (hipMalloc(&p->work, p->work_bytes) != hipSuccess) {
            sc_set_error("hipMalloc of %zu-byte rocFFT work buffer failed", p->work_bytes);
            rocfft_execution_info_destroy(p->info);
            rocfft_plan_destroy(p->plan);
            delete p;
            return SC_ENOMEM;
        }
        SC_CHECK_FFT(rocfft_execution_info_set_work_buffer(p->info, p->work, p->work_bytes));
    }
    *out = p;
    return SC_OK;
}

extern "C" int sc_fft_plan_work_bytes(const sc_fft_plan* plan, size_t* bytes) {
    SC_REQUIRE(plan && bytes, "NULL argument");
    *bytes = plan->work_bytes;
    return SC_OK;
}

// The DC and (even N) Nyquist coefficients of a real sequence are exactly real.  The reference's transform
// (and the fused FFT kernels here) return them so, which is what makes PLI / wPLI exactly 0 at those bins
// (connectivity.py:982-1028: weights < eps -> 1); a generic R2C leaves rounding noise in the imaginary part,
// and sum Im / sum |Im| of noise is O(1).  X is [F][batch], row f = frequency bin.
__global__ void real_bins_kernel(float2* X, int64_t batch, int64_t nyquist_row) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    X[b].y = 0.f;
    if (nyquist_row > 0) X[nyquist_row * batch + b].y = 0.f;
}

extern "C" int sc_fft_execute(sc_fft_plan* plan, const float* d_y, void* d_X, void* stream) {
    SC_REQUIRE(plan && d_y && d_X, "NULL argument");
    SC_CHECK_FFT(rocfft_execution_info_set_stream(plan->info, stream));
    void* in[1] = {(void*)d_y};
    void* outb[1] = {d_X};
    SC_CHECK_FFT(rocfft_execute(plan->plan, in, outb, plan->info));
    const int64_t nyq = (plan->N % 2 == 0) ? plan->N / 2 : 0;
    hipLaunchKernelGGL(real_bins_kernel, dim3((unsigned)((plan->batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (float2*)d_X, plan->batch, nyq);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

extern "C" int sc_fft_plan_destroy(sc_fft_plan* plan) {
    if (!plan) return SC_OK;
    if (plan->info) rocfft_execution_info_destroy(plan->info);
    if (plan->plan) rocfft_plan_destroy(plan->plan);
    if (plan->work) (void)hipFree(plan->work);
    delete plan;
    return SC_OK;
}

// --------------------------------------------------------------------------- accumulators
extern "C" int sc_accum_layout(const sc_spectra_desc* desc, uint32_t planes, int64_t* n_bins,
                               int64_t* floats_per_bin, int64_t* n_groups, int64_t* n_observations) {
    SC_REQUIRE(desc != nullptr, "desc is NULL");
    ScAxes a;
    sc_make_axes(desc, &a);
    SC_REQUIRE(a.C >= 1 && a.F >= 1 && a.W >= 1 && a.R >= 1 && a.K >= 1, "empty dimension");
    int nb = sc_n_blocks(a.C);
    if (n_bins) *n_bins = (int64_t)a.n_groups * a.F;
    if (floats_per_bin) *floats_per_bin = (int64_t)sc_plane_count(planes) * sc_n_tiles(nb) * SC_TILE_ELEMS;
    if (n_groups) *n_groups = a.n_groups;
    if (n_observations) *n_observations = a.n_obs;
    return SC_OK;
}
