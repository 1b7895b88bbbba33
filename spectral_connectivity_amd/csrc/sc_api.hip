// sc_api.hip -- library housekeeping, rocFFT plans, accumulator layout.
#include <stdarg.h>
#include <stdlib.h>
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

// ---- diagnostic switches: a snapshot of the environment (sc_common.h) ------------------------------------------------------
static const char* const g_switch_names[SC_SW_COUNT] = {
    "SC_FUSED_DEBUG", "SC_FUSED2_TERMS", "SC_FUSED_SPLIT", "SC_FUSED_NO_SMALL", "SC_MTFFT_DEBUG", "SC_MTFFT_WIDE", "SC_MTFFT_F64",
    "SC_F64_SPLIT", "SC_F64_OC", "SC_F64_NO_FORK", "SC_F64_NO_BLOCK", "SC_WILSON_FFT", "SC_GLOBAL_EIG", "SC_GLOBAL_NT256",
    "SC_GRANGER_KERNEL", "SC_FUSED_FOLD_OBS", "SC_MTFFT_LONG", "SC_CANON_EIG", "SC_MTFFT_MIXED", "SC_MTFFT_MIXED_GEO", "SC_MTFFT_SLICE", "SC_MVAR_INVERSE"};
static char g_switch_val[SC_SW_COUNT][32];
static bool g_switch_set[SC_SW_COUNT];
extern "C" int sc_debug_reload_env(void) {
    for (int i = 0; i < SC_SW_COUNT; ++i) {
        const char* v = getenv(g_switch_names[i]);
        g_switch_set[i] = v != nullptr;
        if (v) { strncpy(g_switch_val[i], v, sizeof(g_switch_val[i]) - 1); g_switch_val[i][sizeof(g_switch_val[i]) - 1] = 0; }
    }
    return SC_OK;
}
static const int g_switches_loaded = sc_debug_reload_env();          // when the library is loaded
const char* sc_switch(int id) { return (id >= 0 && id < SC_SW_COUNT && g_switch_set[id]) ? g_switch_val[id] : nullptr; }

// Batched in-place complex128 transforms of the Wilson kernels (sc_wilson.hip, sc_mvar.hip), one plan per (direction, length, batch)
// for the life of the process.  rocFFT compiles the kernels of a length outside its prebuilt set at run time and unloads their code
// object when the plan is destroyed; a plan destroyed right before the FIRST launch of one of this library's own kernels (whose code
// object the runtime loads lazily) was followed, one run in five on MI355X / ROCm 7.0, by "illegal shader instruction" + a write to
// address 0 in that kernel -- stale instructions where the unloaded code had been (pairwise Granger at 250 samples, then canonical
// coherence: tests/test_gpu_fp64.py run first in a process).  Plans that are never destroyed never free code memory, and the second
// call with a geometry pays no plan creation (tens of ms with run-time compilation).
#include <mutex>
#include <vector>
struct ScZ2zEntry { int type; size_t N, batch; rocfft_plan plan; };
static std::vector<ScZ2zEntry> g_z2z;              // grows with the distinct geometries of the process: never a destroyed plan
static std::mutex g_z2z_mutex;
static int64_t g_plans_created = 0;                // rocfft_plan_create calls of the process (both pools), for sc_debug_fft_plans
int sc_internal_z2z_plan(rocfft_plan* plan, int forward, size_t N, size_t batch, bool* cached) {
    std::lock_guard<std::mutex> lock(g_z2z_mutex);
    for (const ScZ2zEntry& e : g_z2z)
        if (e.type == forward && e.N == N && e.batch == batch) { *plan = e.plan; *cached = true; return SC_OK; }
    size_t lengths[1] = {N};
    const rocfft_status s = rocfft_plan_create(plan, rocfft_placement_inplace,
                                               forward ? rocfft_transform_type_complex_forward : rocfft_transform_type_complex_inverse,
                                               rocfft_precision_double, 1, lengths, batch, nullptr);
    if (s != rocfft_status_success) {
        sc_set_error("rocfft_plan_create(Z2Z N=%zu batch=%zu) failed: %d", N, batch, (int)s);
        return SC_EFFT;
    }
    ++g_plans_created;
    *cached = true;                                  // (always: the callers' "not cached -> destroy" branch is never taken)
    g_z2z.push_back(ScZ2zEntry{forward, N, batch, *plan});
    return SC_OK;
}

// The real-to-complex row plans of sc_fft_plan (the transform of window lengths no fused kernel has): a pool keyed by (length, rows,
// precision).  A plan whose sc_fft_plan is destroyed goes back to the pool -- same hazard as above: never rocfft_plan_destroy -- and
// the next sc_fft_plan of that geometry takes it instead of creating (and compiling) another: what the pool holds is bounded by the
// DISTINCT geometries a process ever asks for, not by how often the host's plan cache evicts one (round 5 dropped the handle on
// every eviction: an unbounded leak of twiddle tables and code).
struct ScR2cEntry { int64_t N, rows; int f64; rocfft_plan plan; bool busy; };
static std::vector<ScR2cEntry> g_r2c;
static rocfft_plan r2c_pool_take(int64_t N, int64_t rows, bool f64) {
    std::lock_guard<std::mutex> lock(g_z2z_mutex);
    for (ScR2cEntry& e : g_r2c)
        if (!e.busy && e.N == N && e.rows == rows && e.f64 == (f64 ? 1 : 0)) { e.busy = true; return e.plan; }
    return nullptr;
}
static void r2c_pool_add(int64_t N, int64_t rows, bool f64, rocfft_plan plan) {
    std::lock_guard<std::mutex> lock(g_z2z_mutex);
    ++g_plans_created;
    g_r2c.push_back(ScR2cEntry{N, rows, f64 ? 1 : 0, plan, true});
}
static void r2c_pool_release(rocfft_plan plan) {
    if (!plan) return;
    std::lock_guard<std::mutex> lock(g_z2z_mutex);
    for (ScR2cEntry& e : g_r2c)
        if (e.plan == plan) { e.busy = false; return; }
}
extern "C" int sc_debug_fft_plans(int64_t* n_created, int64_t* n_pooled, int64_t* n_idle) {
    std::lock_guard<std::mutex> lock(g_z2z_mutex);
    int64_t idle = 0;
    for (const ScR2cEntry& e : g_r2c) idle += e.busy ? 0 : 1;
    if (n_created) *n_created = g_plans_created;
    if (n_pooled) *n_pooled = (int64_t)(g_r2c.size() + g_z2z.size());
    if (n_idle) *n_idle = idle;
    return SC_OK;
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
// A plan transforms `chunk` rows at a time (unit stride on both sides: the layout rocFFT streams at full rate)
// into a scratch Z[chunk][F] it owns, and a tiled transpose moves every chunk into the frequency-major X.  The
// scratch is sized to stay inside the 256 MB memory-side cache, so the transpose reads it back without touching HBM.
struct sc_fft_plan {
    rocfft_plan plan, tail_plan;          // chunk rows; the last batch % chunk rows
    rocfft_execution_info info;
    void* work;                           // rocFFT work buffer (the larger of the two plans')
    void* Z;                              // [chunk][F] float2 (or double2: f64 plans)
    size_t work_bytes;
    int64_t N, batch, chunk, tail;
    int f64;                              // double-precision plan (the f64 engine)
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

static int make_r2c_rows(rocfft_plan* plan, int64_t N, int64_t rows, bool f64) {
    if ((*plan = r2c_pool_take(N, rows, f64)) != nullptr) return SC_OK;
    rocfft_plan_description desc = nullptr;
    SC_CHECK_FFT(rocfft_plan_description_create(&desc));
    size_t one[1] = {1};
    const size_t F = (size_t)(N / 2 + 1);
    SC_CHECK_FFT(rocfft_plan_description_set_data_layout(
        desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr, nullptr,
        1, one, (size_t)N, 1, one, F));
    size_t lengths[1] = {(size_t)N};
    const rocfft_status s = rocfft_plan_create(plan, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                               f64 ? rocfft_precision_double : rocfft_precision_single, 1, lengths,
                                               (size_t)rows, desc);
    rocfft_plan_description_destroy(desc);
    if (s != rocfft_status_success) {
        sc_set_error("rocfft_plan_create(N=%lld, rows=%lld) failed: status %d", (long long)N, (long long)rows, (int)s);
        *plan = nullptr;
        return SC_EFFT;
    }
    r2c_pool_add(N, rows, f64, *plan);
    return SC_OK;
}

extern "C" int sc_fft_plan_destroy(sc_fft_plan* plan);

static int fft_plan_create(sc_fft_plan** out, int64_t N, int64_t batch, bool f64) {
    SC_REQUIRE(out != nullptr, "plan out pointer is NULL");
    SC_REQUIRE(N >= 1 && batch >= 1, "N and batch must be positive");
    if (!g_rocfft_ready) { SC_CHECK_FFT(rocfft_setup()); g_rocfft_ready = 1; }
    sc_fft_plan* p = new sc_fft_plan();
    memset(p, 0, sizeof(*p));
    p->N = N; p->batch = batch; p->f64 = f64 ? 1 : 0;
    const int64_t F = N / 2 + 1;
    const size_t zsize = f64 ? sizeof(double2) : sizeof(float2);
    int64_t chunk = ((int64_t)(64u << 20) / (F * (int64_t)zsize)) & ~(int64_t)63;      // 64 MB of Z, whole 64-row tiles
    if (chunk < 64) chunk = 64;
    if (chunk > batch) chunk = batch;
    p->chunk = chunk;
    p->tail = batch % chunk;
    int rc = make_r2c_rows(&p->plan, N, chunk, f64);
    if (rc == SC_OK && p->tail) rc = make_r2c_rows(&p->tail_plan, N, p->tail, f64);
    if (rc != SC_OK) { sc_fft_plan_destroy(p); return rc; }
    size_t w1 = 0, w2 = 0;
    if (rocfft_plan_get_work_buffer_size(p->plan, &w1) != rocfft_status_success ||
        (p->tail_plan && rocfft_plan_get_work_buffer_size(p->tail_plan, &w2) != rocfft_status_success) ||
        rocfft_execution_info_create(&p->info) != rocfft_status_success) {
        sc_set_error("rocFFT work-buffer query failed (N=%lld)", (long long)N);
        sc_fft_plan_destroy(p);
        return SC_EFFT;
    }
    p->work_bytes = w1 > w2 ? w1 : w2;
    if ((p->work_bytes && hipMalloc(&p->work, p->work_bytes) != hipSuccess) ||
        hipMalloc((void**)&p->Z, (size_t)chunk * F * zsize) != hipSuccess) {
        sc_set_error("hipMalloc of the FFT plan's buffers failed (%zu + %zu bytes)", p->work_bytes,
                     (size_t)chunk * F * zsize);
        sc_fft_plan_destroy(p);
        return SC_ENOMEM;
    }
    if (p->work_bytes && rocfft_execution_info_set_work_buffer(p->info, p->work, p->work_bytes) != rocfft_status_success) {
        sc_set_error("rocfft_execution_info_set_work_buffer failed");
        sc_fft_plan_destroy(p);
        return SC_EFFT;
    }
    *out = p;
    return SC_OK;
}

extern "C" int sc_fft_plan_create(sc_fft_plan** out, int64_t N, int64_t batch) { return fft_plan_create(out, N, batch, false); }
extern "C" int sc_fft_plan_create_f64(sc_fft_plan** out, int64_t N, int64_t batch) { return fft_plan_create(out, N, batch, true); }

extern "C" int sc_fft_plan_work_bytes(const sc_fft_plan* plan, size_t* bytes) {
    SC_REQUIRE(plan && bytes, "NULL argument");
    *bytes = plan->work_bytes + (size_t)plan->chunk * (plan->N / 2 + 1) * (plan->f64 ? sizeof(double2) : sizeof(float2));
    return SC_OK;
}

// Z[rows][F] -> X[f][b_off + row] through a 64 x 32 LDS tile: 256-byte reads along f, 512-byte writes along the batch.
// The DC and (even N) Nyquist coefficients of a real sequence are exactly real.  The reference's transform
// (and the fused FFT kernels here) return them so, which is what makes PLI / wPLI exactly 0 at those bins
// (connectivity.py:982-1028: weights < eps -> 1); a generic R2C leaves rounding noise in the imaginary part,
// and sum Im / sum |Im| of noise is O(1): the transpose zeroes it on the way through.
template <typename T2>
__global__ void __launch_bounds__(256) rows_to_bins_kernel(const T2* __restrict__ Z, T2* __restrict__ X, int64_t rows,
                                                            int64_t F, int64_t batch, int64_t b_off, int64_t nyquist_row) {
    __shared__ T2 tile[64][33];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64, f0 = (int64_t)blockIdx.y * 32;
    {
        const int fi = t & 31, ri = t >> 5;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t r = r0 + ri + 8 * j, f = f0 + fi;
            if (r < rows && f < F) tile[ri + 8 * j][fi] = Z[r * F + f];
        }
    }
    __syncthreads();
    {
        const int ri = t & 63, fi = t >> 6;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t f = f0 + fi + 4 * j, r = r0 + ri;
            if (r < rows && f < F) {
                T2 v = tile[ri][fi + 4 * j];
                if (f == 0 || f == nyquist_row) v.y = 0;
                X[f * batch + b_off + r] = v;
            }
        }
    }
}

int sc_internal_rows_to_bins(const void* d_Z, void* d_X, int64_t rows, int64_t F, int64_t batch, int64_t b_off,
                             int64_t nyquist_row, hipStream_t st) {
    hipLaunchKernelGGL(rows_to_bins_kernel<float2>, dim3((unsigned)((rows + 63) / 64), (unsigned)((F + 31) / 32)), dim3(256), 0, st,
                       (const float2*)d_Z, (float2*)d_X, rows, F, batch, b_off, nyquist_row);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

static int fft_execute(sc_fft_plan* plan, const void* d_y, void* d_X, bool f64, void* stream) {
    ScTimed timed_("fft_execute", stream);
    SC_REQUIRE(plan && d_y && d_X, "NULL argument");
    SC_REQUIRE((plan->f64 != 0) == f64, "plan precision does not match the call");
    SC_CHECK_FFT(rocfft_execution_info_set_stream(plan->info, stream));
    const int64_t F = plan->N / 2 + 1;
    const int64_t nyq = (plan->N % 2 == 0) ? plan->N / 2 : -1;
    for (int64_t off = 0; off < plan->batch; off += plan->chunk) {
        const int64_t rows = plan->batch - off < plan->chunk ? plan->batch - off : plan->chunk;
        void* in[1] = {(void*)((const char*)d_y + off * plan->N * (f64 ? 8 : 4))};
        void* outb[1] = {plan->Z};
        SC_CHECK_FFT(rocfft_execute(rows == plan->chunk ? plan->plan : plan->tail_plan, in, outb, plan->info));
        const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((F + 31) / 32));
        if (f64)
            hipLaunchKernelGGL(rows_to_bins_kernel<double2>, grid, dim3(256), 0, (hipStream_t)stream, (const double2*)plan->Z,
                               (double2*)d_X, rows, F, plan->batch, off, nyq);
        else
            hipLaunchKernelGGL(rows_to_bins_kernel<float2>, grid, dim3(256), 0, (hipStream_t)stream, (const float2*)plan->Z,
                               (float2*)d_X, rows, F, plan->batch, off, nyq);
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

extern "C" int sc_fft_execute(sc_fft_plan* plan, const float* d_y, void* d_X, void* stream) {
    return fft_execute(plan, d_y, d_X, false, stream);
}
extern "C" int sc_fft_execute_f64(sc_fft_plan* plan, const double* d_y, void* d_X, void* stream) {
    return fft_execute(plan, d_y, d_X, true, stream);
}

// Releases what a plan owns in device memory (work buffer, transform scratch: up to tens of MB).  The rocFFT plan objects themselves
// go back to the pool above, not to rocfft_plan_destroy: destroying a plan whose kernels rocFFT compiled at run time unloads their
// code object, and a kernel of this library launched for the first time right after that has run stale instructions there (see
// sc_internal_z2z_plan above: "illegal shader instruction", one fresh process in five).  A pooled plan keeps its twiddle tables and
// code (KBs to a few MB) and serves the next sc_fft_plan of its geometry.
extern "C" int sc_fft_plan_destroy(sc_fft_plan* plan) {
    if (!plan) return SC_OK;
    r2c_pool_release(plan->plan);
    r2c_pool_release(plan->tail_plan);
    if (plan->info) rocfft_execution_info_destroy(plan->info);
    if (plan->work) (void)hipFree(plan->work);
    if (plan->Z) (void)hipFree(plan->Z);
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
