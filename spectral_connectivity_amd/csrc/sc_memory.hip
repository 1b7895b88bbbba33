// sc_memory.hip -- the host-pointer side of the boundary (ownership row): device and page-locked
// host memory, copies and streams as plain C calls, so that a host with nothing but ctypes + NumPy can drive the
// engine (spectral_connectivity_amd/numpy_host.py, tests/test_abi.py) -- the reference's `xp.asarray(...)` upload and
// `.get()` download of its CuPy backend (transforms.py:405-439, connectivity.py:31-65).  PyTorch hosts keep using
// their own allocator and streams: every compute entry point takes raw device pointers from either.
#include "sc_common.h"

// The default pool hands freed memory back to the driver at the next synchronisation (release threshold 0): every
// allocation of a steady-state loop would then be a real hipMalloc.  Keep what was freed (one setting per device).
static void keep_pool_memory() {
    static bool done[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || done[dev]) return;
    done[dev] = true;
    hipMemPool_t pool = nullptr;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
        uint64_t keep = UINT64_MAX;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    (void)hipGetLastError();
}

extern "C" int sc_device_alloc(void** d_ptr, size_t bytes, void* stream) {
    SC_REQUIRE(d_ptr != nullptr, "NULL argument");
    *d_ptr = nullptr;
    if (bytes == 0) return SC_OK;
    keep_pool_memory();
    // stream-ordered pool allocation: a freed block is handed out again without a trip to the driver
    hipError_t e = hipMallocAsync(d_ptr, bytes, (hipStream_t)stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        sc_set_error("sc_device_alloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        *d_ptr = nullptr;
        return e == hipErrorOutOfMemory ? SC_ENOMEM : SC_EHIP;
    }
    return SC_OK;
}

extern "C" int sc_device_free(void* d_ptr, void* stream) {
    if (!d_ptr) return SC_OK;
    SC_CHECK_HIP(hipFreeAsync(d_ptr, (hipStream_t)stream));
    return SC_OK;
}

extern "C" int sc_host_alloc(void** h_ptr, size_t bytes) {
    SC_REQUIRE(h_ptr != nullptr, "NULL argument");
    *h_ptr = nullptr;
    if (bytes == 0) return SC_OK;
    hipError_t e = hipHostMalloc(h_ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        sc_set_error("sc_host_alloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        *h_ptr = nullptr;
        return SC_ENOMEM;
    }
    return SC_OK;
}

extern "C" int sc_host_free(void* h_ptr) {
    if (!h_ptr) return SC_OK;
    SC_CHECK_HIP(hipHostFree(h_ptr));
    return SC_OK;
}

// Pinning a buffer the host already owns (a NumPy array): the copy engines then read / write it directly, no staging
// copy through the runtime's own bounce buffers.
extern "C" int sc_host_register(void* h_ptr, size_t bytes) {
    SC_REQUIRE(h_ptr != nullptr && bytes > 0, "NULL argument");
    SC_CHECK_HIP(hipHostRegister(h_ptr, bytes, hipHostRegisterDefault));
    return SC_OK;
}
extern "C" int sc_host_unregister(void* h_ptr) {
    SC_REQUIRE(h_ptr != nullptr, "NULL argument");
    SC_CHECK_HIP(hipHostUnregister(h_ptr));
    return SC_OK;
}

extern "C" int sc_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream) {
    if (bytes == 0) return SC_OK;
    SC_REQUIRE(d_dst != nullptr && h_src != nullptr, "NULL argument");
    SC_CHECK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return SC_OK;
}

extern "C" int sc_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream) {
    if (bytes == 0) return SC_OK;
    SC_REQUIRE(h_dst != nullptr && d_src != nullptr, "NULL argument");
    SC_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SC_OK;
}

extern "C" int sc_memset_zero(void* d_ptr, size_t bytes, void* stream) {
    if (bytes == 0) return SC_OK;
    SC_REQUIRE(d_ptr != nullptr, "NULL argument");
    SC_CHECK_HIP(hipMemsetAsync(d_ptr, 0, bytes, (hipStream_t)stream));
    return SC_OK;
}

extern "C" int sc_stream_create(void** stream) {
    SC_REQUIRE(stream != nullptr, "NULL argument");
    hipStream_t s = nullptr;
    SC_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return SC_OK;
}
extern "C" int sc_stream_destroy(void* stream) {
    if (!stream) return SC_OK;
    SC_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return SC_OK;
}
extern "C" int sc_stream_synchronize(void* stream) {
    SC_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SC_OK;
}

// ---- device-side finite check -----------------------------------------------------------------------------------
// The reference scans the time series on the host in its constructor (transforms.py:746-753: a warning when any sample
// is NaN or infinite).  On the device the same scan is one read of the uploaded series at HBM rate: every thread ORs
// the exponent-all-ones test of its samples, one atomicOr per workgroup that saw one.
template <typename T>
__global__ void __launch_bounds__(256) nonfinite_kernel(const T* __restrict__ x, int64_t n, int32_t* flag) {
    int bad = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const T v = x[i];
        bad |= !(fabs((double)v) <= 1.7976931348623157e308);      // false for NaN and for +-inf
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
__global__ void __launch_bounds__(256) nonfinite4_kernel(const float4* __restrict__ x, int64_t n4, int32_t* flag) {
    unsigned bad = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = x[i];
        // exponent field all ones <=> NaN or infinity
        bad |= ((__float_as_uint(v.x) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(v.y) & 0x7f800000u) == 0x7f800000u) |
               ((__float_as_uint(v.z) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(v.w) & 0x7f800000u) == 0x7f800000u);
    }
    if (__any((int)bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// *d_flag |= 1 when any of the n samples is NaN or +-inf (the caller zeroes the flag; reads it after the stream drains)
extern "C" int sc_nonfinite_f32(const float* d_x, int64_t n, int32_t* d_flag, void* stream) {
    SC_REQUIRE(d_flag != nullptr && (d_x != nullptr || n == 0) && n >= 0, "NULL argument");
    if (n == 0) return SC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = (((uintptr_t)d_x) % 16 == 0) ? n / 4 : 0;
    if (n4) {
        const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
        hipLaunchKernelGGL(nonfinite4_kernel, dim3(grid), dim3(256), 0, s, (const float4*)d_x, n4, d_flag);
    }
    if (n - 4 * n4) hipLaunchKernelGGL(nonfinite_kernel<float>, dim3(1), dim3(256), 0, s, d_x + 4 * n4, n - 4 * n4, d_flag);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
extern "C" int sc_nonfinite_f64(const double* d_x, int64_t n, int32_t* d_flag, void* stream) {
    SC_REQUIRE(d_flag != nullptr && (d_x != nullptr || n == 0) && n >= 0, "NULL argument");
    if (n == 0) return SC_OK;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(nonfinite_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_x, n, d_flag);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
