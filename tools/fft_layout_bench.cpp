// Micro-benchmark (GPU box): rocFFT real-forward throughput for the batch layouts the design
// can choose between.  Build: hipcc -O2 --offload-arch=gfx950 tools/fft_layout_bench.cpp -lrocfft
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("FAIL %s -> %d line %d\n", #x, (int)e_, __LINE__); exit(1); } } while (0)

double run(size_t N, size_t batch, bool in_strided, bool out_strided) {
    size_t F = N / 2 + 1;
    float* in; float2* out;
    CK(hipMalloc(&in, sizeof(float) * N * batch));
    CK(hipMalloc(&out, sizeof(float2) * F * batch));
    CK(hipMemset(in, 0, sizeof(float) * N * batch));
    rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
    size_t is[1] = {in_strided ? batch : 1}, os[1] = {out_strided ? batch : 1};
    size_t idist = in_strided ? 1 : N, odist = out_strided ? 1 : F;
    CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                               nullptr, nullptr, 1, is, idist, 1, os, odist));
    rocfft_plan p; size_t len[1] = {N};
    CK(rocfft_plan_create(&p, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                          rocfft_precision_single, 1, len, batch, d));
    size_t ws = 0; CK(rocfft_plan_get_work_buffer_size(p, &ws));
    void* wb = nullptr; rocfft_execution_info info; CK(rocfft_execution_info_create(&info));
    if (ws) { CK(hipMalloc(&wb, ws)); CK(rocfft_execution_info_set_work_buffer(info, wb, ws)); }
    void* ib[1] = {in}; void* ob[1] = {out};
    for (int i = 0; i < 2; ++i) CK(rocfft_execute(p, ib, ob, info));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int it = 5;
    CK(hipEventRecord(a));
    for (int i = 0; i < it; ++i) CK(rocfft_execute(p, ib, ob, info));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= it;
    double gb = (4.0 * N + 8.0 * F) * batch / 1e9;
    printf("N=%zu batch=%zu in=%s out=%s : %.3f ms  %.1f GB/s algorithmic  work=%.1f MB\n", N, batch,
           in_strided ? "strided" : "unit", out_strided ? "strided" : "unit", ms, gb / (ms * 1e-3), ws / 1e6);
    fflush(stdout);
    rocfft_plan_destroy(p); rocfft_plan_description_destroy(d); rocfft_execution_info_destroy(info);
    if (wb) hipFree(wb); hipFree(in); hipFree(out);
    return ms;
}

int main() {
    CK(rocfft_setup());
    size_t cfgs[][2] = {{256, 896000}, {1024, 16000}, {1024, 256000}, {4096, 64000}, {250, 100000}};
    for (auto& c : cfgs)
        for (int m = 0; m < 4; ++m) run(c[0], c[1], m & 1, m & 2);
    return 0;
}
