// Micro-benchmark (GPU box): VALU issue rate per SIMD vs number of resident waves.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, float* out) {
    float acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float x = (float)tid * 1e-3f, y = 1.0f - x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            if (MODE == 0) acc[q] = fmaf(x, y, acc[q]);
            else { float t = x * acc[(q + 1) & 31]; t = fmaf(-y, x, t); acc[q] += fabsf(t); }
        }
        x += 1e-7f;
    }
    float s = 0;
    for (int q = 0; q < 32; ++q) s += acc[q];
    out[tid] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode)
        for (int waves = 4; waves <= 16; waves += 4) {
            auto kern = mode ? k<1> : k<0>;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, iters, out);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            double instr = (double)iters * 32 * (mode ? 3 : 1);   // per wave
            double cyc_per_instr_per_simd = ms * 1e-3 * 2.4e9 / (instr * waves / 4);
            printf("mode %d waves/CU %2d (%d per SIMD): %.3f ms  %.2f cycles per VALU instr per SIMD (at 2.4 GHz)\n",
                   mode, waves, waves / 4, ms, cyc_per_instr_per_simd);
        }
    return 0;
}
