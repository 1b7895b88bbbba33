// Micro-benchmark (GPU box): do f32 MFMA and f32 VALU FMA overlap when they come from two
// different waves on the same SIMD?  (design question for the fused stage-B kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { auto e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void mfma_loop(int iters, float* out, int tid) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    float x = (float)tid * 1e-3f, y = 1.0f - x;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a6, 0, 0, 0);
        a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a7, 0, 0, 0);
    }
    f32x4 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[tid] = s[0] + s[1] + s[2] + s[3];
}
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma_bf16_loop(int iters, float* out, int tid) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    bf16x8 x, y;
    for (int k = 0; k < 8; ++k) { x[k] = (short)(0x3f80 + ((tid + k) & 7)); y[k] = (short)(0x3f00 + ((tid * 3 + k) & 7)); }
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a6, 0, 0, 0);
        a7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, a7, 0, 0, 0);
    }
    f32x4 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[tid] = s[0] + s[1] + s[2] + s[3];
}
typedef short bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_bf16k16_loop(int iters, float* out, int tid) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    bf16x4 x, y;
    for (int k = 0; k < 4; ++k) { x[k] = (short)(0x3f80 + ((tid + k) & 7)); y[k] = (short)(0x3f00 + ((tid * 3 + k) & 7)); }
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y, y, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y, x, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, x, a6, 0, 0, 0);
        a7 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y, y, a7, 0, 0, 0);
    }
    f32x4 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[tid] = s[0] + s[1] + s[2] + s[3];
}
template <int MODE>  // 0: v_fma  1: mul+fma+add|.| (the |Im| pattern)
__device__ __forceinline__ void valu_loop(int iters, float* out, int tid) {
    float acc[32];
    for (int k = 0; k < 32; ++k) acc[k] = 0.f;
    float x = (float)tid * 1e-3f, y = 1.0f - x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (MODE == 0) acc[k] = fmaf(x, y, acc[k]);
            else { float t = x * acc[(k + 1) & 31]; t = fmaf(-y, x, t); acc[k] += fabsf(t); }
        }
        x += 1e-7f;
    }
    float s = 0;
    for (int k = 0; k < 32; ++k) s += acc[k];
    out[tid] = s;
}
// which: 1 = mfma waves only, 2 = valu waves only, 3 = both roles (512 threads)
template <int MODE>
__global__ void __launch_bounds__(512) k(int which, int it_m, int it_v, float* out) {
    extern __shared__ float pad[];
    int wave = threadIdx.x >> 6;
    int tid = blockIdx.x * 512 + threadIdx.x;
    if (wave < 4) { if (which & 1) mfma_loop(it_m, out, tid); if (which & 4) mfma_bf16_loop(it_m, out, tid); if (which & 8) mfma_bf16k16_loop(it_m, out, tid); }
    else { if (which & 2) valu_loop<MODE>(it_v, out, tid); }
}
int main() {
    float* out; CK(hipMalloc(&out, 256 * 512 * 4 * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int it_m = 20000, it_v = 20000;
    auto run = [&](auto kern, int which, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 100 * 1024, 0, which, it_m, it_v, out);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        double mf = 256.0 * 4 * it_m * 8 * 2.0 * 16 * 16 * ((which & 4) ? 32 : (which & 8) ? 16 : 4), vf = 256.0 * 4 * 64 * it_v * 32.0;
        printf("%-28s %8.3f ms   mfma %.1f TF   valu %.1f Ginstr-lane/s\n", name, ms,
               (which & 13) ? mf / ms / 1e9 : 0.0, (which & 2) ? vf / ms / 1e6 : 0.0);
    };
    run(k<0>, 1, "mfma waves only");
    run(k<0>, 2, "valu(fma) waves only");
    run(k<0>, 3, "both roles (fma)");
    run(k<1>, 2, "valu(mul,fma,add|.|) only");
    run(k<1>, 3, "both roles (mul,fma,add)");
    run(k<1>, 8, "bf16 16x16x16 mfma only");
    run(k<1>, 4, "bf16 mfma waves only");
    run(k<1>, 6, "bf16 mfma + valu(mul,fma,add)");
    run(k<0>, 6, "bf16 mfma + valu(fma)");
    return 0;
}
