// Probe (GPU box): semantics of global_load_lds_dwordx4 on gfx950 -- lane l of the wave lands at LDS base + 16 * l.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float4* src, float4* out, const int* perm) {
    __shared__ float4 buf[2][64];
    const int lane = threadIdx.x;
    // lane fetches element perm[lane]; lands in buf[1][lane]
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + perm[lane]),
                                     (__attribute__((address_space(3))) void*)&buf[1][0], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = buf[1][lane];
}
int main() {
    float4 h[64]; int p[64];
    for (int i = 0; i < 64; ++i) { h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f); p[i] = i ^ 5; }
    float4 *d, *o; int* dp;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof h); hipMalloc(&dp, sizeof p);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice); hipMemcpy(dp, p, sizeof p, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, dp);
    float4 r[64]; hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) if (r[i].x != (float)(i ^ 5) || r[i].w != (i ^ 5) + 0.75f) ++bad;
    printf("direct-to-LDS b128: %s (r[0]=%g r[1]=%g r[63]=%g)\n", bad ? "MISMATCH" : "ok", r[0].x, r[1].x, r[63].x);
    return bad != 0;
}
