// Micro-benchmark (GPU box): HBM write-stream ceiling for the store pattern of the fused FFT kernel
// (float4 stores; contiguous vs 256-byte segments scattered at a large stride).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) wr_contig(float4* out, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float4 v = make_float4(i, 1, 2, 3);
    for (; i < n4; i += stride) out[i] = v;
}
// 16 lanes write one 256-byte segment; consecutive segments of a wave are `seg_stride` float4 apart
__global__ void __launch_bounds__(256) wr_seg(float4* out, size_t n_seg, size_t seg_stride, size_t n_groups) {
    size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = n_seg * 16, step = (size_t)gridDim.x * 256;
    float4 v = make_float4(t, 1, 2, 3);
    for (; t < total; t += step) {
        const size_t seg = t >> 4, lane = t & 15;
        // segment s lives at (s % n_groups) * 16 + (s / n_groups) * seg_stride  (like X[f][w,r,k,c-tile])
        const size_t g = seg % n_groups, f = seg / n_groups;
        out[f * seg_stride + g * 16 + lane] = v;
    }
}
__global__ void __launch_bounds__(256) rd_contig(const float4* in, float* sink, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float a = 0;
    for (; i < n4; i += stride) { float4 v = in[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 12345.678f) sink[0] = a;
}
int main() {
    const size_t bytes = (size_t)6 << 30, n4 = bytes / 16;
    float4* buf; float* sink; hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](auto launch, const char* name) {
        for (int r = 0; r < 2; ++r) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); }
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
    };
    time([&] { hipLaunchKernelGGL(wr_contig, dim3(256 * 8), dim3(256), 0, 0, buf, n4); }, "write contiguous float4");
    time([&] { hipLaunchKernelGGL(rd_contig, dim3(256 * 8), dim3(256), 0, 0, buf, sink, n4); }, "read contiguous float4");
    // spectra layout: F=129 rows of 50176 segments (W*R*K*C/32 = 7*1000*7*4 = 196000 -> use 6GB/129/256B)
    const size_t F = 129, n_groups = n4 / 16 / F, seg_stride = n_groups * 16;
    time([&] { hipLaunchKernelGGL(wr_seg, dim3(256 * 8), dim3(256), 0, 0, buf, n_groups * F, seg_stride, n_groups); },
         "write 256B segments, consecutive");
    // each wave handles one group g and walks f (stride seg_stride): the FFT kernel's actual pattern
    time([&] { hipLaunchKernelGGL(wr_seg, dim3(256 * 8), dim3(256), 0, 0, buf, n_groups * F, 16, F); },
         "write 256B segments, f-major walk");
    return 0;
}
