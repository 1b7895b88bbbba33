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
// The fused FFT kernel's store pattern with the arithmetic removed: workgroup (c-tile, trial, window) walks
// K tapers; per taper 129 frequency rows x 256 B (CT=32 channels) or x 1 KB (CT=128).
template <int CT>
__global__ void __launch_bounds__(256) wr_mtfft(float2* X, int W, int R, int K, int C, int F, int lds_work) {
    __shared__ float2 z[16 * 273];
    const int tid = threadIdx.x, c0 = blockIdx.x * CT, r = blockIdx.y, w = blockIdx.z;
    constexpr int NF = CT / 2;
    const size_t sF = (size_t)W * R * K * C;
    for (int i = tid; i < 16 * 273; i += 256) z[i] = make_float2(i, tid);
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        float2* Xk = X + (((size_t)w * R + r) * K + k) * C + c0;
        for (int idx = tid; idx < F * NF; idx += 256) {
            const int f = idx / NF, pr = idx - f * NF;
            float2 a = make_float2(idx, k), b = make_float2(tid, w);
            if (lds_work) { a = z[(pr & 15) * 273 + f]; b = z[(pr & 15) * 273 + ((256 - f) & 255)]; }
            *reinterpret_cast<float4*>(Xk + (size_t)f * sF + 2 * pr) = make_float4(a.x, a.y, b.x, b.y);
        }
        if (lds_work) __syncthreads();
    }
}
int main() {
    const size_t bytes = (size_t)6 << 30, n4 = bytes / 16;   // timed streams use 6 GiB of a 7 GiB buffer
    float4* buf; float* sink; hipMalloc(&buf, (size_t)7 << 30); hipMalloc(&sink, 4);
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
    {
        const int W = 7, R = 1000, K = 7, C = 128, F = 129;
        const size_t mt_bytes = (size_t)F * W * R * K * C * 8;
        auto time2 = [&](auto launch, const char* name) {
            for (int r = 0; r < 2; ++r) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); }
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, mt_bytes / (ms * 1e-3) / 1e12);
        };
        time2([&] { hipLaunchKernelGGL(wr_mtfft<32>, dim3(4, R, W), dim3(256), 0, 0, (float2*)buf, W, R, K, C, F, 0); },
              "mtfft store pattern, 256 B rows, no LDS");
        time2([&] { hipLaunchKernelGGL(wr_mtfft<32>, dim3(4, R, W), dim3(256), 0, 0, (float2*)buf, W, R, K, C, F, 1); },
              "mtfft store pattern, 256 B rows, LDS reads");
        time2([&] { hipLaunchKernelGGL(wr_mtfft<128>, dim3(1, R, W), dim3(256), 0, 0, (float2*)buf, W, R, K, C, F, 0); },
              "mtfft store pattern, 1 KB rows, no LDS");
        time2([&] { hipLaunchKernelGGL(wr_mtfft<64>, dim3(2, R, W), dim3(256), 0, 0, (float2*)buf, W, R, K, C, F, 0); },
              "mtfft store pattern, 512 B rows, no LDS");
    }
    return 0;
}
