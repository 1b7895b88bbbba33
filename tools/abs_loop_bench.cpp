// Micro-benchmark (GPU box): the inner loop of the |Im| role of sc_fused.hip in isolation.
// Each "abs" wave repeats { d = mfma_32x32x16_bf16(a, b, 0); acc[s] += |d_prev| } over NBLK accumulator
// blocks; optionally a "CSM" wave per SIMD issues back-to-back v_mfma_f32_16x16x32_bf16.
// Reports cycles per (block,row) step per SIMD for 1..3 abs waves per SIMD, with and without the CSM wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int NBLK, bool MFMA, bool ADDS>
__device__ __forceinline__ void abs_wave(int iters, float* out, int lane) {
    f32x16 acc[NBLK];
#pragma unroll
    for (int s = 0; s < NBLK; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 a = {(short)(0x3f80 + lane), 1, 2, 3, 4, 5, 6, 7}, b = {(short)0x3f80, 3, 2, 1, 9, 8, 7, 6};
    for (int it = 0; it < iters; ++it) {
        f32x16 dprev = zero;
#pragma unroll
        for (int s = 0; s < NBLK; ++s) {
            f32x16 d = zero;
            if (MFMA) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, zero, 0, 0, 0);
            else asm volatile("" : "+v"(d));
            __builtin_amdgcn_sched_barrier(0);
            if (ADDS && s > 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s - 1][e] += __builtin_fabsf(dprev[e]);
            }
            if (!ADDS) asm volatile("" ::"v"(d));        // keep every MFMA alive
            dprev = d;
            a[1] ^= 1;   // keep the operands live and changing (one VALU op like the fragment permutes)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ADDS) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[NBLK - 1][e] += __builtin_fabsf(dprev[e]);
        } else {
            asm volatile("" ::"v"(dprev));
        }
    }
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < NBLK; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[s][e];
    if (t == 1234.5f) out[lane] = t;
}

__device__ __forceinline__ void csm_wave(int n_mfma, float* out, int lane) {
    f32x4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
    bf16x8 a = {(short)(0x3f80 + lane), 1, 2, 3, 4, 5, 6, 7}, b = {(short)0x3f80, 3, 2, 1, 9, 8, 7, 6};
    for (int it = 0; it < n_mfma / 8; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    }
    float t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (t == 1234.5f) out[lane] = t;
}

// waves 0..3: CSM (if csm_mfma > 0), waves 4..: abs
template <int NBLK, bool MFMA, bool ADDS>
__global__ void __launch_bounds__(1024) k(int iters, int csm_mfma, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    const long long t0 = clock64();
    if (wave < 4) { if (csm_mfma > 0) csm_wave(csm_mfma, out, lane); }
    else abs_wave<NBLK, MFMA, ADDS>(iters, out, lane);
    const long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int NBLK, bool MFMA, bool ADDS>
static void run(const char* name, int abs_per_simd, int csm_mfma_per_step_x100) {
    const int iters = 4000;
    float* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * 16 * 8);
    hipMemset(cyc, 0, 256 * 16 * 8);
    const int waves = 4 + 4 * abs_per_simd;
    const long long steps = (long long)iters * NBLK * abs_per_simd;       // (block,row) steps per SIMD
    const int csm = (int)(steps * csm_mfma_per_step_x100 / 100);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<NBLK, MFMA, ADDS>), dim3(256), dim3(64 * waves), 0, 0, iters, csm, out, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    long long h[256 * 16]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    long long mx_abs = 0, mx_csm = 0;
    for (int i = 0; i < 16; ++i) { if (i < 4) { if (h[i] > mx_csm) mx_csm = h[i]; } else if (h[i] > mx_abs) mx_abs = h[i]; }
    // clock64 counts at 100 MHz (constant clock): convert to ns
    printf("%-34s abs/SIMD %d  csm/step %.2f : %7.3f ms  -> %6.1f ns per step per SIMD (abs waves %lld, csm waves %lld ticks)\n",
           name, abs_per_simd, csm_mfma_per_step_x100 / 100.0, ms, ms * 1e6 / steps, mx_abs, mx_csm);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 3; ++w) run<5, true, true>("mfma + 16 adds", w, 0);
    for (int w = 1; w <= 3; ++w) run<5, false, true>("16 adds only", w, 0);
    for (int w = 1; w <= 3; ++w) run<5, true, false>("mfma only", w, 0);
    run<5, true, true>("mfma + 16 adds + CSM wave", 2, 540);     // 216 CSM MFMAs per 40 steps
    run<5, false, true>("16 adds only + CSM wave", 2, 540);
    run<5, true, false>("mfma only + CSM wave", 2, 540);
    run<2, true, true>("mfma + 16 adds, 2 blocks", 3, 0);
    return 0;
}
