// Micro-benchmark (GPU box, round 4): what one SIMD of gfx950 issues per cycle.
//   * v_add_f32 acc, acc, |d|  (the |Im s| accumulate), v_pk_add_f32, v_perm_b32, v_pk_fma_f32 for 1..4 waves per SIMD;
//   * the same next to a wave that issues back-to-back bf16 MFMAs on the same SIMD;
//   * v_mfma_f32_32x32x16_bf16 vs the legacy v_mfma_f32_32x32x8_bf16_1k and v_mfma_f32_16x16x32_bf16 vs 16x16x16_bf16_1k.
// Every body is inline asm on fixed registers (nothing for the compiler to merge); cycles from s_memtime
// (shader clock), wall time from hipEvents, so the sustained clock = cycles / wall is reported too.
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_rates.cpp -o tools/issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

enum { M_ADDABS = 0, M_PKADD, M_PERM, M_PKFMA, M_FMA, M_ADDABS_DEP16, M_NONE };

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// one VALU wave: `iters` x 64 instructions on 16 (or 32) independent accumulators
template <int MODE>
__device__ __forceinline__ void valu_wave(int iters, float* out, int lane) {
    float acc[16], d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; d[i] = (float)(lane - 31 + i) * 1e-3f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == M_ADDABS) {
#define X(i) asm volatile("v_add_f32 %0, %0, |%1|" : "+v"(acc[i]) : "v"(d[i]));
                REP16(X)
#undef X
            } else if (MODE == M_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(d[i]));
                REP16(X)
#undef X
            } else if (MODE == M_PERM) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(d[i]), "v"(0x07060302u));
                REP16(X)
#undef X
            } else if (MODE == M_PKADD || MODE == M_PKFMA) {
                // 8 packed instructions on register pairs = 16 values; issued twice to keep 16 instructions per r
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#define XP(i)                                                                                                  \
    {                                                                                                          \
        typedef float f2 __attribute__((ext_vector_type(2)));                                                  \
        f2 a = {acc[2 * i], acc[2 * i + 1]};                                                                   \
        const f2 v = {d[2 * i], d[2 * i + 1]};                                                                 \
        if (MODE == M_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(v));                      \
        else asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(a) : "v"(v));                                  \
        acc[2 * i] = a[0]; acc[2 * i + 1] = a[1];                                                              \
    }
                    XP(0) XP(1) XP(2) XP(3) XP(4) XP(5) XP(6) XP(7)
#undef XP
                }
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i];
    if (t == 1234.5f) out[lane] = t;
}

template <int KIND>   // 0: 32x32x16, 1: 32x32x8_1k, 2: 16x16x32, 3: 16x16x16_1k
__device__ __forceinline__ void mfma_wave(int n, float* out, int lane) {
    bf16x8 a = {(short)(0x3f80 + lane), 1, 2, 3, 4, 5, 6, 7}, b = {(short)0x3f80, 3, 2, 1, 9, 8, 7, 6};
    bf16x4 a4 = {(short)(0x3f80 + lane), 1, 2, 3}, b4 = {(short)0x3f80, 3, 2, 1};
    float t = 0.f;
    if (KIND <= 1) {
        f32x16 c[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int it = 0; it < n / 2; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
                else c[i] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, c[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += c[i][e];
    } else {
        f32x4 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < n / 8; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 2) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
                else c[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) t += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    }
    if (t == 1234.5f) out[lane] = t;
}

__device__ __forceinline__ unsigned long long memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

// waves 0..3 (one per SIMD): MFMA stream of kind MK when n_mfma > 0; waves 4..: VALU waves
template <int MODE, int MK>
__global__ void __launch_bounds__(1024) k(int iters, int n_mfma, float* out, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    const unsigned long long t0 = memtime();
    if (wave < 4) { if (n_mfma > 0) mfma_wave<MK>(n_mfma, out, lane); }
    else if (MODE != M_NONE) valu_wave<MODE>(iters, out, lane);
    const unsigned long long t1 = memtime();
    if (lane == 0) cyc[blockIdx.x * 32 + wave] = t1 - t0;
}

template <int MODE, int MK>
static void run(const char* name, int valu_per_simd, int mfma_per_simd) {
    const int iters = 6000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * 32 * 8);
    hipMemset(cyc, 0, 256 * 32 * 8);
    const int waves = 4 + 4 * valu_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE, MK>), dim3(256), dim3(64 * waves), 0, 0, iters, mfma_per_simd, out, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    static unsigned long long h[256 * 32];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long mx_v = 0, mx_m = 0;
    for (int blk = 0; blk < 256; ++blk)
        for (int i = 0; i < waves; ++i) {
            const unsigned long long c = h[blk * 32 + i];
            if (i < 4) { if (c > mx_m) mx_m = c; } else if (c > mx_v) mx_v = c;
        }
    const double instr = (double)iters * 64 * valu_per_simd;     // VALU wave-instructions per SIMD
    const unsigned long long mx = mx_v > mx_m ? mx_v : mx_m;
    printf("%-30s valu waves/SIMD %d  mfma/SIMD %6d : %7.3f ms  clock %.2f GHz", name, valu_per_simd, mfma_per_simd, ms,
           mx / (ms * 1e6));
    if (valu_per_simd) printf("  %.2f cyc per VALU instr per SIMD (valu waves %llu cyc)", mx_v / instr, mx_v);
    if (mfma_per_simd) printf("  %.1f cyc per MFMA (mfma wave %llu cyc)", (double)mx_m / mfma_per_simd, mx_m);
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    printf("# s_memtime ticks taken as shader cycles only if 'clock' reads a plausible GHz figure; otherwise it is the 100 MHz counter\n");
    for (int w = 1; w <= 3; ++w) run<M_ADDABS, 0>("v_add_f32 |d|", w, 0);
    for (int w = 1; w <= 3; ++w) run<M_FMA, 0>("v_fma_f32", w, 0);
    for (int w = 1; w <= 3; ++w) run<M_PKADD, 0>("v_pk_add_f32 (2 values)", w, 0);
    for (int w = 1; w <= 3; ++w) run<M_PKFMA, 0>("v_pk_fma_f32 (2 values)", w, 0);
    for (int w = 1; w <= 3; ++w) run<M_PERM, 0>("v_perm_b32", w, 0);
    run<M_NONE, 0>("mfma 32x32x16 bf16", 0, 40000);
    run<M_NONE, 1>("mfma 32x32x8 bf16_1k", 0, 40000);
    run<M_NONE, 2>("mfma 16x16x32 bf16", 0, 80000);
    run<M_NONE, 3>("mfma 16x16x16 bf16_1k", 0, 80000);
    // VALU next to a matrix-core wave on the same SIMD (6000 * 64 * w VALU instr vs n MFMAs)
    for (int w = 1; w <= 2; ++w) run<M_ADDABS, 0>("v_add |d| + mfma 32x32x16", w, 30000);
    for (int w = 1; w <= 2; ++w) run<M_ADDABS, 2>("v_add |d| + mfma 16x16x32", w, 60000);
    for (int w = 1; w <= 2; ++w) run<M_PKADD, 2>("v_pk_add + mfma 16x16x32", w, 60000);
    return 0;
}
