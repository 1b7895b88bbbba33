// Micro-benchmark (GPU box, round 4): what the waves of ONE SIMD get done side by side in a fixed window of shader cycles.
// Every wave loops over its own body until `window` cycles (s_memtime) have passed and reports how many bodies it
// finished; 256 workgroups (one per CU), wave w sits on SIMD w % 4 (checked with HW_REG_HW_ID is not needed: the roles are
// dealt so that every SIMD gets the same mix whatever the wave -> SIMD order is, as long as it is cyclic).
//   roles: M16 = back-to-back v_mfma_f32_16x16x32_bf16 (8 accumulators), M32 = v_mfma_f32_32x32x16_bf16 (2 accumulators)
//          A = 16 x v_add_f32 acc, acc, |d| on 16 registers, P = 16 x v_perm_b32
//          X = the |Im s| role's step: one v_mfma_f32_32x32x16_bf16 with C = 0 into d, then 16 x v_add_f32 acc, acc, |d_prev|
//              (software pipelined over two d register sets)
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_window.cpp -o tools/issue_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

enum { R_IDLE = 0, R_M16, R_M32, R_A, R_P, R_X };

__device__ __forceinline__ unsigned long long memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int role>
__device__ __forceinline__ void body(unsigned long long window, float* out, unsigned long long* res) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bf16x8 a = {(short)(0x3f80 + lane), 1, 2, 3, 4, 5, 6, 7}, b = {(short)0x3f80, 3, 2, 1, 9, 8, 7, 6};
    unsigned long long n = 0;
    float sink = 0.f;
    asm volatile("s_barrier");
    const unsigned long long t0 = memtime();
    unsigned long long t1 = t0;
    if constexpr (role == R_M16) {
        f32x4 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
        do {
            for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
            n += 32 * 16;
            t1 = memtime();
        } while (t1 - t0 < window);
#pragma unroll
        for (int i = 0; i < 8; ++i) sink += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else if constexpr (role == R_M32) {
        f32x16 c[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        do {
            for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
            n += 16 * 16;
            t1 = memtime();
        } while (t1 - t0 < window);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) sink += c[i][e];
    } else if constexpr (role == R_A || role == R_P) {
        float acc[16], d[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[i] = 0.f; d[i] = (float)(lane - 31 + i) * 1e-3f; }
        do {
            for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (role == R_A) {
#define X(i) asm volatile("v_add_f32 %0, %0, |%1|" : "+v"(acc[i]) : "v"(d[i]));
                    REP16(X)
#undef X
                } else {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(d[i]), "v"(0x07060302u));
                    REP16(X)
#undef X
                }
            }
            n += 64 * 16;
            t1 = memtime();
        } while (t1 - t0 < window);
#pragma unroll
        for (int i = 0; i < 16; ++i) sink += acc[i];
    } else if constexpr (role == R_X) {
        f32x16 acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f32x16 dprev = zero;
        do {
            for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, zero, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s][e] += __builtin_fabsf(dprev[e]);
                __builtin_amdgcn_sched_barrier(0);
                dprev = d;
                asm volatile("v_xor_b32 %0, 1, %0" : "+v"(a[1]));     // operands change (and nothing can be merged)
            }
            n += 4 * 16;
            t1 = memtime();
        } while (t1 - t0 < window);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 16; ++e) sink += acc[s][e];
    }
    if (sink == 1234.5f) out[lane] = sink;
    if (lane == 0) { res[(blockIdx.x * 16 + wave) * 2] = n; res[(blockIdx.x * 16 + wave) * 2 + 1] = t1 - t0; }
}

template <int R0, int R1, int R2, int R3>
__global__ void __launch_bounds__(1024) k(unsigned long long window, float* out, unsigned long long* res) {
    const int slot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    if (slot == 0) body<R0>(window, out, res);
    else if (slot == 1) body<R1>(window, out, res);
    else if (slot == 2) body<R2>(window, out, res);
    else body<R3>(window, out, res);
}

static const char* role_name(int r) {
    switch (r) { case R_M16: return "M16"; case R_M32: return "M32"; case R_A: return "A"; case R_P: return "P"; case R_X: return "X"; default: return "-"; }
}

// per_simd: the roles of the waves of one SIMD (up to 4); every SIMD gets the same
template <int R0, int R1 = R_IDLE, int R2 = R_IDLE, int R3 = R_IDLE>
static void run() {
    std::vector<int> per_simd = {R0};
    if (R1 != R_IDLE) per_simd.push_back(R1);
    if (R2 != R_IDLE) per_simd.push_back(R2);
    if (R3 != R_IDLE) per_simd.push_back(R3);
    const int nw = (int)per_simd.size();
    float* out; unsigned long long* res;
    hipMalloc(&out, 4096); hipMalloc(&res, 256 * 16 * 16);
    const unsigned long long window = 2000000ull;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<R0, R1, R2, R3>), dim3(256), dim3(64 * 4 * nw), 0, 0, window, out, res);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    static unsigned long long h[256 * 16 * 2];
    hipMemcpy(h, res, sizeof h, hipMemcpyDeviceToHost);
    std::string mix;
    for (int j = 0; j < nw; ++j) { mix += role_name(per_simd[j]); mix += ' '; }
    printf("%-18s %.3f ms clock %.2f GHz |", mix.c_str(), ms, (double)h[1] / (ms * 1e6));
    // workgroup 7's waves, SIMD-major: rate of each wave in units per 1000 cycles
    for (int j = 0; j < nw; ++j) {
        double n = 0, t = 0;
        for (int s = 0; s < 4; ++s) { n += (double)h[((7 * 16) + 4 * j + s) * 2]; t += (double)h[((7 * 16) + 4 * j + s) * 2 + 1]; }
        const int r = per_simd[j];
        const double per = t / n;      // cycles per unit (unit = MFMA, VALU instruction or X step)
        printf("  %s: %.2f cyc/%s", role_name(r), per, r == R_X ? "step(1 mfma32+16 add)" : (r == R_A || r == R_P) ? "instr" : "mfma");
    }
    printf("\n");
    hipFree(out); hipFree(res);
}

int main() {
    printf("# each entry: cycles per unit of that wave while all listed waves share one SIMD (2e6-cycle window)\n");
    run<R_A>(); run<R_A, R_A>(); run<R_A, R_A, R_A>();
    run<R_P>(); run<R_P, R_P>();
    run<R_M16>(); run<R_M32>(); run<R_M16, R_M16>(); run<R_M16, R_M32>();
    run<R_M16, R_A>(); run<R_M16, R_A, R_A>(); run<R_M16, R_A, R_A, R_A>();
    run<R_M32, R_A>(); run<R_M32, R_A, R_A>();
    run<R_M16, R_P>(); run<R_M16, R_P, R_P>();
    run<R_X>(); run<R_X, R_X>(); run<R_X, R_X, R_X>();
    run<R_M16, R_X>(); run<R_M16, R_X, R_X>(); run<R_M16, R_X, R_X, R_X>();
    run<R_X, R_A>(); run<R_X, R_A, R_A>();
    return 0;
}
