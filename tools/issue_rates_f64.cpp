// Micro-benchmark (GPU box, round 5): what one SIMD of gfx950 issues per cycle in DOUBLE precision -- the arithmetic of the
// float64 engine's stage B (sc_f64.hip: cross-spectra on v_mfma_f64_16x16x4_f64, the per-observation |Im s| plane as
// v_mul_f64 / v_fma_f64 / v_add_f64 |d|):
//   * v_fma_f64, v_mul_f64, v_add_f64 acc, acc, |d| for 1..3 waves per SIMD;
//   * v_mfma_f64_16x16x4_f64 back to back (one wave per SIMD);
//   * both together on one SIMD: do the matrix instruction and the vector instructions share the fp64 units?
// Same harness as tools/issue_rates.cpp (inline asm on fixed registers, cycles from s_memtime, wall time from hipEvents).
// Build: hipcc -O3 --offload-arch=gfx950 tools/issue_rates_f64.cpp -o tools/issue_rates_f64
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double f64x4 __attribute__((ext_vector_type(4)));
enum { M_FMA = 0, M_MUL, M_ADDABS, M_NONE };

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__device__ __forceinline__ void valu_wave(int iters, double* out, int lane) {
    double acc[16], d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.0; d[i] = (double)(lane - 31 + i) * 1e-3; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == M_FMA) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(d[i]));
                REP16(X)
#undef X
            } else if (MODE == M_MUL) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(acc[i]) : "v"(d[i]));
                REP16(X)
#undef X
            } else if (MODE == M_ADDABS) {
#define X(i) asm volatile("v_add_f64 %0, %0, |%1|" : "+v"(acc[i]) : "v"(d[i]));
                REP16(X)
#undef X
            }
        }
    }
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i];
    if (t == 1234.5) out[lane] = t;
}

__device__ __forceinline__ void mfma_wave(int n, double* out, int lane) {
    f64x4 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = (f64x4){0, 0, 0, 0};
    const double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    for (int it = 0; it < n / 8; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (t == 1234.5) out[lane] = t;
}

__device__ __forceinline__ unsigned long long memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

// waves 0..3 (one per SIMD): MFMA stream when n_mfma > 0; waves 4..: VALU waves
template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, int n_mfma, double* out, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    const unsigned long long t0 = memtime();
    if (wave < 4) { if (n_mfma > 0) mfma_wave(n_mfma, out, lane); }
    else if (MODE != M_NONE) valu_wave<MODE>(iters, out, lane);
    const unsigned long long t1 = memtime();
    if (lane == 0) cyc[blockIdx.x * 32 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int valu_per_simd, int mfma_per_simd) {
    const int iters = 3000;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * 32 * 8);
    hipMemset(cyc, 0, 256 * 32 * 8);
    const int waves = 4 + 4 * valu_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(64 * waves), 0, 0, iters, mfma_per_simd, out, cyc);
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
    printf("%-34s valu waves/SIMD %d  mfma/SIMD %6d : %7.3f ms  clock %.2f GHz", name, valu_per_simd, mfma_per_simd, ms, mx / (ms * 1e6));
    if (valu_per_simd) printf("  %.2f cyc per VALU instr per SIMD", mx_v / instr);
    if (mfma_per_simd) printf("  %.1f cyc per MFMA", (double)mx_m / mfma_per_simd);
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    printf("# tools/issue_rates_f64.cpp: fp64 issue rates of one gfx950 SIMD (wave64: 64 lanes per instruction)\n");
    for (int w = 1; w <= 3; ++w) run<M_FMA>("v_fma_f64", w, 0);
    for (int w = 1; w <= 3; ++w) run<M_MUL>("v_mul_f64", w, 0);
    for (int w = 1; w <= 3; ++w) run<M_ADDABS>("v_add_f64 |d|", w, 0);
    run<M_NONE>("v_mfma_f64_16x16x4_f64", 0, 40000);
    // the vector instructions next to a matrix-core wave on the same SIMD (3000 * 64 * w VALU instr vs n MFMAs)
    for (int w = 1; w <= 2; ++w) run<M_FMA>("v_fma_f64 + v_mfma_f64_16x16x4", w, 12000);
    for (int w = 1; w <= 2; ++w) run<M_ADDABS>("v_add_f64 |d| + v_mfma_f64_16x16x4", w, 12000);
    return 0;
}
