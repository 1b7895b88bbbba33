// GPU box, round 4: what ds_read_b64_tr_b16 returns and what its LDS conflicts cost.
//  (1) semantics: lane l reads 8 bytes at LDS byte offset 64 l (elements 32 l .. 32 l + 3 of a u16 ramp); print for every
//      lane which (source lane, element) each of its four result halves came from.
//  (2) timing: cycles per wave-instruction for address patterns of the stage-B layouts (rows 256 B apart with and
//      without the rotation swizzle, broadcast rows), 4 waves per SIMD issuing 64 loads per s_waitcnt.
// Build: hipcc -O3 --offload-arch=gfx950 tools/tr_load_test.cpp -o tools/tr_load_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void sem_kernel(unsigned short* out) {
    __shared__ unsigned short lds[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + 32 * lane));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}

__device__ __forceinline__ unsigned long long memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

// pattern: byte offset of lane l's 8 bytes; 16 back-to-back loads at offsets +0 (same addresses: pure issue/conflict cost)
template <int KIND>
__global__ void __launch_bounds__(1024) time_kernel(const int* offs, unsigned long long* cyc, int iters, int* sink) {
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned a = (unsigned)offs[lane];
    int acc = 0;
    const unsigned long long t0 = memtime();
    for (int it = 0; it < iters; ++it) {
        s16x4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (KIND == 0) v[q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(smem + a + q * 4608));
            else {
                const uint2 u = *reinterpret_cast<const uint2*>(smem + a + q * 4608);
                v[q] = __builtin_bit_cast(s16x4, u);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q][0] ^ v[q][3];
    }
    const unsigned long long t1 = memtime();
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc == 0x7fffffff) sink[0] = acc;
}

static void run_time(const char* name, const int* offs_host) {
    int* offs; unsigned long long* cyc; int* sink;
    hipMalloc(&offs, 256); hipMalloc(&cyc, 256 * 16 * 8); hipMalloc(&sink, 16);
    hipMemcpy(offs, offs_host, 256, hipMemcpyHostToDevice);
    for (int kind = 0; kind < 2; ++kind) {
        const int iters = 2000, waves = 16;
        if (kind == 0) { hipFuncSetAttribute((const void*)time_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            hipLaunchKernelGGL(time_kernel<0>, dim3(256), dim3(64 * waves), 163840, 0, offs, cyc, iters, sink); }
        else { hipFuncSetAttribute((const void*)time_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            hipLaunchKernelGGL(time_kernel<1>, dim3(256), dim3(64 * waves), 163840, 0, offs, cyc, iters, sink); }
        hipDeviceSynchronize();
        unsigned long long h[16]; hipMemcpy(h, cyc + 16 * 5, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long mx = 0; for (int i = 0; i < waves; ++i) if (h[i] > mx) mx = h[i];
        printf("%-44s %s: %.2f cycles per wave-instruction per CU (16 waves)\n", name, kind == 0 ? "ds_read_b64_tr_b16" : "ds_read_b64        ",
               (double)mx / ((double)iters * 16 * waves));
    }
    hipFree(offs); hipFree(cyc); hipFree(sink);
}

int main() {
    unsigned short* out; hipMalloc(&out, 64 * 4 * 2);
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, out);
    unsigned short h[256]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("# ds_read_b64_tr_b16: result half e of lane l <- (source lane, source element)\n");
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf("  (%2d,%d)", h[l * 4 + e] / 32, h[l * 4 + e] % 32);
        printf("\n");
    }
    int offs[64];
    // (a) contiguous: lane l at 8 l (conflict-free by construction for plain b64)
    for (int l = 0; l < 64; ++l) offs[l] = 8 * l;
    run_time("contiguous 8 l", offs);
    // (b) stage-B rows 256 B apart, no swizzle: 16-lane group g, row r = (l >> 2) & 3, chunk q = l & 3:  row (4 g' ...)
    //     lanes 0-15: rows 0-3 at column span 0; lanes 16-31: rows 0-3 at span 1 (abs role) ; lanes 32-63 the same + 768
    for (int l = 0; l < 64; ++l) offs[l] = ((l >> 2) & 3) * 256 + ((l >> 4) & 1) * 32 + (l & 3) * 8 + (l >> 5) * 768;
    run_time("rows 256 B apart, no swizzle (abs role)", offs);
    // (c) the same with the rotation swizzle: span' = (span + 2 row) mod 8
    for (int l = 0; l < 64; ++l) {
        const int row = (l >> 2) & 3, span = ((l >> 4) & 1);
        offs[l] = row * 256 + ((span + 2 * row) & 7) * 32 + (l & 3) * 8 + (l >> 5) * 768;
    }
    run_time("rows 256 B apart, rotated spans (abs role)", offs);
    // (d) three of four rows identical (planes h h m h): broadcast
    for (int l = 0; l < 64; ++l) {
        const int r = (l >> 2) & 3, row = (r == 2) ? 1 : 0, span = ((l >> 4) & 1);
        offs[l] = row * 256 + ((span + 2 * row) & 7) * 32 + (l & 3) * 8 + (l >> 5) * 768;
    }
    run_time("rows h h m h, rotated spans (abs role)", offs);
    // (e) CSM role: lanes 0-15 obs 0-3, lanes 16-31 obs 4-7, lanes 32-47 obs 8-11, 48-63 obs 12-15; same span; row stride 1536
    for (int l = 0; l < 64; ++l) {
        const int o = ((l >> 4) & 3) * 4 + ((l >> 2) & 3);
        offs[l] = o * 1536 + (l & 3) * 8;
    }
    run_time("CSM rows 1536 B apart, no swizzle", offs);
    for (int l = 0; l < 64; ++l) {
        const int o = ((l >> 4) & 3) * 4 + ((l >> 2) & 3);
        offs[l] = o * 1536 + ((o & 7) * 32) + (l & 3) * 8;
    }
    run_time("CSM rows 1536 B apart, rotated spans", offs);
    return 0;
}
