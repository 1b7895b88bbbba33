// sc_mtfft_bfly.h -- the float32 butterflies shared by the fused multitaper transform kernels (sc_mtfft.hip, sc_mtfft_long.hip).
#pragma once
#include <hip/hip_runtime.h>

__device__ inline float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// ----------------------------------------------------------------------------------------
// N = 64 (16x4), 128 (16x8), 256 (16x16), 512 (16x16x2), 1024 (16x16x4), 2048 (16x16x8), 4096 (16x16x16):
// every thread owns 16 points per pass, so a 256-point transform is TWO register-resident radix-16 butterflies with
// one LDS exchange between them (a radix-4 Stockham kernel, the first version of this file, needed four passes,
// eight barriers and a separate pack pass: 1.3-2.1 TB/s stored where this one reaches 3.0-3.8).  Pass 1 reads the detrended window tile directly (x * taper, packed two
// channels per complex sequence), so the tapered sequences are never materialised.  The exchange
// buffer is skewed, phys(idx) = idx + idx/16, which makes the stride-16 writes of pass 1 and the
// stride-N/16 reads of pass 2 both conflict-free; window rows are padded by 2 floats for the same
// reason.  3 + (1 if N > 256) workgroup barriers per taper instead of 10.
__device__ __forceinline__ void dft4r(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 b0 = make_float2(a0.x + a2.x, a0.y + a2.y), b1 = make_float2(a0.x - a2.x, a0.y - a2.y);
    const float2 b2 = make_float2(a1.x + a3.x, a1.y + a3.y), b3 = make_float2(a1.y - a3.y, a3.x - a1.x);
    a0 = make_float2(b0.x + b2.x, b0.y + b2.y);
    a1 = make_float2(b1.x + b3.x, b1.y + b3.y);
    a2 = make_float2(b0.x - b2.x, b0.y - b2.y);
    a3 = make_float2(b1.x - b3.x, b1.y - b3.y);
}
__device__ __forceinline__ float2 cmulc(float2 a, float c, float s) {   // a * (c + i s)
    return make_float2(a.x * c - a.y * s, a.x * s + a.y * c);
}
// in: x[n], n = 4*n1 + n2 ; out: o[k], k = k1 + 4*k2   (forward DFT, exp(-2 pi i nk/16))
__device__ __forceinline__ void dft16(float2 (&x)[16], float2 (&o)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4r(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);   // x[4*k1 + n2]
    // twiddles W16^(n2*k1)
    x[4 + 1] = cmulc(x[4 + 1], C1, -S1);  x[8 + 1] = cmulc(x[8 + 1], H, -H);    x[12 + 1] = cmulc(x[12 + 1], S1, -C1);
    x[4 + 2] = cmulc(x[4 + 2], H, -H);    x[8 + 2] = make_float2(x[8 + 2].y, -x[8 + 2].x);
    x[12 + 2] = cmulc(x[12 + 2], -H, -H);
    x[4 + 3] = cmulc(x[4 + 3], S1, -C1);  x[8 + 3] = cmulc(x[8 + 3], -H, -H);   x[12 + 3] = cmulc(x[12 + 3], -C1, S1);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        float2 a0 = x[4 * k1], a1 = x[4 * k1 + 1], a2 = x[4 * k1 + 2], a3 = x[4 * k1 + 3];
        dft4r(a0, a1, a2, a3);
        o[k1] = a0; o[k1 + 4] = a1; o[k1 + 8] = a2; o[k1 + 12] = a3;
    }
}

// forward 8-point DFT, natural order in and out: even/odd 4-point DFTs, X[k] = E[k] + W8^k O[k]
__device__ __forceinline__ void dft8r(float2 (&x)[8]) {
    constexpr float H = 0.70710678118654752f;
    float2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    dft4r(e0, e1, e2, e3);
    dft4r(o0, o1, o2, o3);
    o1 = cmulc(o1, H, -H);
    o2 = make_float2(o2.y, -o2.x);
    o3 = cmulc(o3, -H, -H);
    x[0] = make_float2(e0.x + o0.x, e0.y + o0.y); x[4] = make_float2(e0.x - o0.x, e0.y - o0.y);
    x[1] = make_float2(e1.x + o1.x, e1.y + o1.y); x[5] = make_float2(e1.x - o1.x, e1.y - o1.y);
    x[2] = make_float2(e2.x + o2.x, e2.y + o2.y); x[6] = make_float2(e2.x - o2.x, e2.y - o2.y);
    x[3] = make_float2(e3.x + o3.x, e3.y + o3.y); x[7] = make_float2(e3.x - o3.x, e3.y - o3.y);
}
