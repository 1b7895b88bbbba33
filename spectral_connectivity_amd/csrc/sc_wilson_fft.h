// sc_wilson_fft.h -- the register-resident fp64 transform the Wilson kernels share (sc_wilson_fft.hip: one causal projection per
// launch over series in HBM; sc_wilson_pair.hip: the whole 2 x 2 iteration of a channel pair on one compute unit).
// N = 256 .. 4096 (powers of two): 16 points per thread, radix 16 x 16 x R3 with R3 = N / 256 in {1, 2, 4, 8, 16}; the first
// pass takes x[i + t N/16] and the last leaves X[i + t N/16] in the same thread's registers.  Twiddles W_N^m = lo[m & 63] *
// hi[m >> 6] from two small LDS tables.
#pragma once
#include "sc_common.h"

typedef double2 cd;

__device__ __forceinline__ cd zmul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cd zmulc(cd a, double c, double s) { return make_double2(a.x * c - a.y * s, a.x * s + a.y * c); }

__device__ __forceinline__ void zdft2(cd& a0, cd& a1) {
    const cd t = a0;
    a0 = make_double2(t.x + a1.x, t.y + a1.y);
    a1 = make_double2(t.x - a1.x, t.y - a1.y);
}
// forward 4-point DFT, natural order in and out
__device__ __forceinline__ void zdft4(cd& a0, cd& a1, cd& a2, cd& a3) {
    const cd b0 = make_double2(a0.x + a2.x, a0.y + a2.y), b1 = make_double2(a0.x - a2.x, a0.y - a2.y);
    const cd b2 = make_double2(a1.x + a3.x, a1.y + a3.y), b3 = make_double2(a1.y - a3.y, a3.x - a1.x);
    a0 = make_double2(b0.x + b2.x, b0.y + b2.y);
    a1 = make_double2(b1.x + b3.x, b1.y + b3.y);
    a2 = make_double2(b0.x - b2.x, b0.y - b2.y);
    a3 = make_double2(b1.x - b3.x, b1.y - b3.y);
}
// forward 8-point DFT, natural order in and out: even/odd 4-point DFTs, X[k] = E[k] + W8^k O[k]
__device__ __forceinline__ void zdft8(cd (&x)[8]) {
    constexpr double H = 0.70710678118654752440;
    cd e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    zdft4(e0, e1, e2, e3);
    zdft4(o0, o1, o2, o3);
    o1 = zmulc(o1, H, -H);
    o2 = make_double2(o2.y, -o2.x);
    o3 = zmulc(o3, -H, -H);
    x[0] = make_double2(e0.x + o0.x, e0.y + o0.y); x[4] = make_double2(e0.x - o0.x, e0.y - o0.y);
    x[1] = make_double2(e1.x + o1.x, e1.y + o1.y); x[5] = make_double2(e1.x - o1.x, e1.y - o1.y);
    x[2] = make_double2(e2.x + o2.x, e2.y + o2.y); x[6] = make_double2(e2.x - o2.x, e2.y - o2.y);
    x[3] = make_double2(e3.x + o3.x, e3.y + o3.y); x[7] = make_double2(e3.x - o3.x, e3.y - o3.y);
}
// forward 16-point DFT: x[n], n = 4 n1 + n2 in; o[k], k = k1 + 4 k2 out (both natural order)
__device__ __forceinline__ void zdft16(cd (&x)[16], cd (&o)[16]) {
    constexpr double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173, H = 0.70710678118654752440;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) zdft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
    x[4 + 1] = zmulc(x[4 + 1], C1, -S1);  x[8 + 1] = zmulc(x[8 + 1], H, -H);    x[12 + 1] = zmulc(x[12 + 1], S1, -C1);
    x[4 + 2] = zmulc(x[4 + 2], H, -H);    x[8 + 2] = make_double2(x[8 + 2].y, -x[8 + 2].x);
    x[12 + 2] = zmulc(x[12 + 2], -H, -H);
    x[4 + 3] = zmulc(x[4 + 3], S1, -C1);  x[8 + 3] = zmulc(x[8 + 3], -H, -H);   x[12 + 3] = zmulc(x[12 + 3], -C1, S1);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        cd a0 = x[4 * k1], a1 = x[4 * k1 + 1], a2 = x[4 * k1 + 2], a3 = x[4 * k1 + 3];
        zdft4(a0, a1, a2, a3);
        o[k1] = a0; o[k1 + 4] = a1; o[k1 + 8] = a2; o[k1 + 12] = a3;
    }
}

#define WF_PHYS(idx) ((idx) + ((idx) >> 4))

// One forward transform of the NF series of this workgroup: a[t] = x[i + t TPF] in, a[t] = X[i + t TPF] out.
template <int LOG2N>
__device__ __forceinline__ void wf_fft(cd (&a)[16], cd* zf, const cd* lo, const cd* hi, int i) {
    constexpr int N = 1 << LOG2N, TPF = N / 16, R3 = N / 256;
    cd o[16];
    auto W = [&](int m) -> cd { return zmul(lo[m & 63], hi[m >> 6]); };
    __syncthreads();                    // tables visible (first call); the previous transform's reads of z done
    zdft16(a, o);
#pragma unroll
    for (int u = 0; u < 16; ++u) zf[WF_PHYS(16 * i + u)] = o[u];
    __syncthreads();
    const int kk = i & 15;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const cd v = zf[WF_PHYS(i + t * TPF)];
        a[t] = (t == 0) ? v : zmul(v, W(t * kk * (N / 256)));
    }
    zdft16(a, o);
    if constexpr (R3 == 1) {
#pragma unroll
        for (int t = 0; t < 16; ++t) a[t] = o[t];         // X[i + 16 t]
        return;
    } else {
        __syncthreads();
        const int j = ((i - kk) << 4) + kk;
#pragma unroll
        for (int u = 0; u < 16; ++u) zf[WF_PHYS(j + 16 * u)] = o[u];
        __syncthreads();
        if constexpr (R3 == 16) {                          // one radix-16 butterfly per thread, ib = i
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const cd v = zf[WF_PHYS(i + t * 256)];
                a[t] = (t == 0) ? v : zmul(v, W(t * i));
            }
            zdft16(a, o);
#pragma unroll
            for (int t = 0; t < 16; ++t) a[t] = o[t];      // X[i + 256 t]
        } else {
            constexpr int NB3 = 16 / R3;
#pragma unroll
            for (int b = 0; b < NB3; ++b) {
                const int ib = i + b * TPF;                // 0..255
                cd r[R3];
#pragma unroll
                for (int t = 0; t < R3; ++t) {
                    const cd v = zf[WF_PHYS(ib + t * 256)];
                    r[t] = (t == 0) ? v : zmul(v, W(t * ib));
                }
                if constexpr (R3 == 2) zdft2(r[0], r[1]);
                if constexpr (R3 == 4) zdft4(r[0], r[1], r[2], r[3]);
                if constexpr (R3 == 8) zdft8(r);
#pragma unroll
                for (int u = 0; u < R3; ++u) a[b + NB3 * u] = r[u];     // X[ib + 256 u] = X[i + (b + NB3 u) TPF]
            }
        }
    }
}

