// sc_stage.h -- global -> register -> LDS staging of observation rows, shared by the
// MFMA cross-spectral kernel and the VALU non-linear kernel.
//
// An "observation row" is the C complex coefficients X[o][0..C) of one (window, trial,
// taper) at one frequency bin: contiguous in HBM (channel stride 1), so a wave reads it as
// one coalesced segment (16 B/lane when C is even).  Rows are zero padded to CP channels in
// LDS and rows past n_obs are zero, so the consumers never branch on edges.
// LDS row stride RS (floats) is chosen with RS % 64 == 32 so that the two 16-lane halves of
// a ds_read_b64 fragment read (rows o, o+1) fall into different bank halves.
#pragma once
#include "sc_common.h"

struct ScStage {
    const float2* base;   // X + f*sF + group offset
    ScAxes ax;
    int64_t obs_stride;   // >0: offset(o) = o*obs_stride (reduced axes are contiguous)
    int C;                // real channels
    int CP;               // padded channels staged (multiple of 16 or 32)
    int RS;               // LDS row stride in floats
    int n_obs;
};

__host__ __device__ inline int sc_row_stride(int cp) {
    int two = 2 * cp;
    int pad = (32 - (two % 64) + 64) % 64;
    return two + pad;
}

__device__ inline int64_t sc_stage_obs_offset(const ScStage& st, int o) {
    return st.obs_stride > 0 ? (int64_t)o * st.obs_stride : sc_obs_offset(st.ax, o);
}

// E = number of float4 (VEC) or float2 (!VEC) elements each of the 256 threads moves.
template <int OC, int CPMAX, bool VEC>
struct ScStageRegs {
    static constexpr int E = VEC ? (OC * CPMAX / 2 / 256) : (OC * CPMAX / 256);
    float4 v4[VEC ? E : 1];
    float2 v2[VEC ? 1 : E];
};

template <int OC, int CPMAX, bool VEC>
__device__ inline void sc_stage_load(const ScStage& st, int o0, int tid, ScStageRegs<OC, CPMAX, VEC>& r) {
    if constexpr (VEC) {
        const int half = st.CP >> 1;            // float4 elements per row
        const int total = OC * half;
#pragma unroll
        for (int i = 0; i < ScStageRegs<OC, CPMAX, VEC>::E; ++i) {
            const int e = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < total) {
                const int row = e / half, q = e - row * half;
                const int o = o0 + row, c = 2 * q;
                if (o < st.n_obs && c < st.C)   // C even: c+1 < C too
                    v = *reinterpret_cast<const float4*>(st.base + sc_stage_obs_offset(st, o) + c);
            }
            r.v4[i] = v;
        }
    } else {
        const int total = OC * st.CP;
#pragma unroll
        for (int i = 0; i < ScStageRegs<OC, CPMAX, VEC>::E; ++i) {
            const int e = tid + i * 256;
            float2 v = make_float2(0.f, 0.f);
            if (e < total) {
                const int row = e / st.CP, c = e - row * st.CP;
                const int o = o0 + row;
                if (o < st.n_obs && c < st.C) v = st.base[sc_stage_obs_offset(st, o) + c];
            }
            r.v2[i] = v;
        }
    }
}

template <int OC, int CPMAX, bool VEC>
__device__ inline void sc_stage_store(const ScStage& st, float* lds, int tid,
                                      const ScStageRegs<OC, CPMAX, VEC>& r) {
    if constexpr (VEC) {
        const int half = st.CP >> 1;
        const int total = OC * half;
#pragma unroll
        for (int i = 0; i < ScStageRegs<OC, CPMAX, VEC>::E; ++i) {
            const int e = tid + i * 256;
            if (e < total) {
                const int row = e / half, q = e - row * half;
                *reinterpret_cast<float4*>(lds + row * st.RS + 4 * q) = r.v4[i];
            }
        }
    } else {
        const int total = OC * st.CP;
#pragma unroll
        for (int i = 0; i < ScStageRegs<OC, CPMAX, VEC>::E; ++i) {
            const int e = tid + i * 256;
            if (e < total) {
                const int row = e / st.CP, c = e - row * st.CP;
                *reinterpret_cast<float2*>(lds + row * st.RS + 2 * c) = r.v2[i];
            }
        }
    }
}

// Host side: can 16-byte loads be used, and are the reduced axes one linear run?
inline bool sc_stage_vec_ok(const void* X, const ScAxes& a) {
    return (a.C % 2 == 0) && (a.sW % 2 == 0) && (a.sR % 2 == 0) && (a.sK % 2 == 0) &&
           (a.sF % 2 == 0) && (((uintptr_t)X) % 16 == 0);
}
inline int64_t sc_stage_linear_stride(const ScAxes& a) {
    // observation index o = ((ow*rR)+or)*rK+ok ; linear iff strides nest exactly
    int64_t s = 0;       // stride of the innermost reduced axis
    int64_t span = 1;
    bool ok = true;
    if (a.rK > 1) { s = a.sK; span = a.rK; }
    if (a.rR > 1) {
        if (span == 1) { s = a.sR; span = a.rR; }
        else { ok = ok && (a.sR == s * span); span *= a.rR; }
    }
    if (a.rW > 1) {
        if (span == 1) { s = a.sW; span = a.rW; }
        else { ok = ok && (a.sW == s * span); span *= a.rW; }
    }
    if (span == 1) return 1;  // single observation: any positive stride works
    return (ok && s > 0) ? s : 0;
}
