// sc_measure.hip -- measures epilogue: accumulated sums -> connectivity measures.
//
// One thread per output element (bin, i, j).  Reads the packed upper-triangular tile
// records (mirroring with conjugation for i-block > j-block), divides by n_observations
// (AFTER any cross-GPU sum), and applies the reference's algebra literally, in fp64
// (the epilogue touches W*F*C^2 elements once; it is HBM-bound and tiny next to stage B):
//   coherency   connectivity.py:632-657   S_ij / max(sqrt(P_i P_j), eps), diagonal NaN
//   coherence   connectivity.py:675-702   clip(|coherency|^2, 0, 1)
//   imag. coh.  connectivity.py:704-743   clip(|Im S_ij| / max(sqrt(P_i P_j), eps), 0, 1)
//   PLV / PPC   connectivity.py:897-931, :1129-1159
//   PLI / wPLI / debiased  connectivity.py:933-1127  (Im forced to 0 on the diagonal)
#include <math.h>
#include "sc_common.h"

#define SC_EPS64 2.220446049250313e-16

struct MeasureArgs {
    const float* accum;
    void* out;
    int64_t n_bins, floats_per_bin, total;
    int C, NB, n_tiles;
    int p_csm, p_abs, p_sq, p_sign, p_unit;  // plane offsets or -1
    double n_obs;
    int measure;
};

__device__ inline float tile_read(const float* bin_rec, int plane, int n_tiles, int NB, int i, int j,
                                  bool* mirrored) {
    int ti = i >> 4, tj = j >> 4, ii = i & 15, jj = j & 15;
    // lower triangle (tile-wise AND inside diagonal tiles) is read from its mirror: the matrix
    // cores fill diagonal tiles completely, but (i,j) and (j,i) there differ by rounding; using
    // one of them keeps every measure exactly (anti)symmetric like the reference.
    bool m = (ti > tj) || (ti == tj && ii > jj);
    if (m) { int t = ti; ti = tj; tj = t; t = ii; ii = jj; jj = t; }
    *mirrored = m;
    return bin_rec[((int64_t)plane * n_tiles + sc_tile_index(ti, tj, NB)) * SC_TILE_ELEMS + ii * 16 + jj];
}

__global__ void __launch_bounds__(256) measure_kernel(MeasureArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.total) return;
    const double NaN = nan("");
    const double n = a.n_obs;
    if (a.measure == SC_M_POWER) {
        const int64_t bin = idx / a.C;
        const int i = (int)(idx - bin * a.C);
        bool m;
        const float* rec = a.accum + bin * a.floats_per_bin;
        ((float*)a.out)[idx] = (float)((double)tile_read(rec, a.p_csm, a.n_tiles, a.NB, i, i, &m) / n);
        return;
    }
    const int64_t CC = (int64_t)a.C * a.C;
    const int64_t bin = idx / CC;
    const int rem = (int)(idx - bin * CC);
    const int i = rem / a.C, j = rem - i * a.C;
    const bool diag = i == j;
    const float* rec = a.accum + bin * a.floats_per_bin;
    bool m = false;
    double s_re = 0, s_im = 0, p_i = 0, p_j = 0;
    if (a.p_csm >= 0) {
        s_re = (double)tile_read(rec, a.p_csm, a.n_tiles, a.NB, i, j, &m) / n;
        s_im = (double)tile_read(rec, a.p_csm + 1, a.n_tiles, a.NB, i, j, &m) / n;
        if (m) s_im = -s_im;
        if (diag) s_im = 0.0;
        bool mm;
        p_i = (double)tile_read(rec, a.p_csm, a.n_tiles, a.NB, i, i, &mm) / n;
        p_j = (double)tile_read(rec, a.p_csm, a.n_tiles, a.NB, j, j, &mm) / n;
    }
    float* outf = (float*)a.out;
    float2* outc = (float2*)a.out;
    switch (a.measure) {
    case SC_M_CSM:
        outc[idx] = make_float2((float)s_re, (float)s_im);
        break;
    case SC_M_COHERENCY:
    case SC_M_COHERENCE_MAGNITUDE:
    case SC_M_COHERENCE_PHASE: {
        const double den = fmax(sqrt(p_i * p_j), SC_EPS64);
        double c_re = s_re / den, c_im = s_im / den;
        if (diag) { c_re = NaN; c_im = NaN; }
        if (a.measure == SC_M_COHERENCY) outc[idx] = make_float2((float)c_re, (float)c_im);
        else if (a.measure == SC_M_COHERENCE_MAGNITUDE) {
            double mag = c_re * c_re + c_im * c_im;
            outf[idx] = (float)(diag ? NaN : fmin(fmax(mag, 0.0), 1.0));
        } else outf[idx] = (float)(diag ? NaN : atan2(c_im, c_re));
        break;
    }
    case SC_M_IMAGINARY_COHERENCE: {
        const double den = fmax(sqrt(p_i * p_j), SC_EPS64);
        outf[idx] = (float)fmin(fmax(fabs(s_im / den), 0.0), 1.0);
        break;
    }
    case SC_M_PLV:
    case SC_M_PLV_COMPLEX:
    case SC_M_PPC: {
        double u_re = (double)tile_read(rec, a.p_unit, a.n_tiles, a.NB, i, j, &m);
        double u_im = (double)tile_read(rec, a.p_unit + 1, a.n_tiles, a.NB, i, j, &m);
        if (m) u_im = -u_im;
        if (a.measure == SC_M_PPC) outf[idx] = (float)((u_re * u_re + u_im * u_im - n) / (n * n - n));
        else if (a.measure == SC_M_PLV) outf[idx] = (float)(sqrt(u_re * u_re + u_im * u_im) / n);
        else outc[idx] = make_float2((float)(u_re / n), (float)(u_im / n));
        break;
    }
    case SC_M_PLI:
    case SC_M_DEBIASED_PLI2: {
        double sg = (double)tile_read(rec, a.p_sign, a.n_tiles, a.NB, i, j, &m);
        if (m) sg = -sg;
        if (diag) sg = 0.0;
        const double pli = sg / n;
        outf[idx] = (float)(a.measure == SC_M_PLI ? pli : (n * pli * pli - 1.0) / (n - 1.0));
        break;
    }
    case SC_M_WPLI: {
        double w = diag ? 0.0 : (double)tile_read(rec, a.p_abs, a.n_tiles, a.NB, i, j, &m) / n;
        if (w < SC_EPS64) w = 1.0;
        outf[idx] = (float)(s_im / w);
        break;
    }
    case SC_M_DEBIASED_WPLI2: {
        const double si = s_im * n;
        const double sa = diag ? 0.0 : (double)tile_read(rec, a.p_abs, a.n_tiles, a.NB, i, j, &m);
        const double sq = diag ? 0.0 : (double)tile_read(rec, a.p_sq, a.n_tiles, a.NB, i, j, &m);
        double wgt = sa * sa - sq;
        if (wgt == 0.0 || n <= 1.0) wgt = NaN;
        outf[idx] = (float)((si * si - sq) / wgt);
        break;
    }
    default:
        break;
    }
}

extern "C" int sc_measure_f32(const float* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                              int64_t n_observations, int measure, void* d_out, void* stream) {
    SC_REQUIRE(d_accum && d_out, "NULL argument");
    SC_REQUIRE(n_bins >= 1 && n_signals >= 1 && n_observations >= 1, "dimensions must be positive");
    SC_REQUIRE(measure >= SC_M_POWER && measure <= SC_M_PLV_COMPLEX, "unknown measure");
    MeasureArgs a;
    a.accum = d_accum;
    a.out = d_out;
    a.n_bins = n_bins;
    a.C = (int)n_signals;
    a.NB = sc_n_blocks(n_signals);
    a.n_tiles = sc_n_tiles(a.NB);
    a.floats_per_bin = (int64_t)sc_plane_count(planes) * a.n_tiles * SC_TILE_ELEMS;
    a.p_csm = (planes & SC_PLANE_CSM) ? sc_plane_offset(planes, SC_PLANE_CSM) : -1;
    a.p_abs = (planes & SC_PLANE_ABS_IM) ? sc_plane_offset(planes, SC_PLANE_ABS_IM) : -1;
    a.p_sq = (planes & SC_PLANE_IM_SQ) ? sc_plane_offset(planes, SC_PLANE_IM_SQ) : -1;
    a.p_sign = (planes & SC_PLANE_SIGN_IM) ? sc_plane_offset(planes, SC_PLANE_SIGN_IM) : -1;
    a.p_unit = (planes & SC_PLANE_UNIT) ? sc_plane_offset(planes, SC_PLANE_UNIT) : -1;
    a.n_obs = (double)n_observations;
    a.measure = measure;
    uint32_t need = 0;
    switch (measure) {
    case SC_M_POWER: case SC_M_CSM: case SC_M_COHERENCY: case SC_M_COHERENCE_MAGNITUDE:
    case SC_M_COHERENCE_PHASE: case SC_M_IMAGINARY_COHERENCE: need = SC_PLANE_CSM; break;
    case SC_M_PLV: case SC_M_PLV_COMPLEX: case SC_M_PPC: need = SC_PLANE_UNIT; break;
    case SC_M_PLI: case SC_M_DEBIASED_PLI2: need = SC_PLANE_SIGN_IM; break;
    case SC_M_WPLI: need = SC_PLANE_CSM | SC_PLANE_ABS_IM; break;
    case SC_M_DEBIASED_WPLI2: need = SC_PLANE_CSM | SC_PLANE_ABS_IM | SC_PLANE_IM_SQ; break;
    }
    if ((planes & need) != need) {
        sc_set_error("measure %d needs accumulator planes 0x%x, record has 0x%x", measure, need, planes);
        return SC_EINVAL;
    }
    a.total = measure == SC_M_POWER ? n_bins * n_signals : n_bins * n_signals * n_signals;
    const int64_t blocks = (a.total + 255) / 256;
    SC_REQUIRE(blocks < (int64_t)1 << 31, "output too large for one launch");
    hipLaunchKernelGGL(measure_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}
