// sc_measure.hip -- measures epilogue: accumulated sums -> connectivity measures.
//
// One workgroup per (bin, upper-triangular 16x16 tile): the tile's planes are read once, coalesced,
// the measure is evaluated for the tile and for its mirror image (conjugate / sign rules per measure),
// and both 16x16 output blocks are written as 64-byte rows (the mirror through an LDS transpose).
// Divides by n_observations (AFTER any cross-GPU sum) and applies the reference's algebra literally,
// in fp64 (the epilogue touches W*F*C^2 elements once; it is HBM-bound and tiny next to stage B):
//   coherency   connectivity.py:632-657   S_ij / max(sqrt(P_i P_j), eps), diagonal NaN
//   coherence   connectivity.py:675-702   clip(|coherency|^2, 0, 1)
//   imag. coh.  connectivity.py:704-743   clip(|Im S_ij| / max(sqrt(P_i P_j), eps), 0, 1)
//   PLV / PPC   connectivity.py:897-931, :1129-1159
//   PLI / wPLI / debiased  connectivity.py:933-1127  (Im forced to 0 on the diagonal)
#include <math.h>
#include "sc_common.h"

#define SC_EPS64 2.220446049250313e-16

// records as float (f32 engine) or double (f64 engine) elements: the element type is a template parameter of the kernels
// here (the epilogue is launched on the headline path; a run-time switch per load cost 20 %)
template <typename AccT>
struct RecT {
    const AccT* p;
    __device__ RecT operator+(int64_t n) const { return RecT{p + n}; }
    __device__ double operator[](int64_t i) const { return (double)p[i]; }
};

// A record given as n_parts partial records `stride` elements apart (the blocks a rank received in the direct exchange of the
// trial-sharded path, parallel.py): element i = p[i] + p[i + stride] + ... summed in part (= rank) order in the records' own
// precision -- bit for bit the record a separate summation pass would have written, without the pass and its round trip.
template <typename AccT>
struct RecPartsT {
    const AccT* p;               // part 0
    const AccT* q;               // part 1; part k (k >= 1) at q + (k - 1) * stride
    int n_parts;
    int64_t stride;
    __device__ RecPartsT operator+(int64_t n) const { return RecPartsT{p + n, q + n, n_parts, stride}; }
    __device__ double operator[](int64_t i) const {
        AccT s = p[i];
        for (int k = 1; k < n_parts; ++k) s += q[i + (k - 1) * stride];
        return (double)s;
    }
};

struct MeasureArgs {
    ScRec accum;
    int n_parts;                 // measure_tile_multi_kernel: > 1 = the record is the sum of this many partial records: accum.p,
    const void* rest;            // ... then rest, rest + part_stride, ...
    int64_t part_stride;
    void* out;
    int64_t n_bins, floats_per_bin, total;
    int C, NB, n_tiles;
    int p_csm, p_abs, p_sq, p_sign, p_unit;  // plane offsets or -1
    double n_obs, rn_obs;        // observations per bin and 1 / that (one fp64 division on the host instead of one per entry and measure)
    int measure;
    int n_multi;                 // measure_tile_multi_kernel: the real-valued measures of one launch ...
    int multi[SC_MEASURE_MULTI_MAX];
    void* multi_out[SC_MEASURE_MULTI_MAX];      // ... and where each goes
};

template <typename Rec>
__device__ inline double tile_read(Rec bin_rec, int plane, int n_tiles, int NB, int i, int j,
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

// power: one thread per (bin, channel)
template <typename AccT, bool PARTS> struct RecSel { using type = RecT<AccT>; };
template <typename AccT> struct RecSel<AccT, true> { using type = RecPartsT<AccT>; };
template <typename AccT, bool PARTS>
__device__ __forceinline__ typename RecSel<AccT, PARTS>::type make_rec(const MeasureArgs& a) {
    if constexpr (PARTS) return RecPartsT<AccT>{(const AccT*)a.accum.p, (const AccT*)a.rest, a.n_parts, a.part_stride};
    else return RecT<AccT>{(const AccT*)a.accum.p};
}

template <typename OutT, typename AccT, bool PARTS = false>
__global__ void __launch_bounds__(256) power_kernel(MeasureArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.total) return;
    const int64_t bin = idx / a.C;
    const int i = (int)(idx - bin * a.C);
    bool m;
    const auto rec = make_rec<AccT, PARTS>(a) + bin * a.floats_per_bin;
    ((OutT*)a.out)[idx] = (OutT)(tile_read(rec, a.p_csm, a.n_tiles, a.NB, i, i, &m) / a.n_obs);
}

// raw (un-normalised) sums of one matrix entry (i, j), as stored for the upper triangle
struct MeasureIn {
    double s_re, s_im;     // sum x_i conj(x_j)
    double p_i, p_j;       // sum |x_i|^2, sum |x_j|^2
    double sa, sq, sg;     // sum |Im s|, sum (Im s)^2, sum sign Im s
    double u_re, u_im;     // sum s / |s|
};

// the entry (j, i) from the sums of (i, j): s and the unit phasors are conjugated, sign Im s flips
__device__ inline MeasureIn measure_mirror(MeasureIn v) {
    MeasureIn w = v;
    w.s_im = -v.s_im; w.u_im = -v.u_im; w.sg = -v.sg; w.p_i = v.p_j; w.p_j = v.p_i;
    return w;
}

// 1 / x for a normal positive double: v_rcp_f64 and two Newton steps (<= 1 ulp; the IEEE division sequence is three times as long)
__device__ inline double measure_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// one measure of one entry; complex measures return (re, im), real ones (value, 0)
__device__ inline double2 measure_value(int measure, double n, double rn, MeasureIn v, bool diag) {
    const double NaN = nan("");
    // (the expectation = sum * rn, rn = 1 / n from the host: no fp64 division per entry for it -- an fp64 division or square root is
    //  ~12-20 instructions at half rate, and they were HALF of this kernel's time: 0.131 ms for coherence + wPLI at cfg3 against 0.065 with
    //  the algebra switched off (-DMEASURE_AB_NOMATH); 1 ulp of fp64 apart from x / n)
    double s_re = v.s_re * rn, s_im = diag ? 0.0 : v.s_im * rn;
    const double p_i = v.p_i * rn, p_j = v.p_j * rn;
    switch (measure) {
    case SC_M_CSM:
        return make_double2(s_re, s_im);
    case SC_M_COHERENCY:
    case SC_M_COHERENCE_MAGNITUDE:
    case SC_M_COHERENCE_PHASE: {
        // 1 / max(sqrt(p_i p_j), eps) as ONE reciprocal square root (v_rsq_f64 + refinement) instead of a square root and a division
        const double pp = p_i * p_j;
        const double rden = pp > SC_EPS64 * SC_EPS64 ? rsqrt(pp) : 1.0 / SC_EPS64;
        double c_re = s_re * rden, c_im = s_im * rden;
        if (diag) { c_re = NaN; c_im = NaN; }
        if (measure == SC_M_COHERENCY) return make_double2(c_re, c_im);
        if (measure == SC_M_COHERENCE_MAGNITUDE) {
            const double mag = c_re * c_re + c_im * c_im;
            return make_double2((diag ? NaN : fmin(fmax(mag, 0.0), 1.0)), 0.0);
        }
        return make_double2((diag ? NaN : atan2(c_im, c_re)), 0.0);
    }
    case SC_M_IMAGINARY_COHERENCE: {
        const double den = fmax(sqrt(p_i * p_j), SC_EPS64);
        return make_double2(fmin(fmax(fabs(s_im / den), 0.0), 1.0), 0.0);
    }
    case SC_M_PLV:
        return make_double2((sqrt(v.u_re * v.u_re + v.u_im * v.u_im) * rn), 0.0);
    case SC_M_PLV_COMPLEX:
        return make_double2((v.u_re * rn), (v.u_im * rn));
    case SC_M_PPC:
        return make_double2(((v.u_re * v.u_re + v.u_im * v.u_im - n) / (n * n - n)), 0.0);
    case SC_M_PLI:
    case SC_M_DEBIASED_PLI2: {
        const double pli = (diag ? 0.0 : v.sg) * rn;
        return make_double2((measure == SC_M_PLI ? pli : (n * pli * pli - 1.0) / (n - 1.0)), 0.0);
    }
    case SC_M_WPLI: {
        double w = diag ? 0.0 : v.sa * rn;
        if (w < SC_EPS64) w = 1.0;
        return make_double2((s_im * measure_rcp(w)), 0.0);
    }
    case SC_M_DEBIASED_WPLI2: {
        const double si = s_im * n;
        const double sa = diag ? 0.0 : v.sa, sq = diag ? 0.0 : v.sq;
        double wgt = sa * sa - sq;
        if (wgt == 0.0 || n <= 1.0) wgt = NaN;
        return make_double2(((si * si - sq) / wgt), 0.0);
    }
    default:
        return make_double2(0.f, 0.0);
    }
}

template <bool COMPLEX_OUT, typename OutT, typename OutT2, typename AccT, bool PARTS = false>
__global__ void __launch_bounds__(256) measure_tile_kernel(MeasureArgs a) {
    __shared__ MeasureIn raw[256];
    __shared__ double2 mir[256];
    const int tid = threadIdx.x, ii = tid >> 4, jj = tid & 15;
    int ti = 0, len = a.NB, t = blockIdx.y;                 // upper-triangular tile (ti <= tj)
    while (t >= len) { t -= len; ++ti; --len; }
    const int tj = ti + t;
    const int64_t bin = blockIdx.x;
    const auto rec = make_rec<AccT, PARTS>(a) + bin * a.floats_per_bin;
    const int64_t plane = (int64_t)a.n_tiles * SC_TILE_ELEMS;
    const auto tile = rec + ((int64_t)blockIdx.y * SC_TILE_ELEMS + ii * 16 + jj);
    MeasureIn v = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (a.p_csm >= 0) {
        v.s_re = (double)tile[a.p_csm * plane];
        v.s_im = (double)tile[(a.p_csm + 1) * plane];
        v.p_i = (double)rec[a.p_csm * plane + (int64_t)sc_tile_index(ti, ti, a.NB) * SC_TILE_ELEMS + ii * 17];
        v.p_j = (double)rec[a.p_csm * plane + (int64_t)sc_tile_index(tj, tj, a.NB) * SC_TILE_ELEMS + jj * 17];
    }
    if (a.p_abs >= 0 && (a.measure == SC_M_WPLI || a.measure == SC_M_DEBIASED_WPLI2)) v.sa = (double)tile[a.p_abs * plane];
    if (a.p_sq >= 0 && a.measure == SC_M_DEBIASED_WPLI2) v.sq = (double)tile[a.p_sq * plane];
    if (a.p_sign >= 0 && (a.measure == SC_M_PLI || a.measure == SC_M_DEBIASED_PLI2)) v.sg = (double)tile[a.p_sign * plane];
    if (a.p_unit >= 0 && (a.measure == SC_M_PLV || a.measure == SC_M_PLV_COMPLEX || a.measure == SC_M_PPC)) {
        v.u_re = (double)tile[a.p_unit * plane];
        v.u_im = (double)tile[(a.p_unit + 1) * plane];
    }
    const bool dtile = ti == tj;
    if (dtile) {
        // lower triangle inside a diagonal tile comes from its mirror: the matrix cores fill the tile
        // completely, but (i,j) and (j,i) differ by rounding; using one of them keeps every measure
        // exactly (anti)symmetric like the reference
        raw[tid] = v;
        __syncthreads();
        if (ii > jj) v = measure_mirror(raw[jj * 16 + ii]);
    }
    const int i = ti * 16 + ii, j = tj * 16 + jj;
    OutT* outf = (OutT*)a.out;
    OutT2* outc = (OutT2*)a.out;
    const int64_t obase = bin * (int64_t)a.C * a.C;
    const double2 direct = measure_value(a.measure, a.n_obs, a.rn_obs, v, i == j);
    if (i < a.C && j < a.C) {
        if (COMPLEX_OUT) outc[obase + (int64_t)i * a.C + j] = OutT2{(OutT)direct.x, (OutT)direct.y};
        else outf[obase + (int64_t)i * a.C + j] = (OutT)direct.x;
    }
    if (!dtile) {
        mir[jj * 16 + ii] = measure_value(a.measure, a.n_obs, a.rn_obs, measure_mirror(v), false);
        __syncthreads();
        const int r = tj * 16 + ii, c = ti * 16 + jj;       // thread (ii, jj) now owns row ii of the mirrored block
        if (r < a.C && c < a.C) {
            if (COMPLEX_OUT) outc[obase + (int64_t)r * a.C + c] = OutT2{(OutT)mir[tid].x, (OutT)mir[tid].y};
            else outf[obase + (int64_t)r * a.C + c] = (OutT)mir[tid].x;
        }
    }
}

// Several real-valued measures of the same record in ONE launch: the tile's planes are read once (every plane any of the
// measures needs), each measure is evaluated for the tile and its mirror image like measure_tile_kernel does.
#ifndef MEASURE_MULTI_TPW
#define MEASURE_MULTI_TPW 1       // tiles per workgroup.  (Four -- "a tile is too little per launch slot" -- measured slower: two measures
                                  // at the cfg3 shape 0.139 ms with 4, 0.142 with 2, 0.143 with 9, 0.118 with ONE: the tiles of a
                                  // workgroup run one after the other, each behind its own loads, and 32 508 small workgroups hide
                                  // that latency better than 8 127 longer ones; tools/measure_ab.py over variant libraries)
#endif
template <typename OutT, typename AccT, bool PARTS = false>
__global__ void __launch_bounds__(256) measure_tile_multi_kernel(MeasureArgs a) {
    __shared__ MeasureIn raw[256];
    __shared__ double mir[256];
    const int tid = threadIdx.x, ii = tid >> 4, jj = tid & 15;
    const int64_t bin = blockIdx.x;
    using Rec = typename RecSel<AccT, PARTS>::type;
    const Rec rec = make_rec<AccT, PARTS>(a) + bin * a.floats_per_bin;
    const int64_t plane = (int64_t)a.n_tiles * SC_TILE_ELEMS;
    const int64_t obase = bin * (int64_t)a.C * a.C;
    int ti = 0, len = a.NB, t = blockIdx.y * MEASURE_MULTI_TPW;          // upper-triangular tile (ti <= tj), row-major
    while (t >= len) { t -= len; ++ti; --len; }
    for (int u = 0; u < MEASURE_MULTI_TPW; ++u) {
        const int tile_id = blockIdx.y * MEASURE_MULTI_TPW + u;
        if (tile_id >= a.n_tiles) break;
        const int tj = ti + t;
        const Rec tile = rec + ((int64_t)tile_id * SC_TILE_ELEMS + ii * 16 + jj);
        MeasureIn v = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (a.p_csm >= 0) {
            v.s_re = (double)tile[a.p_csm * plane];
            v.s_im = (double)tile[(a.p_csm + 1) * plane];
            v.p_i = (double)rec[a.p_csm * plane + (int64_t)sc_tile_index(ti, ti, a.NB) * SC_TILE_ELEMS + ii * 17];
            v.p_j = (double)rec[a.p_csm * plane + (int64_t)sc_tile_index(tj, tj, a.NB) * SC_TILE_ELEMS + jj * 17];
        }
        if (a.p_abs >= 0) v.sa = (double)tile[a.p_abs * plane];
        if (a.p_sq >= 0) v.sq = (double)tile[a.p_sq * plane];
        if (a.p_sign >= 0) v.sg = (double)tile[a.p_sign * plane];
        if (a.p_unit >= 0) {
            v.u_re = (double)tile[a.p_unit * plane];
            v.u_im = (double)tile[(a.p_unit + 1) * plane];
        }
        const bool dtile = ti == tj;
        if (dtile) {
            raw[tid] = v;
            __syncthreads();
            if (ii > jj) v = measure_mirror(raw[jj * 16 + ii]);
            __syncthreads();
        }
        const int i = ti * 16 + ii, j = tj * 16 + jj;
        for (int m = 0; m < a.n_multi; ++m) {
            OutT* outf = (OutT*)a.multi_out[m];
            const int w = a.multi[m];
#ifdef MEASURE_AB_NOMATH      // (A/B: what the loads, the LDS mirror and the stores cost without the fp64 algebra; results wrong)
            const double direct = v.s_re + v.sa + w;
#else
            const double direct = measure_value(w, a.n_obs, a.rn_obs, v, i == j).x;
#endif
            if (i < a.C && j < a.C) outf[obase + (int64_t)i * a.C + j] = (OutT)direct;
            if (!dtile) {
                // entry (j, i): every real-valued measure is even or odd under (i, j) -> (j, i) (conjugated s, swapped
                // powers), bit for bit what measure_value(measure_mirror(v)) returns -- the fp64 algebra (sqrt, divisions)
                // is what this kernel spends its time on, so it runs once per pair
                const double sgn = (w == SC_M_COHERENCE_PHASE || w == SC_M_PLI || w == SC_M_WPLI) ? -1.0 : 1.0;
                mir[jj * 16 + ii] = sgn * direct;
                __syncthreads();
                const int r = tj * 16 + ii, c = ti * 16 + jj;
                if (r < a.C && c < a.C) outf[obase + (int64_t)r * a.C + c] = (OutT)mir[tid];
                __syncthreads();
            }
        }
        if (++t >= len) { t = 0; ++ti; --len; }
    }
}

static uint32_t measure_needs(int measure) {
    switch (measure) {
    case SC_M_POWER: case SC_M_CSM: case SC_M_COHERENCY: case SC_M_COHERENCE_MAGNITUDE:
    case SC_M_COHERENCE_PHASE: case SC_M_IMAGINARY_COHERENCE: return SC_PLANE_CSM;
    case SC_M_PLV: case SC_M_PLV_COMPLEX: case SC_M_PPC: return SC_PLANE_UNIT;
    case SC_M_PLI: case SC_M_DEBIASED_PLI2: return SC_PLANE_SIGN_IM;
    case SC_M_WPLI: return SC_PLANE_CSM | SC_PLANE_ABS_IM;
    case SC_M_DEBIASED_WPLI2: return SC_PLANE_CSM | SC_PLANE_ABS_IM | SC_PLANE_IM_SQ;
    }
    return 0;
}

static int measure_multi_run(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                             int64_t n_observations, int n_measures, const int* measures, void* const* d_outs,
                             bool wide, void* stream, int n_parts = 1, const void* d_rest = nullptr, int64_t part_stride = 0) {
    ScTimed timed_("measure_epilogue", stream);
    SC_REQUIRE(d_accum && measures && d_outs, "NULL argument");
    SC_REQUIRE(n_bins >= 1 && n_signals >= 1 && n_observations >= 1, "dimensions must be positive");
    SC_REQUIRE(n_measures >= 1 && n_measures <= SC_MEASURE_MULTI_MAX, "1 ... SC_MEASURE_MULTI_MAX measures per launch");
    SC_REQUIRE(n_parts >= 1 && (n_parts == 1 || d_rest != nullptr) && (n_parts <= 2 || part_stride > 0), "bad partial-record layout");
    MeasureArgs a;
    a.accum = sc_rec(d_accum, planes);
    a.n_parts = n_parts;
    a.rest = d_rest;
    a.part_stride = part_stride;
    a.out = nullptr;
    a.n_bins = n_bins;
    a.C = (int)n_signals;
    a.NB = sc_n_blocks(n_signals);
    a.n_tiles = sc_n_tiles(a.NB);
    a.floats_per_bin = (int64_t)sc_plane_count(planes) * a.n_tiles * SC_TILE_ELEMS;
    uint32_t need = 0;
    for (int m = 0; m < n_measures; ++m) {
        const int w = measures[m];
        const bool real_matrix = w == SC_M_COHERENCE_MAGNITUDE || w == SC_M_COHERENCE_PHASE || w == SC_M_IMAGINARY_COHERENCE ||
                                 w == SC_M_PLV || w == SC_M_PPC || w == SC_M_PLI || w == SC_M_DEBIASED_PLI2 ||
                                 w == SC_M_WPLI || w == SC_M_DEBIASED_WPLI2;
        if (!real_matrix) {
            sc_set_error("sc_measure_multi: measure %d is not a real-valued C x C measure (use sc_measure_f32 / _f64)", w);
            return SC_EINVAL;
        }
        SC_REQUIRE(d_outs[m] != nullptr, "NULL output");
        need |= measure_needs(w);
        a.multi[m] = w;
        a.multi_out[m] = d_outs[m];
    }
    a.n_multi = n_measures;
    if ((planes & need) != need) {
        sc_set_error("measures need accumulator planes 0x%x, record has 0x%x", need, planes);
        return SC_EINVAL;
    }
    // only the planes some measure reads are fetched
    a.p_csm = (need & SC_PLANE_CSM) ? sc_plane_offset(planes, SC_PLANE_CSM) : -1;
    a.p_abs = (need & SC_PLANE_ABS_IM) ? sc_plane_offset(planes, SC_PLANE_ABS_IM) : -1;
    a.p_sq = (need & SC_PLANE_IM_SQ) ? sc_plane_offset(planes, SC_PLANE_IM_SQ) : -1;
    a.p_sign = (need & SC_PLANE_SIGN_IM) ? sc_plane_offset(planes, SC_PLANE_SIGN_IM) : -1;
    a.p_unit = (need & SC_PLANE_UNIT) ? sc_plane_offset(planes, SC_PLANE_UNIT) : -1;
    a.n_obs = (double)n_observations;
    a.rn_obs = 1.0 / a.n_obs;
    a.measure = measures[0];
    a.total = n_bins * n_signals * n_signals;
    SC_REQUIRE(n_bins < (int64_t)1 << 31 && a.n_tiles <= 65535, "output too large for one launch");
    const dim3 grid((unsigned)n_bins, (unsigned)((a.n_tiles + MEASURE_MULTI_TPW - 1) / MEASURE_MULTI_TPW));
    hipStream_t st = (hipStream_t)stream;
    if (n_parts > 1) {
        if (a.accum.f64) {
            if (wide) hipLaunchKernelGGL((measure_tile_multi_kernel<double, double, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((measure_tile_multi_kernel<float, double, true>), grid, dim3(256), 0, st, a);
        } else {
            if (wide) hipLaunchKernelGGL((measure_tile_multi_kernel<double, float, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((measure_tile_multi_kernel<float, float, true>), grid, dim3(256), 0, st, a);
        }
        SC_CHECK_HIP(hipGetLastError());
        return SC_OK;
    }
    if (a.accum.f64) {
        if (wide) hipLaunchKernelGGL((measure_tile_multi_kernel<double, double>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((measure_tile_multi_kernel<float, double>), grid, dim3(256), 0, st, a);
    } else {
        if (wide) hipLaunchKernelGGL((measure_tile_multi_kernel<double, float>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((measure_tile_multi_kernel<float, float>), grid, dim3(256), 0, st, a);
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

extern "C" int sc_measure_multi_f32(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                                    int64_t n_observations, int n_measures, const int* measures, void* const* d_outs,
                                    void* stream) {
    return measure_multi_run(d_accum, n_bins, n_signals, planes, n_observations, n_measures, measures, d_outs, false, stream);
}
// The same measures from a record that arrives as n_parts partial records (SC_RECORD_F64 in `planes`: doubles): part 0 at
// d_part0, part k >= 1 at d_rest + (k - 1) * part_stride elements; summed in part order while they are read.
extern "C" int sc_measure_multi_parts(const void* d_part0, const void* d_rest, int n_parts, int64_t part_stride, int64_t n_bins,
                                      int64_t n_signals, uint32_t planes, int64_t n_observations, int n_measures,
                                      const int* measures, void* const* d_outs, int wide, void* stream) {
    return measure_multi_run(d_part0, n_bins, n_signals, planes, n_observations, n_measures, measures, d_outs, wide != 0, stream,
                             n_parts, d_rest, part_stride);
}
extern "C" int sc_measure_multi_f64(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                                    int64_t n_observations, int n_measures, const int* measures, void* const* d_outs,
                                    void* stream) {
    return measure_multi_run(d_accum, n_bins, n_signals, planes, n_observations, n_measures, measures, d_outs, true, stream);
}

static int measure_run(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                       int64_t n_observations, int measure, void* d_out, bool wide, void* stream,
                       int n_parts = 1, const void* d_rest = nullptr, int64_t part_stride = 0) {
    ScTimed timed_("measure_epilogue", stream);
    SC_REQUIRE(d_accum && d_out, "NULL argument");
    SC_REQUIRE(n_parts >= 1 && (n_parts == 1 || d_rest != nullptr) && (n_parts <= 2 || part_stride > 0), "bad partial-record layout");
    SC_REQUIRE(n_bins >= 1 && n_signals >= 1 && n_observations >= 1, "dimensions must be positive");
    SC_REQUIRE(measure >= SC_M_POWER && measure <= SC_M_PLV_COMPLEX, "unknown measure");
    MeasureArgs a;
    a.accum = sc_rec(d_accum, planes);
    a.n_parts = n_parts;
    a.rest = d_rest;
    a.part_stride = part_stride;
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
    a.rn_obs = 1.0 / a.n_obs;
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
    if (measure == SC_M_POWER) {
        a.total = n_bins * n_signals;
        const int64_t blocks = (a.total + 255) / 256;
        SC_REQUIRE(blocks < (int64_t)1 << 31, "output too large for one launch");
        hipStream_t st = (hipStream_t)stream;
        const dim3 g((unsigned)blocks);
        if (n_parts > 1) {
            if (wide && a.accum.f64) hipLaunchKernelGGL((power_kernel<double, double, true>), g, dim3(256), 0, st, a);
            else if (wide) hipLaunchKernelGGL((power_kernel<double, float, true>), g, dim3(256), 0, st, a);
            else if (a.accum.f64) hipLaunchKernelGGL((power_kernel<float, double, true>), g, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((power_kernel<float, float, true>), g, dim3(256), 0, st, a);
        } else if (wide && a.accum.f64) hipLaunchKernelGGL((power_kernel<double, double>), g, dim3(256), 0, st, a);
        else if (wide) hipLaunchKernelGGL((power_kernel<double, float>), g, dim3(256), 0, st, a);
        else if (a.accum.f64) hipLaunchKernelGGL((power_kernel<float, double>), g, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((power_kernel<float, float>), g, dim3(256), 0, st, a);
    } else {
        a.total = n_bins * n_signals * n_signals;
        SC_REQUIRE(n_bins < (int64_t)1 << 31 && a.n_tiles <= 65535, "output too large for one launch");
        const dim3 grid((unsigned)n_bins, (unsigned)a.n_tiles);
        const bool cplx = measure == SC_M_CSM || measure == SC_M_COHERENCY || measure == SC_M_PLV_COMPLEX;
        hipStream_t st = (hipStream_t)stream;
#define MT_LAUNCH(C, O, O2, A)                                                                                  \
    do {                                                                                                        \
        if (n_parts > 1) hipLaunchKernelGGL((measure_tile_kernel<C, O, O2, A, true>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((measure_tile_kernel<C, O, O2, A>), grid, dim3(256), 0, st, a);                 \
    } while (0)
        if (a.accum.f64) {
            if (cplx && wide) MT_LAUNCH(true, double, double2, double);
            else if (cplx) MT_LAUNCH(true, float, float2, double);
            else if (wide) MT_LAUNCH(false, double, double2, double);
            else MT_LAUNCH(false, float, float2, double);
        } else {
            if (cplx && wide) MT_LAUNCH(true, double, double2, float);
            else if (cplx) MT_LAUNCH(true, float, float2, float);
            else if (wide) MT_LAUNCH(false, double, double2, float);
            else MT_LAUNCH(false, float, float2, float);
        }
#undef MT_LAUNCH
    }
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

extern "C" int sc_measure_f32(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                              int64_t n_observations, int measure, void* d_out, void* stream) {
    return measure_run(d_accum, n_bins, n_signals, planes, n_observations, measure, d_out, false, stream);
}

// One measure -- power and the complex-valued ones included -- from a record that arrives as n_parts partial records (layout as for
// sc_measure_multi_parts), summed in part order while they are read.  wide: double / complex128 output.
extern "C" int sc_measure_parts(const void* d_part0, const void* d_rest, int n_parts, int64_t part_stride, int64_t n_bins,
                                int64_t n_signals, uint32_t planes, int64_t n_observations, int measure, void* d_out, int wide,
                                void* stream) {
    return measure_run(d_part0, n_bins, n_signals, planes, n_observations, measure, d_out, wide != 0, stream, n_parts, d_rest, part_stride);
}

// the same measures written as double / complex128: what the reference returns, without a widening pass
extern "C" int sc_measure_f64(const void* d_accum, int64_t n_bins, int64_t n_signals, uint32_t planes,
                              int64_t n_observations, int measure, void* d_out, void* stream) {
    return measure_run(d_accum, n_bins, n_signals, planes, n_observations, measure, d_out, true, stream);
}
