// sc_wilson_fft.hip -- the causal projection of Wilson's iteration in one kernel.
//
// Each Wilson iteration takes every series A[problem][entry][0..N) through
//     a = ifft_n(A);  a[0] *= 1/2, strict lower triangle of a[0] = 0, a[n >= (N+1)/2] = 0;  A+ = fft_n(a)
// (minimum_phase_decomposition.py:96-142, called from :301-305).  As two library transforms with a
// pointwise kernel between them that is six passes over the series in HBM; here a series is read
// once, both transforms run out of LDS/registers in fp64, and A+ is written once.  Series of
// problems that already converged are skipped (the library batch cannot do that).
//
// N = 256 .. 4096 (powers of two): 16 * R3 threads per series, radix 16 x 16 x R3 with R3 = N/256
// in {1, 2, 4, 8, 16}; 256 threads = 4096 points per workgroup whatever N is.  The first radix-16
// pass takes x[i + t N/16] and the last leaves X[i + t N/16] in the same thread's registers, so the
// causal mask between the inverse and the forward transform needs no exchange.  The inverse is
// conj(fft(conj(.))).  Twiddles W_N^m = lo[m & 63] * hi[m >> 6] from two small LDS tables filled
// with sincospi at kernel start.  Other lengths stay on rocFFT + a pointwise kernel in the callers.
#include <stdlib.h>
#include <string.h>
#include "sc_common.h"

#include "sc_wilson_fft.h"

// NAT: the series of a problem in the NATURAL layout of the matrix kernels, A[problem][n][entry] (sc_mvar.hip beyond 64 signals:
// the products around this kernel read and write whole rows of a (window, bin) matrix instead of 16 bytes per 4-KB stride).  The
// NF series of a workgroup are NF consecutive entries: lane = entry, so a load is NF x 16 contiguous bytes per n; the LDS row of a
// series is one element longer (rows of lanes that differ in the series must fall into different banks).
template <int LOG2N, bool NAT>
__global__ void __launch_bounds__(256, 2) causal_fft_pair_kernel(cd* A, const int32_t* status, int64_t n_series, int C) {
    constexpr int N = 1 << LOG2N, TPF = N / 16, NF = 256 / TPF, ZS = N + N / 16 + (NAT ? 1 : 0), NHI = N / 64;
    extern __shared__ __align__(16) unsigned char wf_smem[];
    cd* z = reinterpret_cast<cd*>(wf_smem);
    cd* lo = z + NF * ZS;
    cd* hi = lo + 64;
    const int tid = threadIdx.x, pr = NAT ? tid % NF : tid / TPF, i = NAT ? tid / NF : tid % TPF;
    const int64_t series = (int64_t)blockIdx.x * NF + pr;
    const int E = C * C;
    bool valid = series < n_series;
    if (valid) valid = status[series / E] == 0;
    if (!__syncthreads_or(valid ? 1 : 0)) return;
    if (tid < 64 + NHI) {
        const int m = tid < 64 ? tid : (tid - 64) * 64;
        double s, c;
        sincospi(-2.0 * (double)m / (double)N, &s, &c);
        (tid < 64 ? lo[tid] : hi[tid - 64]) = make_double2(c, s);
    }
    cd* zf = z + pr * ZS;
    const int e = (int)(series % E);
    cd* Ap = NAT ? A + (series / E) * (int64_t)N * E + e : A + series * N;
    const int64_t sn = NAT ? E : 1;
    cd a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        cd v = make_double2(0.0, 0.0);
        if (valid) v = Ap[(i + t * TPF) * sn];
        a[t] = make_double2(v.x, -v.y);
    }
    wf_fft<LOG2N>(a, zf, lo, hi, i);                 // a = conj(N ifft(A))
    const bool lower = (e / C) > (e % C);
    const double invN = 1.0 / (double)N;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int n = i + t * TPF;
        double sc = (n < N / 2) ? invN : 0.0;
        if (n == 0) sc = lower ? 0.0 : 0.5 * invN;
        a[t] = make_double2(a[t].x * sc, -a[t].y * sc);
    }
    wf_fft<LOG2N>(a, zf, lo, hi, i);
    if (valid) {
#pragma unroll
        for (int t = 0; t < 16; ++t) Ap[(i + t * TPF) * sn] = a[t];
    }
}

template <int LOG2N, bool NAT>
static int wf_launch(void* d_A, const int32_t* d_status, int64_t n_series, int C, hipStream_t st) {
    constexpr int N = 1 << LOG2N, NF = 256 / (N / 16), ZS = N + N / 16 + (NAT ? 1 : 0);
    const size_t lds = ((size_t)NF * ZS + 64 + N / 64) * sizeof(cd);
    // (per call: the attribute belongs to the function ON THE CURRENT DEVICE, a process-wide "done" flag would skip the
    //  second device of a multi-GPU process)
    SC_CHECK_HIP(hipFuncSetAttribute((const void*)causal_fft_pair_kernel<LOG2N, NAT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (n_series + NF - 1) / NF;
    SC_REQUIRE(blocks <= 0x7fffffffLL, "too many series for one launch");
    hipLaunchKernelGGL((causal_fft_pair_kernel<LOG2N, NAT>), dim3((unsigned)blocks), dim3(256), lds, st, (cd*)d_A, d_status,
                       n_series, C);
    SC_CHECK_HIP(hipGetLastError());
    return SC_OK;
}

// Lengths the fused kernel takes; SC_WILSON_FFT=rocfft in the environment sends every length to the library path
// (ablation and the cross-check in tests/).
bool sc_internal_causal_fft_supported(int64_t N) {
    const char* e = sc_switch(SC_SW_WILSON_FFT);
    if (e && strcmp(e, "rocfft") == 0) return false;
    return N == 256 || N == 512 || N == 1024 || N == 2048 || N == 4096;
}

// d_A [n_problems][C*C][N] complex128, in place: A <- fft(causal(ifft(A))) for problems with status == 0.
int sc_internal_causal_fft_pair(void* d_A, const int32_t* d_status, int64_t n_problems, int C, int64_t N,
                                hipStream_t st) {
    const int64_t n_series = n_problems * C * C;
    switch (N) {
        case 256: return wf_launch<8, false>(d_A, d_status, n_series, C, st);
        case 512: return wf_launch<9, false>(d_A, d_status, n_series, C, st);
        case 1024: return wf_launch<10, false>(d_A, d_status, n_series, C, st);
        case 2048: return wf_launch<11, false>(d_A, d_status, n_series, C, st);
        case 4096: return wf_launch<12, false>(d_A, d_status, n_series, C, st);
        default: break;
    }
    sc_set_error("causal FFT pair: unsupported length %lld", (long long)N);
    return SC_EINVAL;
}

// The same on d_A [n_problems][N][C*C] (the matrix of a bin contiguous): lengths whose workgroup holds >= 8 series, i.e. loads
// of >= 128 contiguous bytes.
bool sc_internal_causal_fft_natural_supported(int64_t N) {
    return sc_internal_causal_fft_supported(N) && (N == 256 || N == 512) && !sc_switch(SC_SW_WILSON_FFT);
}
int sc_internal_causal_fft_pair_natural(void* d_A, const int32_t* d_status, int64_t n_problems, int C, int64_t N,
                                        hipStream_t st) {
    const int64_t n_series = n_problems * C * C;
    if (N == 256) return wf_launch<8, true>(d_A, d_status, n_series, C, st);
    if (N == 512) return wf_launch<9, true>(d_A, d_status, n_series, C, st);
    sc_set_error("causal FFT pair, natural layout: unsupported length %lld", (long long)N);
    return SC_EINVAL;
}
