// sweep deconvolution (reference analyse/deconvolve.py:124-193).  The transforms are the
// library's long FFTs (ira_rfft_any / ira_rfft_smooth zero-padded to the power-of-two size, ira_band_irfft* with an
// all-pass mask); this file holds what sits between and after them:
//   ira_deconv_divide   H = Y conj(X) / (|X|^2 + eps), eps = reg * max(1e-30, max_k |X_k|^2)     (deconvolve.py:150-166)
//   ira_deconv_finish   per channel: subtract the mean (float32), then per FILE: one peak normalisation over all of its
//                       channels (deconvolve.py:176-189, :100-106)
// Compiled with -ffp-contract=off: the float32 epilogue (h - mean, h * scale) rounds like NumPy's.
#include <cmath>

#include "ira_common.h"

namespace {

typedef ira::cplx<double> cd;
constexpr int DV_THREADS = 256;
constexpr int FIN_THREADS = 1024;

// |X|^2 the way the reference forms it: numpy.abs (hypot) squared
__device__ __forceinline__ double power_of(cd x) {
  const double a = hypot(x.re, x.im);
  return a * a;
}

__global__ __launch_bounds__(DV_THREADS) void deconv_pmax_kernel(const cd* __restrict__ xspec,
                                                                 const int64_t* __restrict__ xoff,
                                                                 const int32_t* __restrict__ nfft,
                                                                 unsigned long long* __restrict__ pmax_bits) {
  __shared__ double part[DV_THREADS / IRA_WAVE];
  const int e = blockIdx.y;
  const long long nbins = (long long)nfft[e] / 2 + 1;
  const cd* x = xspec + xoff[e];
  double m = 0.0;
  for (long long k = (long long)blockIdx.x * DV_THREADS + threadIdx.x; k < nbins; k += (long long)gridDim.x * DV_THREADS)
    m = fmax(m, power_of(x[k]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < DV_THREADS / IRA_WAVE; ++w) m = fmax(m, part[w]);
    atomicMax(pmax_bits + e, (unsigned long long)__double_as_longlong(m));      // non-negative doubles order like their bits
  }
}

__global__ __launch_bounds__(DV_THREADS) void deconv_divide_kernel(cd* __restrict__ yspec,
                                                                   const int64_t* __restrict__ yoff,
                                                                   const cd* __restrict__ xspec,
                                                                   const int64_t* __restrict__ xoff,
                                                                   const int32_t* __restrict__ nfft, double reg,
                                                                   const double* __restrict__ pmax) {
  const int e = blockIdx.y;
  const long long nbins = (long long)nfft[e] / 2 + 1;
  const double eps = reg * fmax(1e-30, pmax[e]);
  cd* y = yspec + yoff[e];
  const cd* x = xspec + xoff[e];
  for (long long k = (long long)blockIdx.x * DV_THREADS + threadIdx.x; k < nbins; k += (long long)gridDim.x * DV_THREADS) {
    const cd xv = x[k], yv = y[k];
    const double denom = power_of(xv) + eps;
    const cd p = ira::cmul(yv, ira::cconj(xv));
    y[k] = {p.re / denom, p.im / denom};
  }
}

// One workgroup per channel: float64 sum in a fixed order (thread-strided partials, then a fixed tree) -> deterministic.
__global__ __launch_bounds__(FIN_THREADS) void deconv_mean_kernel(const float* __restrict__ h,
                                                                  const int64_t* __restrict__ hoff,
                                                                  const int32_t* __restrict__ n_out,
                                                                  float* __restrict__ mean_out) {
  __shared__ double part[FIN_THREADS / IRA_WAVE];
  const int e = blockIdx.x;
  const int n = n_out[e];
  const float* p = h + hoff[e];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += FIN_THREADS) s += (double)p[i];
  s = ira::wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < FIN_THREADS / IRA_WAVE; ++w) t += part[w];
    mean_out[e] = n > 0 ? (float)(t / (double)n) : 0.0f;
  }
}

// h -= mean (optional), and the file's peak |h| (max is order independent)
__global__ __launch_bounds__(DV_THREADS) void deconv_center_kernel(float* __restrict__ h, const int64_t* __restrict__ hoff,
                                                                   const int32_t* __restrict__ n_out,
                                                                   const int32_t* __restrict__ group,
                                                                   const float* __restrict__ mean,
                                                                   unsigned* __restrict__ peak_bits) {
  __shared__ float part[DV_THREADS / IRA_WAVE];
  const int e = blockIdx.y;
  const int n = n_out[e];
  float* p = h + hoff[e];
  const float mu = mean ? mean[e] : 0.0f;
  float m = 0.0f;
  for (long long i = (long long)blockIdx.x * DV_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * DV_THREADS) {
    float v = p[i];
    if (mean) { v = v - mu; p[i] = v; }
    m = fmaxf(m, fabsf(v));
  }
  m = ira::wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < DV_THREADS / IRA_WAVE; ++w) m = fmaxf(m, part[w]);
    atomicMax(peak_bits + group[e], __float_as_uint(m));
  }
}

__global__ __launch_bounds__(DV_THREADS) void deconv_scale_kernel(float* __restrict__ h, const int64_t* __restrict__ hoff,
                                                                  const int32_t* __restrict__ n_out,
                                                                  const int32_t* __restrict__ group, double target,
                                                                  const unsigned* __restrict__ peak_bits) {
  const int e = blockIdx.y;
  const int n = n_out[e];
  const double peak = (double)__uint_as_float(peak_bits[group[e]]);
  if (!(peak > 0.0)) return;                                   // all-zero response: left as it is (deconvolve.py:103-104)
  const float scale = (float)(target / peak);                  // Python float scalar * float32 array -> float32 multiply
  float* p = h + hoff[e];
  for (long long i = (long long)blockIdx.x * DV_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * DV_THREADS)
    p[i] = p[i] * scale;
}

unsigned blocks_for(long long n) {
  long long b = (n + DV_THREADS * 8 - 1) / (DV_THREADS * 8);
  return (unsigned)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" int32_t ira_deconv_divide(double* yspec_dev, const int64_t* yspec_off_dev, const double* xspec_dev,
                                     const int64_t* xspec_off_dev, const int32_t* nfft_dev, int32_t nb, int32_t max_nfft,
                                     double regularization_relative, double* pmax_dev, void* stream) {
  IRA_CHECK_PTR(yspec_dev); IRA_CHECK_PTR(yspec_off_dev); IRA_CHECK_PTR(xspec_dev); IRA_CHECK_PTR(xspec_off_dev);
  IRA_CHECK_PTR(nfft_dev); IRA_CHECK_PTR(pmax_dev);
  if (nb < 0 || max_nfft < 2) return IRA_E_SIZE;
  if (nb == 0) return IRA_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t err = hipMemsetAsync(pmax_dev, 0, sizeof(double) * (size_t)nb, st);
  if (err != hipSuccess) return ira_hip_status(err);
  const dim3 grid(blocks_for((long long)max_nfft / 2 + 1), (unsigned)nb);
  deconv_pmax_kernel<<<grid, DV_THREADS, 0, st>>>(reinterpret_cast<const cd*>(xspec_dev), xspec_off_dev, nfft_dev,
                                                  reinterpret_cast<unsigned long long*>(pmax_dev));
  deconv_divide_kernel<<<grid, DV_THREADS, 0, st>>>(reinterpret_cast<cd*>(yspec_dev), yspec_off_dev,
                                                    reinterpret_cast<const cd*>(xspec_dev), xspec_off_dev, nfft_dev,
                                                    regularization_relative, pmax_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_deconv_finish(float* h_dev, const int64_t* h_off_dev, const int32_t* n_out_dev,
                                     const int32_t* group_dev, int32_t nb, int32_t ngroups, int32_t max_len,
                                     int32_t remove_dc, int32_t normalise_peak, double target_peak, float* mean_dev,
                                     uint32_t* peak_bits_dev, void* stream) {
  IRA_CHECK_PTR(h_dev); IRA_CHECK_PTR(h_off_dev); IRA_CHECK_PTR(n_out_dev); IRA_CHECK_PTR(group_dev);
  IRA_CHECK_PTR(mean_dev); IRA_CHECK_PTR(peak_bits_dev);
  if (nb < 0 || ngroups < 0 || max_len < 0) return IRA_E_SIZE;
  if (nb == 0 || (!remove_dc && !normalise_peak)) return IRA_OK;
  hipStream_t st = (hipStream_t)stream;
  hipError_t err = hipMemsetAsync(peak_bits_dev, 0, sizeof(uint32_t) * (size_t)(ngroups > 0 ? ngroups : 1), st);
  if (err != hipSuccess) return ira_hip_status(err);
  if (remove_dc) deconv_mean_kernel<<<(unsigned)nb, FIN_THREADS, 0, st>>>(h_dev, h_off_dev, n_out_dev, mean_dev);
  const dim3 grid(blocks_for(max_len), (unsigned)nb);
  deconv_center_kernel<<<grid, DV_THREADS, 0, st>>>(h_dev, h_off_dev, n_out_dev, group_dev, remove_dc ? mean_dev : nullptr,
                                                    peak_bits_dev);
  if (normalise_peak)
    deconv_scale_kernel<<<grid, DV_THREADS, 0, st>>>(h_dev, h_off_dev, n_out_dev, group_dev, target_peak, peak_bits_dev);
  IRA_RETURN_LAUNCH();
}
