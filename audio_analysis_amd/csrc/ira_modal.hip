// Waterfall slice normalisation and modal-cloud log-frequency aggregation.
// Compiled with -ffp-contract=off (sequential float64 sums mirror NumPy's axis-0 reduction order).
#include <cmath>

#include "ira_common.h"

namespace {

// ---- a14: waterfall.  in: STFT dB matrix (F, S) of the S selected frames; out: (S, nsel) float32 =
//      clip(slice - ref, -dyn, 0) over bins k_lo .. k_lo+nsel-1, ref = global max or per-slice max.
//      Reference analyse/waterfall.py:308-341 (float32 subtraction and clip).
constexpr int WF_THREADS = 256;

__global__ __launch_bounds__(WF_THREADS) void waterfall_kernel(const float* __restrict__ mag,
                                                               const int64_t* __restrict__ mag_off,
                                                               const int32_t* __restrict__ nsl, int k_lo, int nsel,
                                                               int slice_max, float dyn, float* __restrict__ out,
                                                               const int64_t* __restrict__ out_off) {
  __shared__ float wred[WF_THREADS / IRA_WAVE];
  __shared__ float smax[64];
  const int e = blockIdx.x;
  const int S = nsl[e];
  const float* m = mag + mag_off[e];
  float* o = out + out_off[e];
  const int tid = threadIdx.x;
  float ref_global = -INFINITY;
  if (!slice_max) {
    float mx = -INFINITY;
    for (int i = tid; i < nsel * S; i += WF_THREADS) {
      const int k = i / S, s = i - k * S;
      mx = fmaxf(mx, m[(int64_t)(k_lo + k) * S + s]);
    }
    mx = ira::wave_max(mx);
    if ((tid & 63) == 0) wred[tid >> 6] = mx;
    __syncthreads();
    for (int w = 0; w < WF_THREADS / IRA_WAVE; ++w) ref_global = fmaxf(ref_global, wred[w]);
  } else {
    // per-slice maxima, 64 slices at a time
    for (int s0 = 0; s0 < S; s0 += 64) {
      __syncthreads();
      if (tid < 64) smax[tid] = -INFINITY;
      __syncthreads();
      const int ns = (S - s0 < 64) ? S - s0 : 64;
      // thread handles slice (tid % ns) over a strided set of bins, then a serialised shared max
      const int s = tid % ns;
      float mx = -INFINITY;
      for (int k = tid / ns; k < nsel; k += WF_THREADS / ns > 0 ? WF_THREADS / ns : 1)
        mx = fmaxf(mx, m[(int64_t)(k_lo + k) * S + s0 + s]);
      for (int turn = 0; turn < WF_THREADS; turn += ns) {
        if (tid >= turn && tid < turn + ns) smax[s] = fmaxf(smax[s], mx);
        __syncthreads();
      }
      for (int i = tid; i < nsel * ns; i += WF_THREADS) {
        const int ss = i / nsel, k = i - ss * nsel;
        float rel = m[(int64_t)(k_lo + k) * S + s0 + ss] - smax[ss];
        rel = fminf(fmaxf(rel, -dyn), 0.0f);
        o[(int64_t)(s0 + ss) * nsel + k] = rel;
      }
    }
    return;
  }
  for (int i = tid; i < nsel * S; i += WF_THREADS) {
    const int s = i / nsel, k = i - s * nsel;
    float rel = m[(int64_t)(k_lo + k) * S + s] - ref_global;
    rel = fminf(fmaxf(rel, -dyn), 0.0f);
    o[(int64_t)s * nsel + k] = rel;
  }
}

// ---- a15: modal cloud log-bin aggregation.  in: STFT dB matrix (F, T); for log bin b the rows
//      k_base + first[b] .. + count[b] - 1 are averaged in LINEAR magnitude (10^(dB/20), float64, rows added in
//      order like numpy's axis-0 mean), then 20 log10(max(., 1e-30)) -> float32; empty bin -> NaN.
//      Reference analyse/modalcloud.py:176-207.
__global__ void logbin_kernel(const float* __restrict__ mag, const int64_t* __restrict__ mag_off,
                              const int32_t* __restrict__ nfr, int k_base, const int32_t* __restrict__ first,
                              const int32_t* __restrict__ count, int nbins, float* __restrict__ out,
                              const int64_t* __restrict__ out_off) {
  const int e = blockIdx.z;
  const int T = nfr[e];
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float* m = mag + mag_off[e];
  float* o = out + out_off[e] + (int64_t)b * T;
  const int c = count[b];
  if (c <= 0) {
    o[t] = __uint_as_float(0x7fc00000u);
    return;
  }
  const int k0 = k_base + first[b];
  double acc = 0.0;
  for (int k = 0; k < c; ++k) {
    const double db = (double)m[(int64_t)(k0 + k) * T + t];
    const double lin = exp10(db * 0.05);           // 10^(dB/20); pow() costs 4.5x the instructions for the same f32 result
    acc = (k == 0) ? lin : acc + lin;
  }
  double mean = acc / (double)c;
  mean = fmax(mean, 1e-30);
  o[t] = (float)(20.0 * log10(mean));
}

}  // namespace

extern "C" int32_t ira_waterfall_rel(const float* mag_dev, const int64_t* mag_off_dev, const int32_t* nslices_dev,
                                     int32_t nb, int32_t k_lo, int32_t nsel, int32_t slice_max, double dyn_db,
                                     float* out_dev, const int64_t* out_off_dev, void* stream) {
  IRA_CHECK_PTR(mag_dev); IRA_CHECK_PTR(mag_off_dev); IRA_CHECK_PTR(nslices_dev); IRA_CHECK_PTR(out_dev);
  IRA_CHECK_PTR(out_off_dev);
  if (nb <= 0) return nb == 0 ? IRA_OK : IRA_E_SIZE;
  if (k_lo < 0 || nsel <= 0) return IRA_E_SIZE;
  waterfall_kernel<<<nb, WF_THREADS, 0, (hipStream_t)stream>>>(mag_dev, mag_off_dev, nslices_dev, k_lo, nsel,
                                                               slice_max, (float)dyn_db, out_dev, out_off_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_logbin_aggregate(const float* mag_dev, const int64_t* mag_off_dev, const int32_t* nframes_dev,
                                        int32_t nb, int32_t max_frames, int32_t k_base, const int32_t* first_dev,
                                        const int32_t* count_dev, int32_t nbins, float* out_dev,
                                        const int64_t* out_off_dev, void* stream) {
  IRA_CHECK_PTR(mag_dev); IRA_CHECK_PTR(mag_off_dev); IRA_CHECK_PTR(nframes_dev); IRA_CHECK_PTR(first_dev);
  IRA_CHECK_PTR(count_dev); IRA_CHECK_PTR(out_dev); IRA_CHECK_PTR(out_off_dev);
  if (nb <= 0 || nbins <= 0 || max_frames <= 0) return (nb == 0 || nbins == 0 || max_frames == 0) ? IRA_OK : IRA_E_SIZE;
  if (nbins > 65535 || nb > 65535) return IRA_E_SIZE;
  const int threads = 64;
  logbin_kernel<<<dim3((max_frames + threads - 1) / threads, nbins, nb), threads, 0, (hipStream_t)stream>>>(
      mag_dev, mag_off_dev, nframes_dev, k_base, first_dev, count_dev, nbins, out_dev, out_off_dev);
  IRA_RETURN_LAUNCH();
}
