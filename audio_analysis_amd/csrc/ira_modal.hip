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
  // numpy.max and numpy.clip keep NaN (waterfall.py:318-341): a NaN anywhere in the selection makes the global reference
  // NaN and with it every relative value; in slice_max mode only the slices that hold one.
  const float qnan = __uint_as_float(0x7fc00000u);
  float ref_global = -INFINITY;
  if (!slice_max) {
    float mx = -INFINITY;
    int bad = 0;
    for (int i = tid; i < nsel * S; i += WF_THREADS) {
      const int k = i / S, s = i - k * S;
      const float v = m[(int64_t)(k_lo + k) * S + s];
      bad |= (v != v);
      mx = fmaxf(mx, v);
    }
    mx = ira::wave_max(mx);
    bad = __any(bad) ? 1 : 0;
    if ((tid & 63) == 0) wred[tid >> 6] = bad ? qnan : mx;
    __syncthreads();
    for (int w = 0; w < WF_THREADS / IRA_WAVE; ++w) {
      const float v = wred[w];
      ref_global = (v != v || ref_global != ref_global) ? qnan : fmaxf(ref_global, v);
    }
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
      for (int k = tid / ns; k < nsel; k += WF_THREADS / ns > 0 ? WF_THREADS / ns : 1) {
        const float v = m[(int64_t)(k_lo + k) * S + s0 + s];
        mx = (v != v || mx != mx) ? qnan : fmaxf(mx, v);
      }
      for (int turn = 0; turn < WF_THREADS; turn += ns) {
        if (tid >= turn && tid < turn + ns) smax[s] = (mx != mx || smax[s] != smax[s]) ? qnan : fmaxf(smax[s], mx);
        __syncthreads();
      }
      for (int i = tid; i < nsel * ns; i += WF_THREADS) {
        const int ss = i / nsel, k = i - ss * nsel;
        float rel = m[(int64_t)(k_lo + k) * S + s0 + ss] - smax[ss];
        if (rel == rel) rel = fminf(fmaxf(rel, -dyn), 0.0f);
        o[(int64_t)(s0 + ss) * nsel + k] = rel;
      }
    }
    return;
  }
  for (int i = tid; i < nsel * S; i += WF_THREADS) {
    const int s = i / nsel, k = i - s * nsel;
    float rel = m[(int64_t)(k_lo + k) * S + s] - ref_global;
    if (rel == rel) rel = fminf(fmaxf(rel, -dyn), 0.0f);
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

// Same aggregation on a FRAME-MAJOR (T, F) matrix (ira_stft_mag_db_tf).  One workgroup per frame: the frame's rows
// (contiguous) are converted to linear magnitude in parallel into LDS, then thread b adds the rows of log bin b in
// ascending order (the reference's numpy axis-0 mean adds rows in that order).
constexpr int LBT_THREADS = 256;
constexpr int LBT_MAX_ROWS = 8193;            // n_fft <= 16384

__global__ __launch_bounds__(LBT_THREADS) void logbin_tf_kernel(
    const float* __restrict__ mag, const int64_t* __restrict__ mag_off, const int32_t* __restrict__ nfr, int nrows,
    int k_base, int k_span, const int32_t* __restrict__ first, const int32_t* __restrict__ count, int nbins,
    float* __restrict__ out, const int64_t* __restrict__ out_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* lin = reinterpret_cast<double*>(smem_raw);        // k_span doubles: rows k_base .. k_base + k_span - 1
  const int e = blockIdx.y;
  const int T = nfr[e];
  const int t = blockIdx.x;
  if (t >= T) return;
  const float* m = mag + mag_off[e] + (int64_t)t * nrows + k_base;
  for (int k = threadIdx.x; k < k_span; k += LBT_THREADS) lin[k] = exp10((double)m[k] * 0.05);
  __syncthreads();
  float* o = out + out_off[e];
  for (int b = threadIdx.x; b < nbins; b += LBT_THREADS) {
    const int c = count[b];
    float v = __uint_as_float(0x7fc00000u);
    if (c > 0) {
      const double* r = lin + first[b];
      double acc = r[0];
      // eight LDS reads in flight, then the adds in row order (one read per add would serialise on the LDS latency)
      for (int k0 = 1; k0 < c; k0 += 8) {
        double v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = (k0 + u < c) ? r[k0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < c) acc += v8[u];
      }
      const double mean = fmax(acc / (double)c, 1e-30);
      v = (float)(20.0 * log10(mean));
    }
    o[(int64_t)b * T + t] = v;
  }
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
                                        const int64_t* out_off_dev, int32_t frame_major_rows, void* stream) {
  IRA_CHECK_PTR(mag_dev); IRA_CHECK_PTR(mag_off_dev); IRA_CHECK_PTR(nframes_dev); IRA_CHECK_PTR(first_dev);
  IRA_CHECK_PTR(count_dev); IRA_CHECK_PTR(out_dev); IRA_CHECK_PTR(out_off_dev);
  if (nb <= 0 || nbins <= 0 || max_frames <= 0) return (nb == 0 || nbins == 0 || max_frames == 0) ? IRA_OK : IRA_E_SIZE;
  if (nbins > 65535 || nb > 65535) return IRA_E_SIZE;
  if (frame_major_rows > 0) {
    if (frame_major_rows > LBT_MAX_ROWS || k_base < 0 || k_base >= frame_major_rows) return IRA_E_SIZE;
    // rows the bins can touch: first/count live on the device, so take everything from k_base to the last row
    const int k_span = frame_major_rows - k_base;
    const size_t lds = sizeof(double) * (size_t)k_span;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&logbin_tf_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return ira_hip_status(e);
    }
    logbin_tf_kernel<<<dim3(max_frames, nb), LBT_THREADS, lds, (hipStream_t)stream>>>(
        mag_dev, mag_off_dev, nframes_dev, frame_major_rows, k_base, k_span, first_dev, count_dev, nbins, out_dev,
        out_off_dev);
    IRA_RETURN_LAUNCH();
  }
  const int threads = 64;
  logbin_kernel<<<dim3((max_frames + threads - 1) / threads, nbins, nb), threads, 0, (hipStream_t)stream>>>(
      mag_dev, mag_off_dev, nframes_dev, k_base, first_dev, count_dev, nbins, out_dev, out_off_dev);
  IRA_RETURN_LAUNCH();
}
