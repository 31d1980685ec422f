// a11: STFT magnitude in dB (reference analyse/spectrogram.py:107-160 and its copies in waterfall.py /
// modalcloud.py).  One workgroup handles TB consecutive output columns (frames) of one segment:
//   frame*window -> packed real FFT (n_fft/2-point complex DIF in LDS) -> |X| -> floor -> 20 log10 -> f32,
// collecting the TB columns in an LDS tile so that each output row is written as a TB-float run of the
// C-contiguous (F, T) matrix the reference returns.
#include <cmath>
#include <cstdlib>

#include "ira_fft_lds.h"

namespace {

using ira::cplx;

constexpr int STFT_THREADS = 256;

template <typename T>
__device__ __forceinline__ float mag_to_db(T re, T im, T floor_lin);

template <>
__device__ __forceinline__ float mag_to_db<float>(float re, float im, float floor_lin) {
  const float a = sqrtf(re * re + im * im);
  const float m = (a != a) ? a : fmaxf(a, floor_lin);            // numpy.maximum keeps NaN (spectrogram.py:153-156)
  return 20.0f * log10f(m);
}
template <>
__device__ __forceinline__ float mag_to_db<double>(double re, double im, double floor_lin) {
  const double a = hypot(re, im);
  const double m = (a != a) ? a : fmax(a, floor_lin);            // numpy.maximum keeps NaN (spectrogram.py:153-156)
  return (float)(20.0 * log10(m));
}

template <typename T>
__global__ __launch_bounds__(STFT_THREADS) void stft_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int32_t* __restrict__ nframes, int log2n,
    int hop, const T* __restrict__ window, const cplx<T>* __restrict__ tw, T floor_lin, float* __restrict__ out,
    const int64_t* __restrict__ out_off, const int32_t* __restrict__ frame_sel, const int64_t* __restrict__ sel_off,
    int tb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int seg = blockIdx.y;
  const int T_out = nframes[seg];
  const int col0 = blockIdx.x * tb;
  if (col0 >= T_out) return;
  const int ncol = (T_out - col0 < tb) ? T_out - col0 : tb;

  const int N = 1 << log2n, M = N >> 1, F = M + 1;
  const int log2m = log2n - 1;
  cplx<T>* buf = reinterpret_cast<cplx<T>*>(smem_raw);
  float* tile = reinterpret_cast<float*>(smem_raw + sizeof(cplx<T>) * (size_t)M);  // [F][tb]
  const int tid = threadIdx.x;
  const float* xs = x + off[seg];

  for (int c = 0; c < ncol; ++c) {
    const int col = col0 + c;
    const int64_t frame = frame_sel ? (int64_t)frame_sel[sel_off[seg] + col] : (int64_t)col;
    const float* fx = xs + frame * hop;
    // windowed frame packed as z[m] = xw[2m] + i xw[2m+1]
    for (int m = tid; m < M; m += STFT_THREADS) {
      const T a = (T)fx[2 * m] * window[2 * m];
      const T b = (T)fx[2 * m + 1] * window[2 * m + 1];
      buf[m] = {a, b};
    }
    __syncthreads();
    ira::lds_fft_dif<T>(buf, log2m, tw, 2u, tid, STFT_THREADS);
    // real-FFT split: X[k] = E + W_N^k O,  E = (Z[k] + conj(Z[M-k]))/2,  O = -i (Z[k] - conj(Z[M-k]))/2
    for (int k = tid; k <= M; k += STFT_THREADS) {
      T re, im;
      if (k == 0 || k == M) {
        const cplx<T> z0 = buf[0];
        re = (k == 0) ? z0.re + z0.im : z0.re - z0.im;
        im = (T)0;
      } else {
        const cplx<T> zk = buf[ira::lds_brev((unsigned)k, log2m)];
        const cplx<T> zm = buf[ira::lds_brev((unsigned)(M - k), log2m)];
        const cplx<T> e = {(T)0.5 * (zk.re + zm.re), (T)0.5 * (zk.im - zm.im)};
        const cplx<T> d = {(T)0.5 * (zk.re - zm.re), (T)0.5 * (zk.im + zm.im)};  // (Z[k]-conj(Z[M-k]))/2
        const cplx<T> o = {d.im, -d.re};                                         // times -i
        const cplx<T> wo = ira::cmul(tw[k], o);
        re = e.re + wo.re;
        im = e.im + wo.im;
      }
      tile[k * tb + c] = mag_to_db<T>(re, im, floor_lin);
    }
    __syncthreads();
  }
  // write the tile: row k gets ncol consecutive floats at out[k*T_out + col0 ...]
  float* o = out + out_off[seg];
  const int total = F * ncol;
  for (int i = tid; i < total; i += STFT_THREADS) {
    const int k = i / ncol, c = i - k * ncol;
    o[(int64_t)k * T_out + col0 + c] = tile[k * tb + c];
  }
}

template <typename T>
int32_t launch_stft(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg, int32_t max_frames,
                    int log2n, int32_t hop, const void* window, const void* tw, double floor_db, float* out,
                    const int64_t* out_off, const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st) {
  const int N = 1 << log2n, M = N / 2, F = M + 1;
  const size_t fft_bytes = sizeof(cplx<T>) * (size_t)M;
  // columns per workgroup: as many as fit beside the FFT buffer in 64 KB of LDS, at most 8
  int tb = (int)((65536 - fft_bytes) / (sizeof(float) * (size_t)F));
  if (tb > 8) tb = 8;
  size_t lds = fft_bytes + sizeof(float) * (size_t)F * (size_t)(tb < 1 ? 1 : tb);
  if (tb < 1) {
    tb = 1;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return ira_hip_status(e);
  }
  if (lds > 160 * 1024) return IRA_E_SIZE;
  const double floor_lin = std::pow(10.0, floor_db / 20.0);
  dim3 grid((max_frames + tb - 1) / tb, nseg);
  stft_kernel<T><<<grid, STFT_THREADS, lds, st>>>(x, off, nframes, log2n, hop, static_cast<const T*>(window),
                                                  static_cast<const cplx<T>*>(tw), (T)floor_lin, out, out_off,
                                                  frame_sel, sel_off, tb);
  IRA_RETURN_LAUNCH();
}

}  // namespace

// register-resident configurations (ira_stft2.hip); IRA_E_UNSUPPORTED = use the generic kernel above
int32_t ira_stft2_dispatch(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                           int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                           int32_t precision, double floor_db, float* out, const int64_t* out_off,
                           const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st);

// float32 / n_fft 4096 at 16 one-wave teams per CU (ira_stft3.hip)
int32_t ira_stft3_dispatch(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                           int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                           int32_t precision, double floor_db, float* out, const int64_t* out_off,
                           const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st);

extern "C" int32_t ira_stft_mag_db(const float* x_dev, const int64_t* off_dev, const int32_t* nframes_dev,
                                   int32_t nseg, int32_t max_frames, int32_t n_fft, int32_t hop,
                                   const void* window_dev, const void* twiddle_dev, int32_t precision,
                                   double floor_db, float* out_dev, const int64_t* out_off_dev,
                                   const int32_t* frame_sel_dev, const int64_t* sel_off_dev, void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(nframes_dev); IRA_CHECK_PTR(window_dev);
  IRA_CHECK_PTR(twiddle_dev); IRA_CHECK_PTR(out_dev); IRA_CHECK_PTR(out_off_dev);
  if (frame_sel_dev != nullptr && sel_off_dev == nullptr) return IRA_E_NULL;
  if (nseg < 0 || max_frames < 0 || hop <= 0) return IRA_E_SIZE;
  if (nseg == 0 || max_frames == 0) return IRA_OK;
  if (n_fft < 64 || n_fft > 16384 || (n_fft & (n_fft - 1)) != 0) return IRA_E_SIZE;
  int log2n = 0;
  while ((1 << log2n) < n_fft) ++log2n;
  hipStream_t st = (hipStream_t)stream;
  if (precision != 32 && precision != 64) return IRA_E_UNSUPPORTED;
  const bool force_generic = ira_tune_flag("IRA_STFT_GENERIC");   // A/B switch for benchmarking
  const bool no_v3 = ira_tune_flag("IRA_STFT_NO_V3");             // A/B switch for benchmarking
  if (!force_generic && !no_v3) {
    const int32_t rc = ira_stft3_dispatch(x_dev, off_dev, nframes_dev, nseg, max_frames, n_fft, hop, window_dev,
                                          twiddle_dev, precision, floor_db, out_dev, out_off_dev, frame_sel_dev,
                                          sel_off_dev, st);
    if (rc != IRA_E_UNSUPPORTED) return rc;
  }
  if (!force_generic) {
    const int32_t rc = ira_stft2_dispatch(x_dev, off_dev, nframes_dev, nseg, max_frames, n_fft, hop, window_dev,
                                          twiddle_dev, precision, floor_db, out_dev, out_off_dev, frame_sel_dev,
                                          sel_off_dev, st);
    if (rc != IRA_E_UNSUPPORTED) return rc;
  }
  if (precision == 32)
    return launch_stft<float>(x_dev, off_dev, nframes_dev, nseg, max_frames, log2n, hop, window_dev, twiddle_dev,
                              floor_db, out_dev, out_off_dev, frame_sel_dev, sel_off_dev, st);
  if (precision == 64)
    return launch_stft<double>(x_dev, off_dev, nframes_dev, nseg, max_frames, log2n, hop, window_dev, twiddle_dev,
                               floor_db, out_dev, out_off_dev, frame_sel_dev, sel_off_dev, st);
  return IRA_E_UNSUPPORTED;
}

int32_t ira_stft3_dispatch_tf(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                              int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                              int32_t precision, double floor_db, float* out, const int64_t* out_off,
                              const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st);

int32_t ira_stft4_dispatch_tf(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                              int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                              int32_t precision, double floor_db, float* out, const int64_t* out_off,
                              const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st);

extern "C" int32_t ira_stft_mag_db_tf(const float* x_dev, const int64_t* off_dev, const int32_t* nframes_dev,
                                      int32_t nseg, int32_t max_frames, int32_t n_fft, int32_t hop,
                                      const void* window_dev, const void* twiddle_dev, int32_t precision,
                                      double floor_db, float* out_dev, const int64_t* out_off_dev,
                                      const int32_t* frame_sel_dev, const int64_t* sel_off_dev, void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(nframes_dev); IRA_CHECK_PTR(window_dev);
  IRA_CHECK_PTR(twiddle_dev); IRA_CHECK_PTR(out_dev); IRA_CHECK_PTR(out_off_dev);
  if (frame_sel_dev != nullptr && sel_off_dev == nullptr) return IRA_E_NULL;
  if (nseg < 0 || max_frames < 0 || hop <= 0) return IRA_E_SIZE;
  if (nseg == 0 || max_frames == 0) return IRA_OK;
  if (nseg > 65535) return IRA_E_SIZE;
  const int32_t rc = ira_stft3_dispatch_tf(x_dev, off_dev, nframes_dev, nseg, max_frames, n_fft, hop, window_dev,
                                           twiddle_dev, precision, floor_db, out_dev, out_off_dev, frame_sel_dev,
                                           sel_off_dev, (hipStream_t)stream);
  if (rc != IRA_E_UNSUPPORTED) return rc;
  return ira_stft4_dispatch_tf(x_dev, off_dev, nframes_dev, nseg, max_frames, n_fft, hop, window_dev, twiddle_dev,
                               precision, floor_db, out_dev, out_off_dev, frame_sel_dev, sel_off_dev,
                               (hipStream_t)stream);
}

int32_t ira_stft4_dispatch_logbin(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                                  int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                                  int32_t precision, double floor_db, int32_t k_base, const int32_t* first,
                                  const int32_t* count, int32_t nbins, float* curves, const int64_t* curves_off,
                                  hipStream_t st);

extern "C" int32_t ira_stft_logbin(const float* x_dev, const int64_t* off_dev, const int32_t* nframes_dev,
                                   int32_t nseg, int32_t max_frames, int32_t n_fft, int32_t hop,
                                   const void* window_dev, const void* twiddle_dev, int32_t precision, double floor_db,
                                   int32_t k_base, const int32_t* first_dev, const int32_t* count_dev, int32_t nbins,
                                   float* curves_dev, const int64_t* curves_off_dev, void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(nframes_dev); IRA_CHECK_PTR(window_dev);
  IRA_CHECK_PTR(twiddle_dev); IRA_CHECK_PTR(first_dev); IRA_CHECK_PTR(count_dev); IRA_CHECK_PTR(curves_dev);
  IRA_CHECK_PTR(curves_off_dev);
  if (nseg < 0 || max_frames < 0 || hop <= 0 || nbins <= 0 || k_base < 0) return IRA_E_SIZE;
  if (nseg == 0 || max_frames == 0) return IRA_OK;
  if (nseg > 65535) return IRA_E_SIZE;
  return ira_stft4_dispatch_logbin(x_dev, off_dev, nframes_dev, nseg, max_frames, n_fft, hop, window_dev, twiddle_dev,
                                   precision, floor_db, k_base, first_dev, count_dev, nbins, curves_dev,
                                   curves_off_dev, (hipStream_t)stream);
}
