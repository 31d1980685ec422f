// diffusion / decorrelation metrics per time window (reference analyse/diffusion.py:139-226, :258-276,
// :346-358).  One workgroup per (window, channel):
//   mean removal        w0 = w - mean(w) in FLOAT32 with numpy's pairwise float32 summation reproduced exactly
//                       (np.mean of a float32 array: blocks <= 128 with eight running sums, halving above), because
//   echo density        fraction(|w0| > thr_rms * rms) is a COUNT: it only matches the reference if mean, w0, rms and the
//                       float32 threshold are bit-identical.  rms = sqrt(mean(w0*w0)) in float32, same summation.
//   max |autocorr|      r(lag) = sum_k w0[k] w0[k+lag] / sum w0^2 for lag = 1..min(max_lag, N-2): exact float32 products
//                       accumulated in float64 (the reference uses float32 BLAS dots, so this side is the more accurate
//                       one; they agree to ~1e-6).  A thread owns eight consecutive lags and a sub-range of k and slides
//                       an 8-value window over LDS: two LDS reads per eight FMAs.  The window is staged with a zero halo
//                       in front, which turns the lag-dependent summation range into a fixed one.
//   corr0 / IACC        the same machinery on two channels (both cross directions).
#include <cmath>

#include "ira_common.h"

namespace {

constexpr int DF_THREADS = 256;
constexpr int DF_MAX_WIN = 8192;
constexpr int DF_MAX_LAG = 4096;
constexpr int DF_MAX_LEAVES = DF_MAX_WIN / 64 + 2;

// numpy's float32 pairwise sum of n values (np.mean / np.sum of a float32 array):
//   n < 8:            res = 0; res += a[i] in order
//   n <= 128:         r[j] = a[j]; r[j] += a[i+j] for i = 8, 16, ..; res = ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)); tail in order
//   n > 128:          n2 = n/2 - (n/2) % 8;  sum(a, n2) + sum(a + n2, n - n2)
// The recursion depends on n alone, and n (the window length) is the same for every workgroup of a launch: the HOST unrolls it
// once per call into a PLAN -- the leaves (<= 128 values each, left to right) and the post-order list of additions
// sum[dst] += sum[src] that folds them -- and the plan travels as a kernel argument.  (Round 3 let thread 0 of every workgroup
// walk the recursion with explicit stacks, twice per window: dynamically indexed private arrays live in scratch memory, and
// those two serial walks plus the serial fold were most of a window's run time -- the autocorrelation itself is 15 k cycles.)
struct PwPlan {
  unsigned short start[DF_MAX_LEAVES];
  unsigned short len[DF_MAX_LEAVES];
  unsigned char dst[DF_MAX_LEAVES];
  unsigned char src[DF_MAX_LEAVES];
  int nleaves, nops;
};
static_assert(DF_MAX_LEAVES <= 256 && DF_MAX_WIN <= 65535, "plan fields");

static void pw_plan_build(PwPlan& P, int start, int n, int& first_leaf) {
  if (n <= 128) {
    first_leaf = P.nleaves;
    P.start[P.nleaves] = (unsigned short)start;
    P.len[P.nleaves] = (unsigned short)n;
    ++P.nleaves;
    return;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  int left = 0, right = 0;
  pw_plan_build(P, start, n2, left);
  pw_plan_build(P, start + n2, n - n2, right);
  P.dst[P.nops] = (unsigned char)left;                 // post-order: both subtrees are folded before this addition
  P.src[P.nops] = (unsigned char)right;
  ++P.nops;
  first_leaf = left;
}

static PwPlan pw_plan_of(int n) {
  PwPlan P{};
  int first = 0;
  if (n > 0) pw_plan_build(P, 0, n, first);
  return P;
}

struct PwShared {
  float sum[DF_MAX_LEAVES];
  float result;
};

// the float32 pairwise sum of a[0..n) held in LDS (all threads call; barriers inside)
__device__ float np_pairwise_sum_f32(const float* a, const PwPlan& plan, PwShared& pw) {
  const int tid = threadIdx.x;
  for (int j = tid; j < plan.nleaves; j += DF_THREADS) {
    const float* p = a + plan.start[j];
    const int l = plan.len[j];
    float res;
    if (l < 8) {
      res = 0.0f;
      for (int i = 0; i < l; ++i) res += p[i];
    } else {
      float r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3], r4 = p[4], r5 = p[5], r6 = p[6], r7 = p[7];
      int i = 8;
      for (; i < l - (l % 8); i += 8) {
        r0 += p[i]; r1 += p[i + 1]; r2 += p[i + 2]; r3 += p[i + 3];
        r4 += p[i + 4]; r5 += p[i + 5]; r6 += p[i + 6]; r7 += p[i + 7];
      }
      res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
      for (; i < l; ++i) res += p[i];
    }
    pw.sum[j] = res;
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < plan.nops; ++k) {
      const int d = plan.dst[k], sidx = plan.src[k];
      pw.sum[d] = pw.sum[d] + pw.sum[sidx];              // left + right, float32
    }
    pw.result = plan.nleaves > 0 ? pw.sum[0] : 0.0f;
  }
  __syncthreads();
  const float r = pw.result;
  __syncthreads();
  return r;
}

// sum_n cur[n] * lagsrc[n - b] for b = b0..b0+LG-1 over n in [r_begin, r_end); lagsrc has a zero halo below index 0.
#ifndef IRA_DIFF_LG
#define IRA_DIFF_LG 9
#endif
// lags per thread: two LDS reads feed LG FMAs.  ODD on purpose (round 4): neighbouring lanes read lagsrc LG doubles apart, and with
// LG = 8 (16 dwords) the 30 lag groups of a half wave fell on FOUR bank pairs -- an 8-way conflict on every second LDS read,
// which made the windowed autocorrelation LDS-bound at 15 % of the float64 vector peak.  9 doubles = 18 dwords: 32 lanes, 32 bank
// pairs.  (16 lags: 32 KB of partials, one workgroup fewer per CU: 1.79 vs 1.27 ms.)
constexpr int LG = IRA_DIFF_LG;
__device__ __forceinline__ void lag_group(const double* cur, const double* lagsrc, int b0, int r_begin, int r_end,
                                          double (&acc)[LG]) {
  double a[LG], w[LG];
#pragma unroll
  for (int j = 0; j < LG; ++j) a[j] = 0.0;
  if (r_begin < r_end) {
    const double* c = cur + r_begin;
    const double* l = lagsrc + r_begin - b0;
#pragma unroll
    for (int j = 1; j < LG; ++j) w[j] = l[-j];
    w[0] = 0.0;
    int r = r_begin;
    // LG rows per trip with the window renamed at compile time instead of moved (at unrolled step u slot (j - u) mod LG
    // holds the value for lag offset j; the newest value takes the oldest one's slot): 2 LG LDS reads in flight together, no
    // register moves; every accumulator still sees its rows in order (same sums as the one-row loop below).
    for (; r + LG <= r_end; r += LG) {
#pragma unroll
      for (int u = 0; u < LG; ++u) {
        const double sn = c[u];
        w[(LG - u) % LG] = l[u];
#pragma unroll
        for (int j = 0; j < LG; ++j) a[j] = fma(sn, w[(j - u + LG) % LG], a[j]);
      }
      c += LG; l += LG;
    }
    for (; r < r_end; ++r) {
      const double sn = *c++;
      w[0] = *l++;
#pragma unroll
      for (int j = 0; j < LG; ++j) a[j] = fma(sn, w[j], a[j]);
#pragma unroll
      for (int j = LG - 1; j > 0; --j) w[j] = w[j - 1];
    }
  }
#pragma unroll
  for (int j = 0; j < LG; ++j) acc[j] = a[j];
}

// max over lags lag_lo..lag_hi of |sum_n cur[n] lagsrc[n-lag]| (all threads call; `red` holds nsub * LG * ngroups doubles)
__device__ double max_abs_lag_sum(const double* cur, const double* lagsrc, int n, int lag_lo, int lag_hi, double* red,
                                  double* wave_max) {
  const int tid = threadIdx.x;
  const int nlag = lag_hi - lag_lo + 1;
  double best = 0.0;
  if (nlag > 0) {
    const int ngroups = (nlag + LG - 1) / LG;
    const int nsub = ngroups >= DF_THREADS ? 1 : DF_THREADS / ngroups;
    const int sub_len = (n + nsub - 1) / nsub;
    for (int item = tid; item < ngroups * nsub; item += DF_THREADS) {
      const int g = item % ngroups, sub = item / ngroups;
      const int r_begin = sub * sub_len;
      const int r_end = (r_begin + sub_len < n) ? r_begin + sub_len : n;
      double acc[LG];
      lag_group(cur, lagsrc, lag_lo + LG * g, r_begin, r_end, acc);
      double* o = red + (size_t)sub * (LG * ngroups) + LG * g;
#pragma unroll
      for (int j = 0; j < LG; ++j) o[j] = acc[j];
    }
    __syncthreads();
    for (int b = tid; b < nlag; b += DF_THREADS) {
      double s = 0.0;
      for (int sub = 0; sub < nsub; ++sub) s += red[(size_t)sub * (LG * ngroups) + b];
      best = fmax(best, fabs(s));
    }
  }
  // block max
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmax(best, __shfl_xor(best, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) wave_max[tid >> 6] = best;
  __syncthreads();
  double m = 0.0;
  for (int w = 0; w < DF_THREADS / 64; ++w) m = fmax(m, wave_max[w]);
  __syncthreads();
  return m;
}

__device__ double block_sum(double v, double* wave_tmp) {
  v = ira::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wave_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < DF_THREADS / 64; ++w) s += wave_tmp[w];
  __syncthreads();
  return s;
}

// Partial sums of max_abs_lag_sum: nsub * LG * ngroups doubles with nsub = floor(DF_THREADS / ngroups) for ANY lag count up to
// the largest one a kernel asks for (2 max_lag + 1, stereo) -- not just for that count itself: fewer lags mean more row
// sub-ranges, and nsub * ngroups can come closer to DF_THREADS (the mono kernel clips its lag range to the window length).
__host__ __device__ inline size_t red_doubles(int max_lag) {
  const int ngroups_max = (2 * max_lag + 1 + LG - 1) / LG;
  return (size_t)LG * (ngroups_max > DF_THREADS ? ngroups_max : DF_THREADS);
}

struct DiffLayout {
  // dynamic LDS carve-up for a window of n samples and max_lag
  float* wf;        // n raw float32 samples, then reused for w0 (float32) and w0^2
  double* a0;       // halo + n  (zero halo of `halo` doubles in front)
  double* b0;       // stereo only
  double* red;      // lag partials
  int halo;
};

__device__ __forceinline__ DiffLayout carve(unsigned char* smem, int n, int max_lag, bool stereo) {
  DiffLayout L;
  L.halo = max_lag + LG;
  double* d = reinterpret_cast<double*>(smem);
  L.a0 = d + L.halo;
  d += L.halo + n;
  if (stereo) {
    L.b0 = d + L.halo;
    d += L.halo + n;
  } else {
    L.b0 = nullptr;
  }
  L.red = d;
  d += red_doubles(max_lag);
  L.wf = reinterpret_cast<float*>(d);
  return L;
}

size_t diff_lds_bytes(int n, int max_lag, bool stereo) {
  const int halo = max_lag + LG;
  size_t doubles = (size_t)(halo + n) * (stereo ? 2 : 1) + red_doubles(max_lag);
  return doubles * sizeof(double) + (size_t)n * sizeof(float);
}

// Mean-removed window in float32 (numpy semantics) -> dst (float64 copy with zero halo) and wf (float32 w0).
__device__ void stage_mean_removed(const float* __restrict__ src, int n, float* wf, double* dst, int halo,
                                   const PwPlan& plan, PwShared& pw) {
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += DF_THREADS) wf[i] = src[i];
  for (int i = tid; i < halo; i += DF_THREADS) dst[-1 - i] = 0.0;
  __syncthreads();
  const float mean = np_pairwise_sum_f32(wf, plan, pw) / (float)n;
  for (int i = tid; i < n; i += DF_THREADS) {
    const float v = wf[i] - mean;
    wf[i] = v;
    dst[i] = (double)v;
  }
  __syncthreads();
}

__global__ __launch_bounds__(DF_THREADS) void diffusion_mono_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ xoff, const int32_t* __restrict__ nframes, int win, int hop,
    int max_lag, double thr_rms, double gauss_expected, float* __restrict__ ac_out, float* __restrict__ ed_out,
    const int64_t* __restrict__ out_off, const PwPlan plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ PwShared pw;
  __shared__ double wtmp[DF_THREADS / 64];
  const int e = blockIdx.y, f = blockIdx.x;
  if (f >= nframes[e]) return;
  const int tid = threadIdx.x;
  const int n = win;
  const DiffLayout L = carve(smem, n, max_lag, false);
  const float* src = x + xoff[e] + (int64_t)f * hop;
  const float qnan = __uint_as_float(0x7fc00000u);
  stage_mean_removed(src, n, L.wf, L.a0, L.halo, plan, pw);

  // ---- max |autocorrelation| -----------------------------------------------------------------------------------
  double den = 0.0;
  for (int i = tid; i < n; i += DF_THREADS) den = fma(L.a0[i], L.a0[i], den);
  den = block_sum(den, wtmp);
  const int lmax = max_lag < n - 2 ? max_lag : n - 2;
  const double peak = max_abs_lag_sum(L.a0, L.a0, n, 1, lmax, L.red, wtmp);
  if (tid == 0) ac_out[out_off[e] + f] = (n < 4 || den <= 1e-20) ? qnan : (float)(peak / den);

  // ---- echo density (float32 arithmetic of the reference, bit for bit) --------------------------------------------
  // rms = sqrt(mean(w0 * w0)) with float32 products and numpy's pairwise float32 sum; wf is overwritten by the squares,
  // the comparison below reads |w0| back from the float64 copy (exact).
  for (int i = tid; i < n; i += DF_THREADS) L.wf[i] = L.wf[i] * L.wf[i];
  __syncthreads();
  const float msq = np_pairwise_sum_f32(L.wf, plan, pw) / (float)n;
  const float rms = sqrtf(msq);
  const float thr = (float)(thr_rms * (double)rms);          // python float product, then the weak-scalar cast to float32
  double cnt = 0.0;
  for (int i = tid; i < n; i += DF_THREADS) cnt += (fabsf((float)L.a0[i]) > thr) ? 1.0 : 0.0;
  cnt = block_sum(cnt, wtmp);
  if (tid == 0) {
    float out;
    if (n < 4 || (double)rms <= 1e-20) out = qnan;
    else {
      const double frac = cnt / (double)n;
      if (gauss_expected < 0.0) out = (float)frac;                       // no normalisation
      else out = gauss_expected <= 1e-12 ? qnan : (float)(frac / gauss_expected);
    }
    ed_out[out_off[e] + f] = out;
  }
}

__global__ __launch_bounds__(DF_THREADS) void diffusion_stereo_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ loff, const int64_t* __restrict__ roff,
    const int32_t* __restrict__ nframes, int win, int hop, int max_lag, float* __restrict__ corr0_out,
    float* __restrict__ iacc_out, const int64_t* __restrict__ out_off, const PwPlan plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ PwShared pw;
  __shared__ double wtmp[DF_THREADS / 64];
  const int e = blockIdx.y, f = blockIdx.x;
  if (f >= nframes[e]) return;
  const int tid = threadIdx.x;
  const int n = win;
  const DiffLayout L = carve(smem, n, max_lag, true);
  const float qnan = __uint_as_float(0x7fc00000u);
  stage_mean_removed(x + loff[e] + (int64_t)f * hop, n, L.wf, L.a0, L.halo, plan, pw);
  stage_mean_removed(x + roff[e] + (int64_t)f * hop, n, L.wf, L.b0, L.halo, plan, pw);
  double aa = 0.0, bb = 0.0, ab = 0.0;
  for (int i = tid; i < n; i += DF_THREADS) {
    aa = fma(L.a0[i], L.a0[i], aa); bb = fma(L.b0[i], L.b0[i], bb); ab = fma(L.a0[i], L.b0[i], ab);
  }
  aa = block_sum(aa, wtmp); bb = block_sum(bb, wtmp); ab = block_sum(ab, wtmp);
  const double den = sqrt(aa * bb);
  if (tid == 0) corr0_out[out_off[e] + f] = (n < 4 || aa <= 1e-20 || bb <= 1e-20) ? qnan : (float)(ab / den);
  const int lmax = max_lag < n - 2 ? max_lag : n - 2;
  // positive lags: sum_k a0[k] b0[k+lag] = sum_n b0[n] a0[n-lag]  (lag 0 included); negative: roles swapped, lag >= 1
  const double p1 = max_abs_lag_sum(L.b0, L.a0, n, 0, lmax, L.red, wtmp);
  const double p2 = max_abs_lag_sum(L.a0, L.b0, n, 1, lmax, L.red, wtmp);
  if (tid == 0) iacc_out[out_off[e] + f] = (n < 4 || den <= 1e-20) ? qnan : (float)(fmax(p1, p2) / den);
}

int32_t diff_check(int32_t nb, int32_t max_frames, int32_t win, int32_t hop, int32_t max_lag) {
  if (nb < 0 || max_frames < 0) return IRA_E_SIZE;
  if (win < 4 || win > DF_MAX_WIN || hop < 1 || max_lag < 1 || max_lag > DF_MAX_LAG) return IRA_E_SIZE;
  if (nb > 65535) return IRA_E_SIZE;
  return IRA_OK;
}

}  // namespace

extern "C" int32_t ira_diffusion(const float* x_dev, const int64_t* xoff_dev, const int32_t* nframes_dev, int32_t nb,
                                 int32_t max_frames, int32_t win, int32_t hop, int32_t max_lag, double thr_rms,
                                 double gauss_expected, float* ac_dev, float* ed_dev, const int64_t* out_off_dev,
                                 void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(xoff_dev); IRA_CHECK_PTR(nframes_dev); IRA_CHECK_PTR(ac_dev);
  IRA_CHECK_PTR(ed_dev); IRA_CHECK_PTR(out_off_dev);
  const int32_t rc = diff_check(nb, max_frames, win, hop, max_lag);
  if (rc != IRA_OK || nb == 0 || max_frames == 0) return rc;
  const size_t lds = diff_lds_bytes(win, max_lag, false);
  if (lds > 150 * 1024) return IRA_E_SIZE;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&diffusion_mono_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return ira_hip_status(e);
  }
  diffusion_mono_kernel<<<dim3(max_frames, nb), DF_THREADS, lds, (hipStream_t)stream>>>(
      x_dev, xoff_dev, nframes_dev, win, hop, max_lag, thr_rms, gauss_expected, ac_dev, ed_dev, out_off_dev, pw_plan_of(win));
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_diffusion_stereo(const float* x_dev, const int64_t* loff_dev, const int64_t* roff_dev,
                                        const int32_t* nframes_dev, int32_t nb, int32_t max_frames, int32_t win,
                                        int32_t hop, int32_t max_lag, float* corr0_dev, float* iacc_dev,
                                        const int64_t* out_off_dev, void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(loff_dev); IRA_CHECK_PTR(roff_dev); IRA_CHECK_PTR(nframes_dev);
  IRA_CHECK_PTR(corr0_dev); IRA_CHECK_PTR(iacc_dev); IRA_CHECK_PTR(out_off_dev);
  const int32_t rc = diff_check(nb, max_frames, win, hop, max_lag);
  if (rc != IRA_OK || nb == 0 || max_frames == 0) return rc;
  const size_t lds = diff_lds_bytes(win, max_lag, true);
  if (lds > 150 * 1024) return IRA_E_SIZE;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&diffusion_stereo_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return ira_hip_status(e);
  }
  diffusion_stereo_kernel<<<dim3(max_frames, nb), DF_THREADS, lds, (hipStream_t)stream>>>(
      x_dev, loff_dev, roff_dev, nframes_dev, win, hop, max_lag, corr0_dev, iacc_dev, out_off_dev, pw_plan_of(win));
  IRA_RETURN_LAUNCH();
}
