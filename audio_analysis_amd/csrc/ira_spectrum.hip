// Post-processing of whole-segment spectra (complex float64 half spectra produced by ira_rfft_any):
//   magnitude in dB (float32), phase (float64 angle), numpy.unwrap as a parallel scan, and the summary
//   statistics of reference frequency_response.py:238-260 and filterplot.py:173-191.
// Compiled with -ffp-contract=off (the unwrap correction arithmetic mirrors NumPy's operation by operation).
#include <cmath>

#include "ira_common.h"
#include "ira_log.h"

namespace {

typedef ira::cplx<double> cd;
constexpr double kPi = 3.14159265358979323846;

// ---- |X| -> dB (float32) and angle(X) (float64) --------------------------------------------------------------
// 20 log10(hypot(re, im)) = 10 log10(re^2 + im^2) through the table log2 of ira_log.h (25 instructions instead of ~160 for
// hypot + log10; the float32 result is the same unless the float64 value lies within 1e-15 of a rounding tie), and the angle
// through a 65-entry arctangent table: t = min / max of |re|, |im|, atan t = atan(k/64) + atan((t - k/64) / (1 + t k/64))
// with the second term by its series to r^9 (|r| <= 1/128: next term < 1e-24), then the usual octant / sign fix-ups:
// ~50 instructions instead of ~150, 1-2 ulp.  Zero, infinite and NaN operands take the library routines.  With both the
// kernel is bound by its 28 bytes per bin instead of by the VALU (times: DESIGN.md section 4).
constexpr int ATAN_TAB = 64;

__device__ __forceinline__ void build_atan_table(double* tab, int tid) {
  if (tid <= ATAN_TAB) tab[tid] = atan((double)tid * (1.0 / ATAN_TAB));
}

__device__ __forceinline__ double atan2_table(double y, double x, const double* tab) {
  const double ax = fabs(x), ay = fabs(y);
  const double mx = fmax(ax, ay), mn = fmin(ax, ay);
  if (!(mx > 0.0 && mx < 1.0e300 && mn > 1.0e-300)) return atan2(y, x);        // zero / tiny / huge / infinite / NaN: library
  const double t = mn / mx;
  const int k = (int)(t * (double)ATAN_TAB + 0.5);
  const double t0 = (double)k * (1.0 / ATAN_TAB);
  const double r = (t - t0) / fma(t, t0, 1.0);
  const double r2 = r * r;
  double s = fma(r2, 1.0 / 9.0, -1.0 / 7.0);
  s = fma(r2, s, 0.2);
  s = fma(r2, s, -1.0 / 3.0);
  s = fma(r2, s, 1.0);
  double a = fma(r, s, tab[k]);                                                // atan(t), 0 <= t <= 1
  if (ay > ax) a = 1.5707963267948966 - a;
  if (x < 0.0) a = kPi - a;
  return copysign(a, y);
}

// ---- numpy.unwrap correction between two neighbouring phases (the definition follows numpy operation by operation) ----
__device__ __forceinline__ double unwrap_correction(double prev, double cur) {
  const double dd = cur - prev;
  const double period = 2.0 * kPi;
  double a = dd + kPi;               // dd - interval_low
  // numpy floor-mod for a positive divisor: md = fmod(a, period), moved up by one period when negative.  Phases come from
  // atan2, so |dd| <= 2 pi and a lies in [-pi, 3 pi]: there fmod(a, period) is a itself or the EXACT difference a - period
  // (Sterbenz), no division loop needed -- the library fmod (~150 instructions for a float64) was most of the arithmetic of
  // the unwrap, which made a kernel that moves 12 bytes per bin VALU-bound.  Anything outside (a caller's own phase array)
  // takes the library routine.
  double md;
  if (a >= 0.0 && a < period) md = a;
  else if (a >= period && a < 2.0 * period) md = a - period;
  else if (a < 0.0 && a > -period) md = a;
  else md = fmod(a, period);
  if (md != 0.0) {
    if (md < 0.0) md += period;
  } else {
    md = 0.0;
  }
  double ddmod = md + (-kPi);
  if (ddmod == -kPi && dd > 0.0) ddmod = kPi;
  double corr = ddmod - dd;
  if (fabs(dd) < kPi) corr = 0.0;
  return corr;
}

__global__ __launch_bounds__(256) void mag_phase_kernel(const cd* __restrict__ spec, const int64_t* __restrict__ spec_off,
                                 const int32_t* __restrict__ L, double floor_lin, float* __restrict__ mag_db,
                                 const int64_t* __restrict__ mag_off, double* __restrict__ phase,
                                 const int64_t* __restrict__ phase_off, const int32_t* __restrict__ packed) {
  __shared__ ira::LogTabEntry ltab[ira::LOGTAB_N];
  __shared__ double atab[ATAN_TAB + 1];
  ira::build_log_table(ltab, threadIdx.x);
  build_atan_table(atab, threadIdx.x);
  __syncthreads();
  const int e = blockIdx.y;
  const long long nb = (long long)L[e] / 2 + 1;
  const cd* s = spec + spec_off[e];
  float* mo = mag_db + mag_off[e];
  double* po = phase ? phase + phase_off[e] : nullptr;
  const double floor_p = floor_lin * floor_lin;
  const float floor_db32 = (float)(20.0 * log10(floor_lin));
  // packed element: s holds Z = DFT_l(x[2m] + i x[2m+1]), l = L/2 values (ira_rfft_any, keep_packed), and bin k of the real
  // signal's spectrum is formed here, where Z[k] and Z[l - k] are two contiguous streams -- the formulas of
  // half_split_kernel (ira_fftlong.hip) without its pass over memory
  const bool pk = packed != nullptr && ira::uniform(packed[e]) != 0;
  const long long l = (long long)L[e] / 2;
  // W_2l^k = exp(-i pi k / l) along a thread's bins (a grid stride apart) by rotation from an exactly reduced start value: two
  // sincospi per thread instead of one per bin (16 bins per thread; per bin it cost as much as the dB and the angle together)
  const long long k_first = (long long)blockIdx.x * blockDim.x + threadIdx.x, k_step = (long long)gridDim.x * blockDim.x;
  double cs = 1.0, sn = 0.0, rc = 1.0, rs = 0.0;
  if (pk) {
    sincospi(-(double)(k_first < nb ? k_first : 0) / (double)l, &sn, &cs);
    sincospi(-(double)(k_step % (2 * l)) / (double)l, &rs, &rc);
  }
  for (long long k = k_first; k < nb; k += k_step) {
    cd v;
    if (pk) {
      const cd zk = s[k == l ? 0 : k], zl = s[(k == 0 || k == l) ? 0 : l - k];
      const cd ev = {0.5 * (zk.re + zl.re), 0.5 * (zk.im - zl.im)};
      const cd od = {0.5 * (zk.im + zl.im), 0.5 * (zl.re - zk.re)};
      v = {ev.re + (cs * od.re - sn * od.im), ev.im + (cs * od.im + sn * od.re)};
      if (k == 0 || k == l) v.im = 0.0;                            // DC / Nyquist of a real signal
      const double nc = cs * rc - sn * rs;
      sn = sn * rc + cs * rs;
      cs = nc;
    } else {
      v = s[k];
    }
    const double p = v.re * v.re + v.im * v.im;
    float db;
    if (p > 1.0e-280 && p < 1.0e280 && floor_lin > 1.0e-140) {
      // numpy.maximum(|X|, floor) in the squared domain; a value AT the floor gives the floor's own float32 dB value
      db = p > floor_p ? (float)(3.0102999566398120 * ira::log2_table(p, ltab)) : floor_db32;
    } else {
      const double a = hypot(v.re, v.im);
      const double m = (a != a) ? a : fmax(a, floor_lin);        // numpy.maximum keeps NaN (frequency_response.py:215-218)
      db = (float)(20.0 * log10(m));
    }
    mo[k] = db;
    if (po) po[k] = atan2_table(v.im, v.re, atab);
  }
}

// ---- numpy.unwrap (period 2*pi) ------------------------------------------------------------------------------
//   dd = diff(p); ddmod = mod(dd + pi, 2pi) - pi; ddmod[(ddmod == -pi) & (dd > 0)] = pi
//   corr = ddmod - dd; corr[|dd| < pi] = 0; out[1:] = p[1:] + cumsum(corr)
constexpr int UW_THREADS = 1024;
constexpr int UW_PER = 4;
constexpr int UW_TILE = UW_THREADS * UW_PER;

__global__ __launch_bounds__(UW_THREADS) void unwrap_kernel(const double* __restrict__ phase,
                                                            const int64_t* __restrict__ phase_off,
                                                            const int32_t* __restrict__ L, int do_unwrap,
                                                            double out_scale, float* __restrict__ out,
                                                            const int64_t* __restrict__ out_off,
                                                            double* __restrict__ out64) {
  __shared__ double wave_tot[UW_THREADS / IRA_WAVE];
  __shared__ double tile_total;
  const int e = blockIdx.x;
  const long long n = (long long)L[e] / 2 + 1;
  const double* p = phase + phase_off[e];
  float* o = out ? out + out_off[e] : nullptr;
  double* o64 = out64 ? out64 + out_off[e] : nullptr;     // float64 radians (group delay differentiates this)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double carry = 0.0;
  // One workgroup walks the spectrum tile by tile (the correction prefix is sequential across tiles); the NEXT tile's
  // phases are loaded before the current tile is scanned, so a tile costs max(memory latency, scan) instead of their sum.
  double nv[UW_PER], nprev;
  {
    const long long i0 = (long long)UW_PER * t;
    nprev = (i0 >= 1 && i0 - 1 < n) ? p[i0 - 1] : 0.0;
#pragma unroll
    for (int r = 0; r < UW_PER; ++r) nv[r] = (i0 + r < n) ? p[i0 + r] : 0.0;
  }
  for (long long base = 0; base < n; base += UW_TILE) {
    double c[UW_PER], v[UW_PER];
    const long long i0 = base + (long long)UW_PER * t;
    double prev = nprev;
#pragma unroll
    for (int r = 0; r < UW_PER; ++r) v[r] = nv[r];
    if (base + UW_TILE < n) {
      const long long j0 = i0 + UW_TILE;
      nprev = (j0 - 1 < n) ? p[j0 - 1] : 0.0;
#pragma unroll
      for (int r = 0; r < UW_PER; ++r) nv[r] = (j0 + r < n) ? p[j0 + r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < UW_PER; ++r) {
      const long long i = i0 + r;
      c[r] = (do_unwrap && i >= 1 && i < n) ? unwrap_correction(prev, v[r]) : 0.0;
      prev = v[r];
    }
    // inclusive prefix within the thread, then across the wave, then across waves
    c[1] += c[0]; c[2] += c[1]; c[3] += c[2];
    double incl = c[3];
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) {
      const double dn = __shfl_up(incl, o2, 64);
      if (lane >= o2) incl += dn;
    }
    double excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0.0;
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    double before = 0.0;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    const double add = carry + (before + excl);
#pragma unroll
    for (int r = 0; r < UW_PER; ++r) {
      const long long i = i0 + r;
      if (i < n) {
        const double u = v[r] + (c[r] + add);
        if (o) o[i] = (float)(u * out_scale);
        if (o64) o64[i] = u;
      }
    }
    if (t == UW_THREADS - 1) tile_total = c[3] + add;
    __syncthreads();
    carry = tile_total;
    // no third barrier: the next tile rewrites wave_tot / tile_total only after barriers every thread must reach
    // after its reads of this tile's values
  }
}

// ---- group delay: gd = -numpy.gradient(phase, w), w[k] = 2 pi ((k * val) / sr)  (reference group_delay.py:113-124) ------
// numpy.gradient with a coordinate ARRAY takes the uniform-spacing formula only if every diff(w) is bit-identical and
// the three-point non-uniform formula otherwise; both are reproduced, the choice is made by gd_uniform_kernel.
__device__ __forceinline__ double gd_w(long long k, double val, double sr) {
  return 6.283185307179586 * (((double)k * val) / sr);      // (2.0 * np.pi) * (freq / sr)
}

__global__ void gd_uniform_kernel(const int32_t* __restrict__ nbins, const double* __restrict__ val, double sr,
                                  int32_t* __restrict__ nonuniform) {
  const int e = blockIdx.y;
  const long long n = nbins[e];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // diff index, 0 .. n-2
  if (i + 1 >= n || i == 0) return;
  const double v = val[e];
  const double d0 = gd_w(1, v, sr) - gd_w(0, v, sr);
  const double di = gd_w(i + 1, v, sr) - gd_w(i, v, sr);
  if (di != d0) nonuniform[e] = 1;                                            // benign race: every writer stores 1
}

__global__ void gd_gradient_kernel(const double* __restrict__ phase, const int64_t* __restrict__ off,
                                   const int32_t* __restrict__ nbins, const double* __restrict__ val, double sr,
                                   const int32_t* __restrict__ nonuniform, double* __restrict__ out) {
  const int e = blockIdx.y;
  const long long n = nbins[e];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* f = phase + off[e];
  double* o = out + off[e];
  const double v = val[e];
  if (n < 2) { o[i] = __longlong_as_double(0x7ff8000000000000ll); return; }   // numpy raises; callers never ask
  double g;
  if (i == 0) {
    g = (f[1] - f[0]) / (gd_w(1, v, sr) - gd_w(0, v, sr));
  } else if (i == n - 1) {
    g = (f[n - 1] - f[n - 2]) / (gd_w(n - 1, v, sr) - gd_w(n - 2, v, sr));
  } else if (!nonuniform[e]) {
    const double dx = gd_w(1, v, sr) - gd_w(0, v, sr);
    g = (f[i + 1] - f[i - 1]) / (2.0 * dx);
  } else {
    const double dx1 = gd_w(i, v, sr) - gd_w(i - 1, v, sr);
    const double dx2 = gd_w(i + 1, v, sr) - gd_w(i, v, sr);
    const double a = -(dx2) / (dx1 * (dx1 + dx2));
    const double b = (dx2 - dx1) / (dx1 * dx2);
    const double c = dx1 / (dx2 * (dx1 + dx2));
    g = a * f[i - 1] + b * f[i] + c * f[i + 1];
  }
  o[i] = -g;
}

// ---- order statistics (numpy.median / numpy.percentile need the k-th smallest of ~1e5..1e6 float64 values) -----------
// MSB-first radix select on the order-preserving uint64 image of the doubles, eight 8-bit digits, up to OS_MAX_RANKS
// ranks per segment in one sweep: every pass histograms the next digit of the values that still match each rank's
// prefix, then each rank descends into the bucket holding it.  One workgroup per segment; the data (<= a few MB)
// stays in L2 across the eight passes.  -0.0 sorts before +0.0 and NaNs sort last (numpy would propagate NaN; group
// delay has none).
constexpr int OS_THREADS = 1024;
constexpr int OS_MAX_RANKS = 8;

__device__ __forceinline__ unsigned long long os_key(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double os_unkey(unsigned long long k) {
  const unsigned long long u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)u);
}

// Round 4: once the buckets that hold the wanted ranks are small (after two digits as a rule: a few hundred values of
// ~2e5), the values that still matter are copied into LDS lists -- one more sweep over the segment -- and the remaining
// digits are resolved there: three to four sweeps over memory instead of eight (1.02 -> ~0.45 ms per 256 x 2.2e5 values).
constexpr int OS_LISTS = 4;          // distinct prefixes that may be compacted (three pairs of neighbouring ranks: three)
constexpr int OS_CAP = 1536;         // values per list (4 x 1536 doubles = 48 KB: the kernel stays below 64 KB of static LDS)

__global__ __launch_bounds__(OS_THREADS) void order_stats_kernel(const double* __restrict__ values,
                                                                 const int64_t* __restrict__ off,
                                                                 const int32_t* __restrict__ count,
                                                                 const int64_t* __restrict__ ranks, int nranks,
                                                                 double* __restrict__ out) {
  __shared__ unsigned int hist[OS_MAX_RANKS][256];
  __shared__ unsigned long long prefix[OS_MAX_RANKS];
  __shared__ long long remaining[OS_MAX_RANKS];
  __shared__ long long bucket[OS_MAX_RANKS];           // size of the bucket the rank descended into in the last pass
  __shared__ double list[OS_LISTS][OS_CAP];
  __shared__ int list_n[OS_LISTS];
  __shared__ int list_of[OS_MAX_RANKS];                 // which list holds rank r's candidates (after compaction)
  __shared__ int compact_now, compacted;
  __shared__ long long scan_tot[4];
  const int e = blockIdx.x, tid = threadIdx.x;
  const long long n = count[e];
  const double* v = values + off[e];
  double* o = out + (long long)e * nranks;
  if (n <= 0) {
    if (tid < nranks) o[tid] = __longlong_as_double(0x7ff8000000000000ll);
    return;
  }
  if (tid < nranks) {
    prefix[tid] = 0ull;
    long long r = ranks[(long long)e * nranks + tid];
    remaining[tid] = r < 0 ? 0 : (r > n - 1 ? n - 1 : r);
    bucket[tid] = n;
  }
  if (tid == 0) { compact_now = 0; compacted = 0; }
  __shared__ int leader[OS_MAX_RANKS];
  const int lane = tid & 63;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    for (int i = tid; i < nranks * 256; i += OS_THREADS) hist[i >> 8][i & 255] = 0u;
    if (tid == 0) {
      // ranks that still share a prefix (all of them in the first passes) share one histogram
      int nlead = 0;
      long long biggest = 0;
      for (int r = 0; r < nranks; ++r) {
        int l = r;
        for (int q = 0; q < r; ++q)
          if (prefix[q] == prefix[r]) { l = q; break; }
        leader[r] = l;
        if (l == r) { ++nlead; biggest = bucket[r] > biggest ? bucket[r] : biggest; }
      }
      compact_now = (!compacted && pass >= 1 && nlead <= OS_LISTS && biggest <= OS_CAP) ? 1 : 0;
      if (compact_now) {
        int next = 0;
        for (int r = 0; r < nranks; ++r) {
          if (leader[r] == r) { list_of[r] = next; list_n[next] = 0; ++next; }
          else list_of[r] = list_of[leader[r]];
        }
      }
    }
    __syncthreads();
    unsigned long long pf[OS_MAX_RANKS];
    bool lead[OS_MAX_RANKS];
#pragma unroll
    for (int r = 0; r < OS_MAX_RANKS; ++r) {
      pf[r] = r < nranks ? prefix[r] : 0ull;
      lead[r] = r < nranks && leader[r] == r;
    }
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    constexpr int OS_U = 16;         // loads in flight per thread: at 4 a sweep was 53 dependent round trips per workgroup (0.85 of 0.95 ms)
    const long long trips = (n + (long long)OS_THREADS * OS_U - 1) / ((long long)OS_THREADS * OS_U);
    if (compact_now) {
      // one more sweep over the segment: every value whose decided digits match a leader's prefix goes to that leader's list
      for (long long it = 0; it < trips; ++it) {
        double dv[OS_U];
        bool vv[OS_U];
#pragma unroll
        for (int u = 0; u < OS_U; ++u) {
          const long long i = (it * OS_U + u) * OS_THREADS + tid;
          vv[u] = i < n;
          dv[u] = v[vv[u] ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < OS_U; ++u) {
          const unsigned long long k = os_key(dv[u]);
#pragma unroll
          for (int r = 0; r < OS_MAX_RANKS; ++r) {
            if (!lead[r]) continue;
            if (vv[u] && (k & himask) == pf[r]) {
              const int slot = atomicAdd(&list_n[list_of[r]], 1);
              if (slot < OS_CAP) list[list_of[r]][slot] = dv[u];
            }
          }
        }
      }
      __syncthreads();
      if (tid == 0) { compacted = 1; compact_now = 0; }
      __syncthreads();
    }
    if (compacted) {
      // the candidates live in LDS: histogram the next digit of those that match each leading rank's full prefix
#pragma unroll
      for (int r = 0; r < OS_MAX_RANKS; ++r) {
        if (!lead[r]) continue;
        const int li = list_of[r];
        const int cnt = list_n[li] < OS_CAP ? list_n[li] : OS_CAP;
        for (int i = tid; i < cnt; i += OS_THREADS) {
          const unsigned long long k = os_key(list[li][i]);
          if ((k & himask) == pf[r]) atomicAdd(&hist[r][(unsigned int)(k >> shift) & 255u], 1u);
        }
      }
    } else {
      // uniform trip count (ballots below need every lane of the wave in the loop); OS_U independent loads per thread are
      // in flight before the first one is used (one workgroup walking 2e5 values with a load per iteration is bound by
      // one memory latency per value)
      for (long long it = 0; it < trips; ++it) {
        unsigned long long kk[OS_U];
        bool vv[OS_U];
#pragma unroll
        for (int u = 0; u < OS_U; ++u) {
          const long long i = (it * OS_U + u) * OS_THREADS + tid;
          vv[u] = i < n;
          kk[u] = os_key(v[vv[u] ? i : n - 1]);
        }
#pragma unroll
        for (int u = 0; u < OS_U; ++u) {
          const bool valid = vv[u];
          const unsigned long long k = kk[u];
          const unsigned int digit = (unsigned int)(k >> shift) & 255u;
#pragma unroll
          for (int r = 0; r < OS_MAX_RANKS; ++r) {
            if (!lead[r]) continue;                                      // wave-uniform
            const bool m = valid && (k & himask) == pf[r];
            const unsigned long long act = __ballot(m);
            if (act == 0ull) continue;
            // the early digits are the same for almost every value (sign / exponent): one atomic for the whole group
            const int first = __ffsll((long long)act) - 1;
            const unsigned int d0 = (unsigned int)__shfl((int)digit, first, 64);
            const unsigned long long same = __ballot(m && digit == d0);
            if (lane == first) atomicAdd(&hist[r][d0], (unsigned int)__popcll(same));
            if (m && digit != d0) atomicAdd(&hist[r][digit], 1u);
          }
        }
      }
    }
    __syncthreads();
    // descend: the bucket d of rank r is the one whose cumulative count first exceeds its remaining rank -- a 256-bin scan by
    // the first four waves per rank (one thread walking the bins was 255 dependent LDS reads per pass and rank)
    for (int r = 0; r < nranks; ++r) {
      const int l = leader[r];
      const long long rem = remaining[r];
      const long long c = tid < 256 ? (long long)hist[l][tid] : 0ll;
      long long incl = c;
#pragma unroll
      for (int o2 = 1; o2 < 64; o2 <<= 1) {
        const long long up = __shfl_up(incl, o2, 64);
        if (lane >= o2) incl += up;
      }
      if (tid < 256 && lane == 63) scan_tot[tid >> 6] = incl;
      __syncthreads();
      if (tid < 256) {
        for (int w = 0; w < (tid >> 6); ++w) incl += scan_tot[w];
        const long long excl = incl - c;
        // (the last bin takes whatever is left, like the sequential walk that stopped at d = 255)
        if ((excl <= rem && rem < incl) || (tid == 255 && rem >= incl)) {
          remaining[r] = rem - excl;
          bucket[r] = c;
          prefix[r] = prefix[r] | ((unsigned long long)tid << shift);
        }
      }
      __syncthreads();
    }
  }
  if (tid < nranks) o[tid] = os_unkey(prefix[tid]);
}

// ---- summary statistics ------------------------------------------------------------------------------------------
// out record (8 doubles): [0] bins in range [1] peak bin (absolute index) [2] peak frequency (float32 value)
// [3] sum(f*lin) [4] sum(lin) [5] first in-range frequency [6] index nearest 1 kHz [7] magnitude_db there
constexpr int ST_THREADS = 1024;

__global__ __launch_bounds__(ST_THREADS) void stats_kernel(const float* __restrict__ mag_db,
                                                           const int64_t* __restrict__ mag_off,
                                                           const int32_t* __restrict__ L,
                                                           const double* __restrict__ freq_val, float f_min, float f_max,
                                                           float probe_hz, double* __restrict__ out) {
  __shared__ double red[4][ST_THREADS / IRA_WAVE];
  __shared__ unsigned long long kred[2][ST_THREADS / IRA_WAVE];
  __shared__ long long lred[ST_THREADS / IRA_WAVE];
  __shared__ double etab[ira::EXPTAB_N];
  const int e = blockIdx.x;
  const long long n = (long long)L[e] / 2 + 1;
  const float* m = mag_db + mag_off[e];
  const double val = freq_val[e];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  ira::build_exp_table(etab, t);
  __syncthreads();
  double cnt = 0.0, sfl = 0.0, sl = 0.0;
  // argmax of float32 dB with first-max-wins: order-preserving key of the float, then smallest index
  unsigned long long best = 0ull;
  // argmin |f - probe| first-min-wins: key = (bits(|d|) << 32 | idx) minimised
  unsigned long long near = ~0ull;
  long long first_in = n;
  // ST_U loads per thread in flight before the first is used; the per-thread order of the additions is unchanged
  constexpr int ST_U = 16;          // (64 KB of loads in flight per CU: one workgroup per spectrum has only its own loads to hide memory latency with)
  for (long long k0 = t; k0 < n; k0 += (long long)ST_THREADS * ST_U) {
    float dbv[ST_U];
#pragma unroll
    for (int u = 0; u < ST_U; ++u) {
      const long long k = k0 + (long long)ST_THREADS * u;
      dbv[u] = m[k < n ? k : n - 1];
    }
#pragma unroll
    for (int u = 0; u < ST_U; ++u) {
      const long long k = k0 + (long long)ST_THREADS * u;
      if (k >= n) continue;
      const float f = (float)((double)k * val);
      const float d = fabsf(f - probe_hz);
      const unsigned long long nk = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(uint32_t)k;
      near = nk < near ? nk : near;
      if (f >= f_min && f <= f_max) {
        const float db = dbv[u];
        uint32_t uu = __float_as_uint(db);
        uu = (uu & 0x80000000u) ? ~uu : (uu | 0x80000000u);  // monotone map float -> uint
        const unsigned long long key = ((unsigned long long)uu << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)k);
        best = key > best ? key : best;
        // 10^(dB/20): table-driven (ira_log.h) for finite values in the table's range -- the float64 exp10 was 70 of this
        // kernel's 100 instructions per bin, and the kernel is VALU-bound (one workgroup per spectrum); NaN / infinite / huge
        // values take the library routine
        const double y = (double)db * 0.05;
        const double lin = fabs(y) < 15.0 ? ira::exp10_table(y, etab) : exp10(y);
        cnt += 1.0; sfl += (double)f * lin; sl += lin;
        first_in = k < first_in ? k : first_in;
      }
    }
  }
  cnt = ira::wave_sum(cnt); sfl = ira::wave_sum(sfl); sl = ira::wave_sum(sl);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long ob = __shfl_xor(best, o, 64);
    best = ob > best ? ob : best;
    const unsigned long long on = __shfl_xor(near, o, 64);
    near = on < near ? on : near;
    const long long of = __shfl_xor(first_in, o, 64);
    first_in = of < first_in ? of : first_in;
  }
  if (lane == 0) {
    red[0][wave] = cnt; red[1][wave] = sfl; red[2][wave] = sl;
    kred[0][wave] = best; kred[1][wave] = near; lred[wave] = first_in;
  }
  __syncthreads();
  if (t == 0) {
    cnt = sfl = sl = 0.0;
    for (int w = 0; w < ST_THREADS / IRA_WAVE; ++w) {
      cnt += red[0][w]; sfl += red[1][w]; sl += red[2][w];
      best = kred[0][w] > best ? kred[0][w] : best;
      near = kred[1][w] < near ? kred[1][w] : near;
      first_in = lred[w] < first_in ? lred[w] : first_in;
    }
    double* o = out + (int64_t)e * 8;
    const long long pk = (cnt > 0.0) ? (long long)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull)) : 0;
    const long long i1k = (long long)(near & 0xFFFFFFFFull);
    o[0] = cnt;
    o[1] = (double)pk;
    o[2] = (double)(float)((double)pk * val);
    o[3] = sfl; o[4] = sl;
    o[5] = first_in < n ? (double)(float)((double)first_in * val) : 0.0;
    o[6] = (double)i1k;
    o[7] = (double)m[i1k];
  }
}

// ---- optional log-frequency smoothing of a dB curve (reference waterfall.py:140-185, frequency_response.py:117-169; default
// off).  Per curve: linear interpolation of the selected bins onto a uniform log2(f) grid (numpy.interp), a moving average of
// `window` grid points (numpy.convolve(.., ones(w)/w, "same": zero beyond the ends), linear interpolation back, float32.
// through_f32: the waterfall module round-trips the gridded curve through float32 around the convolution.
// One workgroup per curve; the grid (count <= LS_MAX points) lives in LDS.  The bins of a curve are `stride` elements apart
// (1 for a spectrum, the slice count for a column of an (F, S) matrix); frequencies are float32(k * fstep) as everywhere.
constexpr int LS_MAX = 2048;
constexpr int LS_THREADS = 256;

__device__ __forceinline__ double ls_freq(long long k, double fstep) { return (double)(float)((double)k * fstep); }

__global__ __launch_bounds__(LS_THREADS) void log_smooth_kernel(float* __restrict__ mag, const int64_t* __restrict__ off,
                                                                const int32_t* __restrict__ stride_a,
                                                                const int32_t* __restrict__ k_lo_a,
                                                                const int32_t* __restrict__ nsel_a,
                                                                const double* __restrict__ fstep_a,
                                                                const double* __restrict__ lg_lo, const double* __restrict__ lg_hi,
                                                                const int32_t* __restrict__ count_a, int window,
                                                                int through_f32) {
  __shared__ double xg[LS_MAX], on[LS_MAX], sm[LS_MAX];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int count = count_a[c], nsel = nsel_a[c], k_lo = k_lo_a[c];
  const long long stride = stride_a[c];
  const double fstep = fstep_a[c], a = lg_lo[c], b = lg_hi[c];
  if (count < 2 || count > LS_MAX || nsel < 1) return;
  float* m = mag + off[c];
  const double step = (b - a) / (double)(count - 1);
  // ---- onto the grid: numpy.interp(2 ** linspace(a, b, count), fs, ms) ----------------------------------------------
  for (int g = tid; g < count; g += LS_THREADS) {
    const double y = (g == count - 1) ? b : (double)g * step + a;
    const double x = exp2(y);
    xg[g] = x;
    double v;
    const double f0 = ls_freq(k_lo, fstep), fl = ls_freq((long long)k_lo + nsel - 1, fstep);
    if (nsel == 1 || x <= f0) v = (double)m[(long long)k_lo * stride];
    else if (x >= fl) v = (double)m[(long long)(k_lo + nsel - 1) * stride];
    else {
      int lo = 0, hi = nsel - 1;                             // largest j with fs[j] <= x
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ls_freq((long long)k_lo + mid, fstep) <= x) lo = mid; else hi = mid;
      }
      const double x0 = ls_freq((long long)k_lo + lo, fstep), x1 = ls_freq((long long)k_lo + lo + 1, fstep);
      const double y0 = (double)m[(long long)(k_lo + lo) * stride], y1 = (double)m[(long long)(k_lo + lo + 1) * stride];
      const double slope = (y1 - y0) / (x1 - x0);
      v = slope * (x - x0) + y0;
    }
    on[g] = through_f32 ? (double)(float)v : v;
  }
  __syncthreads();
  // ---- moving average, "same" -------------------------------------------------------------------------------------------
  const double inv_w = 1.0 / (double)window;
  const int h = (window - 1) / 2;
  for (int g = tid; g < count; g += LS_THREADS) {
    int j0 = g + h - (window - 1), j1 = g + h;
    if (j0 < 0) j0 = 0;
    if (j1 > count - 1) j1 = count - 1;
    double acc = 0.0;
    for (int j = j0; j <= j1; ++j) acc += on[j] * inv_w;
    sm[g] = through_f32 ? (double)(float)acc : acc;
  }
  __syncthreads();
  // ---- back onto the bins: numpy.interp(fs, grid, smoothed) -> float32 ------------------------------------------------
  for (int i = tid; i < nsel; i += LS_THREADS) {
    const double x = ls_freq((long long)k_lo + i, fstep);
    double v;
    if (x <= xg[0]) v = sm[0];
    else if (x >= xg[count - 1]) v = sm[count - 1];
    else {
      int lo = 0, hi = count - 1;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (xg[mid] <= x) lo = mid; else hi = mid;
      }
      const double slope = (sm[lo + 1] - sm[lo]) / (xg[lo + 1] - xg[lo]);
      v = slope * (x - xg[lo]) + sm[lo];
    }
    m[(long long)(k_lo + i) * stride] = (float)v;
  }
}

}  // namespace

extern "C" int32_t ira_spectrum_mag_phase(const double* spec_dev, const int64_t* spec_off_dev, const int32_t* L_dev,
                                          int32_t nb, int32_t max_len, double floor_db, float* mag_db_dev,
                                          const int64_t* mag_off_dev, double* phase_dev,
                                          const int64_t* phase_off_dev, const int32_t* packed_dev, void* stream) {
  IRA_CHECK_PTR(spec_dev); IRA_CHECK_PTR(spec_off_dev); IRA_CHECK_PTR(L_dev); IRA_CHECK_PTR(mag_db_dev);
  IRA_CHECK_PTR(mag_off_dev);
  if (phase_dev != nullptr && phase_off_dev == nullptr) return IRA_E_NULL;
  if (nb <= 0) return nb == 0 ? IRA_OK : IRA_E_SIZE;
  const double floor_lin = std::pow(10.0, floor_db / 20.0);
  int blocks = (max_len / 2 + 1 + 256 * 16 - 1) / (256 * 16);      // 16 bins per thread: the LDS tables are built once per workgroup
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  mag_phase_kernel<<<dim3(blocks, nb), 256, 0, (hipStream_t)stream>>>(
      reinterpret_cast<const cd*>(spec_dev), spec_off_dev, L_dev, floor_lin, mag_db_dev, mag_off_dev, phase_dev,
      phase_off_dev, packed_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_phase_unwrap(const double* phase_dev, const int64_t* phase_off_dev, const int32_t* L_dev,
                                    int32_t nb, int32_t do_unwrap, int32_t to_degrees, float* out_dev,
                                    const int64_t* out_off_dev, double* out64_dev, void* stream) {
  IRA_CHECK_PTR(phase_dev); IRA_CHECK_PTR(phase_off_dev); IRA_CHECK_PTR(L_dev); IRA_CHECK_PTR(out_off_dev);
  if (out_dev == nullptr && out64_dev == nullptr) return IRA_E_NULL;
  if (nb <= 0) return nb == 0 ? IRA_OK : IRA_E_SIZE;
  const double scale = to_degrees ? (180.0 / kPi) : 1.0;
  unwrap_kernel<<<nb, UW_THREADS, 0, (hipStream_t)stream>>>(phase_dev, phase_off_dev, L_dev, do_unwrap, scale,
                                                            out_dev, out_off_dev, out64_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_group_delay(const double* phase_dev, const int64_t* off_dev, const int32_t* nbins_dev,
                                   int32_t nb, int32_t max_bins, const double* bin_step_dev, double sample_rate_hz,
                                   int32_t* flags_dev, int32_t flags_known, double* gd_dev, void* stream) {
  IRA_CHECK_PTR(phase_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(nbins_dev); IRA_CHECK_PTR(bin_step_dev);
  IRA_CHECK_PTR(flags_dev); IRA_CHECK_PTR(gd_dev);
  if (nb <= 0 || max_bins <= 0) return (nb == 0 || max_bins == 0) ? IRA_OK : IRA_E_SIZE;
  if (nb > 65535 || !(sample_rate_hz > 0.0)) return IRA_E_SIZE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((max_bins + 255) / 256, nb);
  if (!flags_known) {
    // which formula numpy.gradient takes is a function of (bins, bin step, sample rate) alone: a caller that has decided
    // it once per distinct transform length passes the answers in flags_dev (flags_known) and this sweep is skipped
    hipError_t e = hipMemsetAsync(flags_dev, 0, sizeof(int32_t) * (size_t)nb, st);
    if (e != hipSuccess) return ira_hip_status(e);
    gd_uniform_kernel<<<grid, 256, 0, st>>>(nbins_dev, bin_step_dev, sample_rate_hz, flags_dev);
  }
  gd_gradient_kernel<<<grid, 256, 0, st>>>(phase_dev, off_dev, nbins_dev, bin_step_dev, sample_rate_hz, flags_dev,
                                           gd_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_spectrum_stats(const float* mag_db_dev, const int64_t* mag_off_dev, const int32_t* L_dev,
                                      int32_t nb, const double* freq_val_dev, double f_min_hz, double f_max_hz,
                                      double probe_hz, double* out_dev, void* stream) {
  IRA_CHECK_PTR(mag_db_dev); IRA_CHECK_PTR(mag_off_dev); IRA_CHECK_PTR(L_dev); IRA_CHECK_PTR(freq_val_dev);
  IRA_CHECK_PTR(out_dev);
  if (nb <= 0) return nb == 0 ? IRA_OK : IRA_E_SIZE;
  stats_kernel<<<nb, ST_THREADS, 0, (hipStream_t)stream>>>(mag_db_dev, mag_off_dev, L_dev, freq_val_dev,
                                                           (float)f_min_hz, (float)f_max_hz, (float)probe_hz, out_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_order_stats(const double* values_dev, const int64_t* off_dev, const int32_t* count_dev,
                                   int32_t nseg, const int64_t* ranks_dev, int32_t nranks, double* out_dev,
                                   void* stream) {
  IRA_CHECK_PTR(values_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(count_dev); IRA_CHECK_PTR(ranks_dev);
  IRA_CHECK_PTR(out_dev);
  if (nranks < 1 || nranks > OS_MAX_RANKS) return IRA_E_SIZE;
  if (nseg <= 0) return nseg == 0 ? IRA_OK : IRA_E_SIZE;
  order_stats_kernel<<<nseg, OS_THREADS, 0, (hipStream_t)stream>>>(values_dev, off_dev, count_dev, ranks_dev, nranks,
                                                                    out_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_log_smooth_db(float* mag_dev, const int64_t* off_dev, const int32_t* stride_dev,
                                     const int32_t* k_lo_dev, const int32_t* nsel_dev, const double* fstep_dev,
                                     const double* log2_lo_dev, const double* log2_hi_dev, const int32_t* count_dev,
                                     int32_t ncurves, int32_t max_count, int32_t window, int32_t through_float32,
                                     void* stream) {
  IRA_CHECK_PTR(mag_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(stride_dev); IRA_CHECK_PTR(k_lo_dev); IRA_CHECK_PTR(nsel_dev);
  IRA_CHECK_PTR(fstep_dev); IRA_CHECK_PTR(log2_lo_dev); IRA_CHECK_PTR(log2_hi_dev); IRA_CHECK_PTR(count_dev);
  if (ncurves <= 0) return ncurves == 0 ? IRA_OK : IRA_E_SIZE;
  if (window < 1) return IRA_E_SIZE;
  if (max_count > LS_MAX) return IRA_E_UNSUPPORTED;              // grid does not fit the workgroup's LDS
  log_smooth_kernel<<<ncurves, LS_THREADS, 0, (hipStream_t)stream>>>(mag_dev, off_dev, stride_dev, k_lo_dev, nsel_dev,
                                                                     fstep_dev, log2_lo_dev, log2_hi_dev, count_dev, window,
                                                                     through_float32 ? 1 : 0);
  IRA_RETURN_LAUNCH();
}
