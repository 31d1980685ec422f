// Table-based float64 log2 for the dB conversions (STFT magnitude, Schroeder EDC).
#pragma once
#include "ira_common.h"

namespace ira {

// float64 log2 of a positive normal number from a 128-entry table in LDS: p = 2^e * m, m in [1, 2);
// m = c_k (1 + r) with c_k the centre of the k-th of 128 mantissa intervals, |r| <= 2^-8;
// log2 p = e + log2 c_k + log1p(r) / ln 2, log1p by its series to r^6 (next term < 2e-18).  ~25 instructions
// instead of ~130 for hypot + log10 (which were two thirds of the float64 kernel's VALU work); the result
// differs from 20 log10(hypot) by a few 1e-16 relative, invisible after the rounding to float32.
struct LogTabEntry { double inv_c, log2_c; };
constexpr int LOGTAB_N = 128;

__device__ __forceinline__ void build_log_table(LogTabEntry* tab, int tid) {
  if (tid < LOGTAB_N) {
    const double inv_c = 1.0 / (1.0 + ((double)tid + 0.5) / (double)LOGTAB_N);
    tab[tid] = {inv_c, -log2(inv_c)};              // consistent with the ROUNDED reciprocal
  }
}

// TERMS = 6: log1p series to r^6 (next term < 2e-18); TERMS = 4: to r^4 (next term r^5/5 < 2e-13 relative to 1, i.e.
// < 1e-12 dB after the scaling to decibels -- for results that are rounded to float32 at once).
template <int TERMS = 6>
__device__ __forceinline__ double log2_table(double p, const LogTabEntry* tab) {
  const long long bits = __double_as_longlong(p);
  const int e = (int)((bits >> 52) & 0x7ff) - 1023;
  const int k = (int)((bits >> 45) & (LOGTAB_N - 1));
  const double m = __longlong_as_double((bits & 0x000fffffffffffffll) | 0x3ff0000000000000ll);
  const LogTabEntry t = tab[k];
  const double r = fma(m, t.inv_c, -1.0);
  double s;
  if constexpr (TERMS >= 6) {
    s = fma(r, -1.0 / 6.0, 0.2);
    s = fma(r, s, -0.25);
    s = fma(r, s, 1.0 / 3.0);
  } else {
    s = fma(r, -0.25, 1.0 / 3.0);
  }
  s = fma(r, s, -0.5);
  s = fma(r, s, 1.0);
  return (double)e + fma(r * s, 1.4426950408889634, t.log2_c);
}

}  // namespace ira
