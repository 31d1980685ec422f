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

// float64 10^y for |y| <= ~15 (dB / 20 of a float32 dB value) from a 64-entry table of 2^(j/64) in LDS:
// 10^y = 2^z, z = y log2(10) carried as hi + lo (the product's rounding error and the constant's second word: a one-word
// z would cost |z| 2^-53 ln 2 = 2e-15 relative at -120 dB); z = k/64 + r, |r| <= 1/128, 2^r by the exponential series in
// r ln 2 to the fifth power (next term 3.5e-17).  ~25 instructions instead of ~70 for the library exp10; agrees with it to
// 2-3 units in the last place.
constexpr int EXPTAB_N = 64;

__device__ __forceinline__ void build_exp_table(double* tab, int tid) {
  if (tid < EXPTAB_N) tab[tid] = exp2((double)tid / (double)EXPTAB_N);
}

__device__ __forceinline__ double exp10_table(double y, const double* tab) {
  constexpr double L_HI = 3.3219280948873622, L_LO = 1.6616175169735920e-16;     // log2(10) in two words
  const double zh = y * L_HI;
  const double zl = fma(y, L_HI, -zh) + y * L_LO;
  const double kf = rint(zh * (double)EXPTAB_N);
  const int k = (int)kf;
  const double r = fma(kf, -1.0 / (double)EXPTAB_N, zh) + zl;                     // exact difference + the low word
  const double p = r * 0.69314718055994531;
  double s = fma(p, 1.0 / 120.0, 1.0 / 24.0);
  s = fma(p, s, 1.0 / 6.0);
  s = fma(p, s, 0.5);
  s = fma(p, s, 1.0);
  const double t = tab[k & (EXPTAB_N - 1)];
  return ldexp(fma(t * p, s, t), k >> 6);                                        // 2^(k/64) (1 + p s)
}

}  // namespace ira
