// Band masks of the RT60 filter bank (reference analyse/rt60bands.py:116-167), evaluated on the device in the reference's
// float32 arithmetic on the float32 frequency axis.  Shared by the Bluestein (ira_fftlong.hip) and the direct
// (ira_fftsmooth.hip) inverse transforms.
#pragma once
#include "ira_common.h"

namespace ira {

constexpr double kPiMask = 3.14159265358979323846;

// numpy's float32 cosine, bit for bit.  The reference evaluates its raised-cosine ramps with numpy.cos on a float32 array
// (rt60bands.py:116-124), and numpy's float32 sin/cos is NOT libm's cosf nor the correctly rounded cosine: it is its own
// SIMD routine (numpy/_core/src/umath/loops_trigonometric: three-term Cody-Waite reduction by pi/2 with fused multiply-adds,
// then a degree-8 / degree-9 minimax polynomial in float32, documented at <= 1.49 ulp) -- 17 % of its results differ from
// the correctly rounded value by one ulp.  Same constants, same operation order, fmaf for every fused step (valid for
// |x| < 71476, the routine's own fast-path limit; the ramps only pass 0 <= x <= pi).  Pinned by the reference's own mask
// arrays: tests/golden/band_signals.npz, tests/test_gpu_longfft.py::test_band_masks_and_band_signals_vs_reference_goldens
// (the round-1 version computed a correctly rounded cosine and was off by one ulp in one transition bin out of six).
__device__ __forceinline__ float np_cos_f32(float x) {
  const float q = rintf(x * 0x1.45f306p-1f);                                  // quadrant: round-to-nearest-even of x * 2/pi
  float r = fmaf(q, -0x1.921fb0p+00f, x);
  r = fmaf(q, -0x1.5110b4p-22f, r);
  r = fmaf(q, -0x1.846988p-48f, r);
  const float x2 = r * r;
  float cp = fmaf(0x1.98e616p-16f, x2, -0x1.6c06dcp-10f);
  cp = fmaf(cp, x2, 0x1.55553cp-05f);
  cp = fmaf(cp, x2, -0x1.000000p-01f);
  cp = fmaf(cp, x2, 0x1.000000p+00f);
  float sp = fmaf(0x1.7d3bbcp-19f, x2, -0x1.a06bbap-13f);
  sp = fmaf(sp, x2, 0x1.11119ap-07f);
  sp = fmaf(sp, x2, -0x1.555556p-03f);
  sp = fmaf(sp, x2, 0.0f);
  sp = fmaf(sp, r, r);
  const int iq = (int)q + 1;                                                   // cos(x) = sin(x + pi/2): one quadrant on
  const float v = (iq & 1) ? cp : sp;
  return (iq & 2) ? -v : v;
}

struct BandMask {
  // kind: 0 zero mask, 1 low-pass, 2 high-pass, 3 band-pass (= hp * lp)
  double kind, hp_x0, hp_x1, lp_x0, lp_x1, pad0, pad1, pad2;
};

// The transition-band value 0.5 - 0.5 cos(pi t), 0 < t < 1, in the reference's float32 arithmetic.  Deliberately NOT
// inlined: only a few thousand bins per band lie inside a transition, and twenty inlined copies of the polynomial in the
// pass-1 input stage cost registers (and so resident workgroups) on every bin.
__device__ __noinline__ float ramp_inside(float t) {
  const float arg = (float)kPiMask * t;
  return 0.5f - 0.5f * np_cos_f32(arg);
}

__device__ __forceinline__ float ramp_f32(float f, double x0, double x1) {
  if (x1 <= x0) return f >= (float)x1 ? 1.0f : 0.0f;
  float t = (f - (float)x0) / (float)(x1 - x0);
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  // outside the transition band (almost every bin) the float32 formula below gives exactly 0 and 1:
  // cos(0) = 1 -> 0.5 - 0.5 = 0;  float32(cos(float32(pi))) = -1 -> 0.5 + 0.5 = 1.  Skip the float64 cosine there.
  if (t <= 0.0f) return 0.0f;
  if (t >= 1.0f) return 1.0f;
  return ramp_inside(t);
}
__device__ __forceinline__ float lowpass_f32(float f, double pass, double stop) {
  float m = 1.0f - ramp_f32(f, pass, stop);
  if (f <= (float)pass) m = 1.0f;
  if (f >= (float)stop) m = 0.0f;
  return m;
}
__device__ __forceinline__ float highpass_f32(float f, double stop, double pass) {
  float m = ramp_f32(f, stop, pass);
  if (f <= (float)stop) m = 0.0f;
  if (f >= (float)pass) m = 1.0f;
  return m;
}
__device__ __forceinline__ float mask_at(const BandMask& b, float f) {
  const int kind = (int)b.kind;
  if (kind == 1) return lowpass_f32(f, b.lp_x0, b.lp_x1);
  if (kind == 2) return highpass_f32(f, b.hp_x0, b.hp_x1);
  if (kind == 3) return highpass_f32(f, b.hp_x0, b.hp_x1) * lowpass_f32(f, b.lp_x0, b.lp_x1);
  return 0.0f;
}

// ---- the same masks without the per-bin float32 division ---------------------------------------------------------------
// Outside its transition bands (almost every bin) a mask is exactly 0 or 1, and WHICH bins those are follows from two
// float32 comparisons per edge, f(k) <= edge and f(k) >= edge with f(k) = float32(float64(k) * step) -- both monotone in k.
// So the first bin that passes each comparison is found once per wave (MaskCuts, eight lanes in parallel), and a bin then
// needs four integer compares; only bins inside a transition go through mask_at (the identical code as before, so the
// values are bit for bit the same).  The full formula was ~140 of the ~290 VALU instructions per element of the inverse
// pass-1 kernel, which is VALU-bound.
struct MaskCuts {
  int hp_a, hp_b;     // high-pass: k < hp_a -> 0,  k >= hp_b -> 1
  int lp_a, lp_b;     // low-pass:  k < lp_a -> 1,  k >= lp_b -> 0
};

__device__ __forceinline__ float bin_freq(int k, double step) { return (float)((double)k * step); }

// min { k in [0, kmax] : f(k) > c (strict) or f(k) >= c }, kmax + 1 if there is none; -1 if the walk from the estimate
// does not settle (the caller then sends every bin through the full formula)
__device__ __forceinline__ int first_bin(float c, bool strict, double step, int kmax) {
  if (!(fabsf(c) < 3.0e38f)) return c < 0.0f ? 0 : kmax + 1;                   // infinite edge (NaN never gets here)
  const double est = (double)c / step;
  if (est >= (double)kmax + 2.0) return kmax + 1;
  int k = est > 3.0 ? (int)est - 3 : 0;
  for (int it = 0; it < 10; ++it) {
    if (k > kmax) return kmax + 1;
    const float f = bin_freq(k, step);
    if (strict ? f > c : f >= c) return (it == 0 && k > 0) ? -1 : k;            // true at the start: estimate too high
    ++k;
  }
  return -1;
}

// Wave-cooperative (lanes 0..7, every lane gets the result): the cuts of the two bands of a job.
__device__ __forceinline__ void band_cuts(const BandMask& b1, const BandMask& b2, double step, int kmax, MaskCuts& c1,
                                          MaskCuts& c2) {
  const int lane = threadIdx.x & 63;
  const int which = lane & 3;
  const bool second = (lane & 4) != 0;
  const double hx0 = second ? b2.hp_x0 : b1.hp_x0, hx1 = second ? b2.hp_x1 : b1.hp_x1;
  const double lx0 = second ? b2.lp_x0 : b1.lp_x0, lx1 = second ? b2.lp_x1 : b1.lp_x1;
  const double x = which == 0 ? hx0 : which == 1 ? hx1 : which == 2 ? lx0 : lx1;
  // an edge pair that is not strictly ordered in float32 (or NaN) takes the full formula everywhere
  const bool degenerate = which < 2 ? !((float)hx0 < (float)hx1) : !((float)lx0 < (float)lx1);
  int v = degenerate ? -1 : first_bin((float)x, (which & 1) == 0, step, kmax);
  int cut[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) cut[l] = __builtin_amdgcn_readlane(v, l);
  auto fill = [](MaskCuts& c, const int* q) {
    const bool hp_ok = q[0] >= 0 && q[1] >= 0, lp_ok = q[2] >= 0 && q[3] >= 0;
    c.hp_a = hp_ok ? q[0] : 0;  c.hp_b = hp_ok ? q[1] : 0x7fffffff;
    c.lp_a = lp_ok ? q[2] : 0;  c.lp_b = lp_ok ? q[3] : 0x7fffffff;
  };
  fill(c1, cut);
  fill(c2, cut + 4);
}

// Bins outside [lo, hi) have mask 0 for certain (kind 0: nothing passes).
__device__ __forceinline__ void band_support(const BandMask& b, const MaskCuts& c, int& lo, int& hi) {
  const int kind = (int)b.kind;
  lo = 0; hi = 0;
  if (kind < 1 || kind > 3) return;
  lo = (kind & 2) ? c.hp_a : 0;
  hi = (kind & 1) ? c.lp_b : 0x7fffffff;
}

__device__ __forceinline__ float mask_cut(const BandMask& b, const MaskCuts& c, int k, double step) {
  const int kind = (int)b.kind;                          // uniform
  if (kind < 1 || kind > 3) return 0.0f;
  float hp = 1.0f, lp = 1.0f;
  bool inside = false;
  if (kind & 2) {
    if (k < c.hp_a) hp = 0.0f;
    else if (k < c.hp_b) inside = true;
  }
  if (kind & 1) {
    if (k >= c.lp_b) lp = 0.0f;
    else if (k >= c.lp_a) inside = true;
  }
  if (inside) return mask_at(b, bin_freq(k, step));
  return hp * lp;
}

}  // namespace ira
