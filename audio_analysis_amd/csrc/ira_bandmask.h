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

}  // namespace ira
