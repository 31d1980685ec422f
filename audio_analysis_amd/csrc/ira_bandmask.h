// Band masks of the RT60 filter bank (reference analyse/rt60bands.py:116-167), evaluated on the device in the reference's
// float32 arithmetic on the float32 frequency axis.  Shared by the Bluestein (ira_fftlong.hip) and the direct
// (ira_fftsmooth.hip) inverse transforms.
#pragma once
#include "ira_common.h"

namespace ira {

constexpr double kPiMask = 3.14159265358979323846;

// cos(x) for 0 <= x <= pi in float64 (fdlibm's kernel polynomials after one Cody-Waite step by pi/2; error < 2 ulp of
// float64, far below the float32 rounding it feeds).  The library cos() carries its large-argument reduction along:
// inlined into the pass-1 input stage it cost 40 registers (121 instead of 79) and a wave per SIMD.
__device__ __forceinline__ double cos_0_pi(double x) {
  const double k = rint(x * 0.63661977236758134308);                       // 0, 1 or 2
  double r = fma(-k, 1.57079632679489655800e+00, x);
  r = fma(-k, 6.12323399573676603587e-17, r);
  const double z = r * r;
  const double c = 1.0 - (0.5 * z - z * (z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                   z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))))));
  const double sn = r + (z * r) * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                    z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
  return k == 0.0 ? c : (k == 1.0 ? -sn : -c);
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
  const float cs = (float)cos_0_pi((double)arg);  // float64 cosine rounded once: the correctly rounded float32 cosine
  return 0.5f - 0.5f * cs;
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
