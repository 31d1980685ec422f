// Band masks of the RT60 filter bank (reference analyse/rt60bands.py:116-167), evaluated on the device in the reference's
// float32 arithmetic on the float32 frequency axis.  Shared by the Bluestein (ira_fftlong.hip) and the direct
// (ira_fftsmooth.hip) inverse transforms.
#pragma once
#include "ira_common.h"

namespace ira {

constexpr double kPiMask = 3.14159265358979323846;

struct BandMask {
  // kind: 0 zero mask, 1 low-pass, 2 high-pass, 3 band-pass (= hp * lp)
  double kind, hp_x0, hp_x1, lp_x0, lp_x1, pad0, pad1, pad2;
};

__device__ __forceinline__ float ramp_f32(float f, double x0, double x1) {
  if (x1 <= x0) return f >= (float)x1 ? 1.0f : 0.0f;
  float t = (f - (float)x0) / (float)(x1 - x0);
  t = fminf(fmaxf(t, 0.0f), 1.0f);
  // outside the transition band (almost every bin) the float32 formula below gives exactly 0 and 1:
  // cos(0) = 1 -> 0.5 - 0.5 = 0;  float32(cos(float32(pi))) = -1 -> 0.5 + 0.5 = 1.  Skip the float64 cosine there.
  if (t <= 0.0f) return 0.0f;
  if (t >= 1.0f) return 1.0f;
  const float arg = (float)kPiMask * t;
  const float cs = (float)cos((double)arg);  // correctly rounded float32 cosine
  return 0.5f - 0.5f * cs;
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
