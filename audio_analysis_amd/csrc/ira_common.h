// Shared device/host helpers for libira (gfx950 only; wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ira.h"

#define IRA_WAVE 64

#define IRA_CHECK_PTR(p)      \
  do {                        \
    if ((p) == nullptr) return IRA_E_NULL; \
  } while (0)

static inline int32_t ira_hip_status(hipError_t e) {
  return e == hipSuccess ? IRA_OK : (int32_t)(IRA_E_HIP_BASE - (int32_t)e);
}

// Launch epilogue: report launch-configuration errors without synchronising.
#define IRA_RETURN_LAUNCH() return ira_hip_status(hipGetLastError())

namespace ira {

template <typename T>
struct cplx {
  T re, im;
};

template <typename T>
__device__ __forceinline__ cplx<T> cadd(cplx<T> a, cplx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T>
__device__ __forceinline__ cplx<T> csub(cplx<T> a, cplx<T> b) { return {a.re - b.re, a.im - b.im}; }
template <typename T>
__device__ __forceinline__ cplx<T> cmul(cplx<T> a, cplx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename T>
__device__ __forceinline__ cplx<T> cconj(cplx<T> a) { return {a.re, -a.im}; }
// multiply by -i
template <typename T>
__device__ __forceinline__ cplx<T> cmul_mi(cplx<T> a) { return {a.im, -a.re}; }
// multiply by +i
template <typename T>
__device__ __forceinline__ cplx<T> cmul_pi(cplx<T> a) { return {-a.im, a.re}; }

// ---- wave-level reductions (64 lanes) ---------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace ira
