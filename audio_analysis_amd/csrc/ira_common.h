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

// Tuning / ablation knobs (IRA_* environment variables) exist only in the TUNING build (-DIRA_TUNING_BUILD:
// `python -m audio_analysis_amd.build --tuning` -> csrc/libira_tuning.so, never loaded by the product).  In the product
// library these helpers are constants: it reads no environment variable and keeps no state between calls -- every entry
// point is a pure function of its arguments.
#ifdef IRA_TUNING_BUILD
#include <cstdlib>
static inline int ira_tune_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : dflt;
}
static inline const char* ira_tune_str(const char* name) { return std::getenv(name); }
static inline bool ira_tune_flag(const char* name) { return std::getenv(name) != nullptr; }
#else
static inline constexpr int ira_tune_int(const char*, int dflt) { return dflt; }
static inline constexpr const char* ira_tune_str(const char*) { return nullptr; }
static inline constexpr bool ira_tune_flag(const char*) { return false; }
#endif

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
