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
// timing-only ablation tests inside kernels ("results are wrong by construction"): IRA_ABL(mask & bit) is the test in the
// tuning build and the constant 0 in the product library, which therefore carries no ablation path and pays no
// per-element test for one
#define IRA_ABL(expr) (expr)
#else
#define IRA_ABL(expr) 0
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

// ---- wave-uniform values -------------------------------------------------------------------------------------------------
// A per-job value read through a pointer the compiler cannot prove read-only (the job tables live beside the work arrays the
// kernel stores to) is fetched with a VECTOR load every time the source mentions it, each followed by s_waitcnt vmcnt(0) --
// which also waits for every tile load or store still in flight and turns one round of loads into one round trip per
// element.  Kernels therefore read their job's values ONCE, before the tile loops, through uniform(): the value moves to
// scalar registers and the loops contain no loads but the tile's own.
template <typename T>
__device__ __forceinline__ T uniform(T v) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit values");
  if constexpr (sizeof(T) == 4) {
    int i;
    __builtin_memcpy(&i, &v, 4);
    i = __builtin_amdgcn_readfirstlane(i);
    __builtin_memcpy(&v, &i, 4);
  } else {
    int w[2];
    __builtin_memcpy(w, &v, 8);
    w[0] = __builtin_amdgcn_readfirstlane(w[0]);
    w[1] = __builtin_amdgcn_readfirstlane(w[1]);
    __builtin_memcpy(&v, w, 8);
  }
  return v;
}

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
