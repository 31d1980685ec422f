// In-register building blocks shared by the register-resident STFT (ira_stft2.hip) and the radix-16 passes of the
// in-LDS FFT (ira_fft_lds.h): constant twiddles, R-point decimation-in-frequency DFTs on register arrays, and a
// shallow power tree for per-butterfly twiddles.
#pragma once
#include "ira_common.h"

namespace ira {

template <typename T>
struct Wc {  // constants in the working precision
  static constexpr T c1 = (T)0.92387953251128675612818318939679;
  static constexpr T s1 = (T)0.38268343236508977172845998403040;
  static constexpr T h = (T)0.70710678118654752440084436210485;
};

// v * W16^j  (W16 = exp(-2 pi i / 16)), j compile-time after unrolling.
// NOT recursive on purpose: a self-recursive helper cannot be inlined and every twiddle became an s_swappc call.
template <typename T>
__device__ __forceinline__ cplx<T> mul_w16(cplx<T> v, int j) {
  const T c1 = Wc<T>::c1, s1 = Wc<T>::s1, h = Wc<T>::h;
  cplx<T> t;
  switch (j & 7) {
    case 0: t = v; break;
    case 1: t = {v.re * c1 + v.im * s1, v.im * c1 - v.re * s1}; break;
    case 2: t = {(v.re + v.im) * h, (v.im - v.re) * h}; break;
    case 3: t = {v.re * s1 + v.im * c1, v.im * s1 - v.re * c1}; break;
    case 4: t = {v.im, -v.re}; break;
    case 5: t = {v.im * c1 - v.re * s1, -v.re * c1 - v.im * s1}; break;
    case 6: t = {(v.im - v.re) * h, -(v.re + v.im) * h}; break;
    default: t = {v.im * s1 - v.re * c1, -v.re * s1 - v.im * c1}; break;
  }
  if (j & 8) t = {-t.re, -t.im};   // W16^(j+8) = -W16^j
  return t;
}

__host__ __device__ constexpr int brev_bits(int k, int bits) {
  int r = 0;
  for (int b = 0; b < bits; ++b) r |= ((k >> b) & 1) << (bits - 1 - b);
  return r;
}

// In-register R-point DFT, decimation in frequency; result X[k] is left at v[brev(k)].
template <typename T, int R>
__device__ __forceinline__ void dft_dif(cplx<T> (&v)[R]) {
  constexpr int LOG = (R == 16) ? 4 : 3;
#pragma unroll
  for (int s = 0; s < LOG; ++s) {
    const int span = R >> s, half = span >> 1;
#pragma unroll
    for (int b = 0; b < R; b += span) {
#pragma unroll
      for (int j = 0; j < half; ++j) {
        const cplx<T> a = v[b + j], c = v[b + j + half];
        v[b + j] = {a.re + c.re, a.im + c.im};
        const cplx<T> d = {a.re - c.re, a.im - c.im};
        v[b + j + half] = mul_w16<T>(d, j * (16 / span));   // W_span^j = W16^(j*16/span)
      }
    }
  }
}

// w^k for k = 0..15 from w (tree of depth <= 4 multiplications to limit rounding growth)
template <typename T>
__device__ __forceinline__ void powers16(cplx<T> w, cplx<T> (&p)[16]) {
  p[0] = {(T)1, (T)0};
  p[1] = w;
  p[2] = ira::cmul(w, w);
  p[3] = ira::cmul(p[2], w);
  p[4] = ira::cmul(p[2], p[2]);
  p[5] = ira::cmul(p[4], w);
  p[6] = ira::cmul(p[4], p[2]);
  p[7] = ira::cmul(p[4], p[3]);
  p[8] = ira::cmul(p[4], p[4]);
#pragma unroll
  for (int k = 9; k < 16; ++k) p[k] = ira::cmul(p[8], p[k - 8]);
}

// v[idx(k)] *= w^k for k = 1..15, idx = bit reversal (BREV: v[i] holds index brev4(i), the order dft_dif leaves) or
// identity.  Same products as powers16, but only w^1..w^7 and w^8 are ever live together: 36 registers of
// twiddles instead of 64 beside the 64 of v in float64, which is what keeps the radix-16 LDS passes at 4 waves/SIMD.
template <typename T, bool BREV>
__device__ __forceinline__ void twiddle16(cplx<T> (&v)[16], cplx<T> w) {
  cplx<T> p[8];
  p[1] = w;
  p[2] = ira::cmul(w, w);
  p[3] = ira::cmul(p[2], w);
  p[4] = ira::cmul(p[2], p[2]);
  p[5] = ira::cmul(p[4], w);
  p[6] = ira::cmul(p[4], p[2]);
  p[7] = ira::cmul(p[4], p[3]);
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const int i = BREV ? brev_bits(k, 4) : k;
    v[i] = ira::cmul(v[i], p[k]);
  }
  const cplx<T> w8 = ira::cmul(p[4], p[4]);
  {
    const int i = BREV ? brev_bits(8, 4) : 8;
    v[i] = ira::cmul(v[i], w8);
  }
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const int i = BREV ? brev_bits(8 + k, 4) : 8 + k;
    v[i] = ira::cmul(v[i], ira::cmul(w8, p[k]));
  }
}

// The same for 8-point butterflies: v[idx(k)] *= w^k, k = 1..7, idx = 3-bit reversal or identity.
template <typename T, bool BREV>
__device__ __forceinline__ void twiddle8(cplx<T> (&v)[8], cplx<T> w) {
  const cplx<T> w2 = ira::cmul(w, w);
  const cplx<T> w3 = ira::cmul(w2, w);
  const cplx<T> w4 = ira::cmul(w2, w2);
  const cplx<T> p[8] = {{(T)1, (T)0}, w, w2, w3, w4, ira::cmul(w4, w), ira::cmul(w4, w2), ira::cmul(w4, w3)};
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const int i = BREV ? brev_bits(k, 3) : k;
    v[i] = ira::cmul(v[i], p[k]);
  }
}

// radix selected by its log2 (3 or 4)
template <typename T, int LR, bool BREV>
__device__ __forceinline__ void twiddle_r(cplx<T> (&v)[1 << LR], cplx<T> w) {
  if constexpr (LR == 4) twiddle16<T, BREV>(v, w);
  else twiddle8<T, BREV>(v, w);
}

}  // namespace ira
