// In-LDS power-of-two complex FFTs shared by the STFT and long-FFT kernels.
//
//  lds_fft_dif : decimation-in-frequency, natural-order input -> BIT-REVERSED output (forward, e^{-i...}).
//  lds_fft_dit : decimation-in-time, BIT-REVERSED input -> natural-order output; with conj_tw it is the
//                (unnormalised) inverse of lds_fft_dif, so a forward/inverse pair needs no reorder pass.
// Radix-2^LR passes, LR = 4 (default) or 3 (16- or 8-point DFT in registers, ONE twiddle lookup per butterfly, powers by
// a shallow tree) while at least LR bits remain, then radix-4 and radix-2 for the remainder; in place;
// (LR = 3 costs a pass more at 1024 points but needs ~40 registers less: the long-FFT kernels trade it for occupancy) `nbat` independent transforms laid out
// `bstride` elements apart are processed together (all `nt` threads must call; the functions contain barriers).
//
// Twiddles come from a caller-provided table tw[k] = exp(-2*pi*i*k/TWN), k < TWN/2, TWN = M * tw_per_m;
// angles in [pi, 2pi) use W^(k+TWN/2) = -W^k.
#pragma once
#include "ira_fft_reg.h"

namespace ira {

__device__ __forceinline__ unsigned lds_brev(unsigned k, int log2m) {
  return log2m == 0 ? 0u : (__brev(k) >> (32 - log2m));
}

// Twiddle source of the passes.  SPLIT = false: `tw` is the table itself (global memory: one DEPENDENT load per butterfly
// and pass -- a memory round trip on the critical path of every pass).  SPLIT = true: `tw` points at a two-level copy in
// LDS, built once per workgroup (tw_split_fetch / tw_split_put): TW_COARSE entries W^(32 i) followed by 32 entries W^j;
// W^t = coarse[t >> 5] * fine[t & 31], two LDS reads and one complex multiply.  Covers half <= 32 * TW_COARSE = 4096.
constexpr int TW_COARSE = 128;
constexpr int TW_SPLIT_ENTRIES = TW_COARSE + 32;

template <typename T, bool SPLIT>
__device__ __forceinline__ cplx<T> tw_get(const cplx<T>* __restrict__ tw, unsigned idx) {
  if constexpr (SPLIT) return cmul(tw[idx >> 5], tw[TW_COARSE + (idx & 31u)]);
  else return tw[idx];
}

template <typename T, bool SPLIT = false>
__device__ __forceinline__ cplx<T> tw_lookup(const cplx<T>* __restrict__ tw, unsigned idx, unsigned half) {
  // idx in [0, 2*half)
  if (idx >= half) {
    const cplx<T> w = tw_get<T, SPLIT>(tw, idx - half);
    return {-w.re, -w.im};
  }
  return tw_get<T, SPLIT>(tw, idx);
}

// The split table in two halves, so that its global load can be issued BEFORE a tile's loads and its LDS write after them
// (one round trip for both).  `half` = entries of the source table that the passes use (M * tw_per_m / 2).
template <typename T>
__device__ __forceinline__ cplx<T> tw_split_fetch(const cplx<T>* __restrict__ tw, unsigned half, int tid) {
  const unsigned idx = tid < TW_COARSE ? 32u * (unsigned)tid : (unsigned)(tid - TW_COARSE);
  return (tid < TW_SPLIT_ENTRIES && idx < half) ? tw[idx] : cplx<T>{(T)1, (T)0};
}
template <typename T>
__device__ __forceinline__ void tw_split_put(cplx<T>* tab, cplx<T> v, int tid) {
  if (tid < TW_SPLIT_ENTRIES) tab[tid] = v;
}

template <typename T, int LR = 4, bool SPLIT = false>
__device__ __forceinline__ void lds_fft_dif(cplx<T>* buf, int log2m, const cplx<T>* __restrict__ tw,
                                            unsigned tw_per_m, int tid, int nt, int nbat = 1,
                                            unsigned bstride = 0) {
  constexpr int R = 1 << LR;
  const unsigned M = 1u << log2m;
  const unsigned half = (M * tw_per_m) >> 1;
  int s = log2m;  // log2 of the current block length S
  while (s >= LR) {
    // radix-R: elements base + r*q (r < R), q = S/R.  dft_dif leaves X[k] in v[brev(k)], and storing v[i] at
    // base + i*q is exactly the position the equivalent LR radix-2 stages would have used (bit-reversed order).
    const unsigned q = 1u << (s - LR);
    const unsigned step = tw_per_m << (log2m - s);
    const unsigned per = M >> LR;
    for (unsigned g = tid; g < per * (unsigned)nbat; g += nt) {
      const unsigned which = g >> (log2m - LR), b = g & (per - 1);
      cplx<T>* p = buf + which * bstride;
      const unsigned j = b & (q - 1);
      const unsigned base = ((b >> (s - LR)) << s) + j;
      cplx<T> v[R];
#pragma unroll
      for (int r = 0; r < R; ++r) v[r] = p[base + r * q];
      dft_dif<T, R>(v);
      if (j != 0) twiddle_r<T, LR, true>(v, tw_get<T, SPLIT>(tw, j * step));     // W_S^(j k), k < R  (j*step < half/(R/2))
#pragma unroll
      for (int i = 0; i < R; ++i) p[base + i * q] = v[i];
    }
    __syncthreads();
    s -= LR;
  }
  while (s >= 2) {
    const unsigned q = 1u << (s - 2);               // quarter block
    const unsigned step = tw_per_m << (log2m - s);  // table steps per unit of j at block length S
    const unsigned per = M >> 2;
    for (unsigned g = tid; g < per * (unsigned)nbat; g += nt) {
      const unsigned which = g >> (log2m - 2), b = g & (per - 1);
      cplx<T>* p = buf + which * bstride;
      const unsigned j = b & (q - 1);
      const unsigned base = ((b >> (s - 2)) << s) + j;
      const cplx<T> a0 = p[base], a1 = p[base + q], a2 = p[base + 2 * q], a3 = p[base + 3 * q];
      const cplx<T> p02 = cadd(a0, a2), m02 = csub(a0, a2);
      const cplx<T> p13 = cadd(a1, a3), m13 = cmul_mi(csub(a1, a3));  // -i (a1 - a3)
      cplx<T> c0 = cadd(p02, p13);
      cplx<T> c1 = csub(p02, p13);
      cplx<T> c2 = cadd(m02, m13);
      cplx<T> c3 = csub(m02, m13);
      if (j != 0) {
        const cplx<T> w1 = tw_get<T, SPLIT>(tw, j * step);                  // W_S^j      (j*step < half/2)
        const cplx<T> w2 = tw_get<T, SPLIT>(tw, 2 * j * step);              // W_S^(2j)   (< half)
        const cplx<T> w3 = tw_lookup<T, SPLIT>(tw, 3 * j * step, half);     // W_S^(3j)   (< 3/2 half)
        c1 = cmul(c1, w2);
        c2 = cmul(c2, w1);
        c3 = cmul(c3, w3);
      }
      p[base] = c0; p[base + q] = c1; p[base + 2 * q] = c2; p[base + 3 * q] = c3;
    }
    __syncthreads();
    s -= 2;
  }
  if (s == 1) {  // final radix-2 pass, block length 2, twiddle 1
    const unsigned per = M >> 1;
    for (unsigned g = tid; g < per * (unsigned)nbat; g += nt) {
      const unsigned which = g >> (log2m - 1), b = g & (per - 1);
      cplx<T>* p = buf + which * bstride;
      const cplx<T> a0 = p[2 * b], a1 = p[2 * b + 1];
      p[2 * b] = cadd(a0, a1);
      p[2 * b + 1] = csub(a0, a1);
    }
    __syncthreads();
  }
}

// Bit-reversed input -> natural output.  conj_tw = true uses exp(+i...) twiddles (inverse transform,
// unnormalised); false gives the forward transform of a bit-reversed-order input.
template <typename T, int LR = 4, bool SPLIT = false>
__device__ __forceinline__ void lds_fft_dit(cplx<T>* buf, int log2m, const cplx<T>* __restrict__ tw,
                                            unsigned tw_per_m, bool conj_tw, int tid, int nt, int nbat = 1,
                                            unsigned bstride = 0) {
  constexpr int R = 1 << LR;
  const unsigned M = 1u << log2m;
  const unsigned half = (M * tw_per_m) >> 1;
  const int rem = log2m % LR;   // low bits the radix-2 / radix-4 passes cover (mirror of the DIF tail); radix-R does the rest
  int s = 0;  // log2 of the block length already combined
  if (rem & 1) {  // first radix-2 pass, block length 2
    const unsigned per = M >> 1;
    for (unsigned g = tid; g < per * (unsigned)nbat; g += nt) {
      const unsigned which = g >> (log2m - 1), b = g & (per - 1);
      cplx<T>* p = buf + which * bstride;
      const cplx<T> a0 = p[2 * b], a1 = p[2 * b + 1];
      p[2 * b] = cadd(a0, a1);
      p[2 * b + 1] = csub(a0, a1);
    }
    __syncthreads();
    s = 1;
  }
  while (s < rem) {
    // combine blocks of length q = 2^s into blocks of length S = 4q
    const unsigned q = 1u << s;
    const int sS = s + 2;
    const unsigned step = tw_per_m << (log2m - sS);
    const unsigned per = M >> 2;
    for (unsigned g = tid; g < per * (unsigned)nbat; g += nt) {
      const unsigned which = g >> (log2m - 2), b = g & (per - 1);
      cplx<T>* p = buf + which * bstride;
      const unsigned j = b & (q - 1);
      const unsigned base = ((b >> s) << sS) + j;
      const cplx<T> a0 = p[base];
      cplx<T> t1 = p[base + q], t2 = p[base + 2 * q], t3 = p[base + 3 * q];
      if (j != 0) {
        cplx<T> w1 = tw_get<T, SPLIT>(tw, j * step);
        cplx<T> w2 = tw_get<T, SPLIT>(tw, 2 * j * step);
        cplx<T> w3 = tw_lookup<T, SPLIT>(tw, 3 * j * step, half);
        if (conj_tw) { w1.im = -w1.im; w2.im = -w2.im; w3.im = -w3.im; }
        t1 = cmul(t1, w2);
        t2 = cmul(t2, w1);
        t3 = cmul(t3, w3);
      }
      const cplx<T> s01 = cadd(a0, t1), d01 = csub(a0, t1);
      const cplx<T> s23 = cadd(t2, t3);
      const cplx<T> d23 = conj_tw ? cmul_pi(csub(t2, t3)) : cmul_mi(csub(t2, t3));  // (+-i)(t2 - t3)
      p[base] = cadd(s01, s23);
      p[base + q] = cadd(d01, d23);
      p[base + 2 * q] = csub(s01, s23);
      p[base + 3 * q] = csub(d01, d23);
    }
    __syncthreads();
    s = sS;
  }
  while (s < log2m) {
    // radix-R: the exact inverse of the DIF pass above.  Position base + i*q holds frequency index k = brev(i) of
    // the R-point stage; untwiddle, then the R-point (inverse) DFT puts sample n at base + n*q.
    const unsigned q = 1u << s;
    const int sS = s + LR;
    const unsigned step = tw_per_m << (log2m - sS);
    const unsigned per = M >> LR;
    for (unsigned g = tid; g < per * (unsigned)nbat; g += nt) {
      const unsigned which = g >> (log2m - LR), b = g & (per - 1);
      cplx<T>* p = buf + which * bstride;
      const unsigned j = b & (q - 1);
      const unsigned base = ((b >> s) << sS) + j;
      cplx<T> u[R];
#pragma unroll
      for (int k = 0; k < R; ++k) u[k] = p[base + brev_bits(k, LR) * q];
      if (j != 0) {
        cplx<T> w = tw_get<T, SPLIT>(tw, j * step);
        if (conj_tw) w.im = -w.im;
        twiddle_r<T, LR, false>(u, w);
      }
      if (conj_tw) {
        // inverse R-point DFT = conj(DFT(conj(.)))
#pragma unroll
        for (int k = 0; k < R; ++k) u[k].im = -u[k].im;
        dft_dif<T, R>(u);
#pragma unroll
        for (int n = 0; n < R; ++n) { cplx<T> r = u[brev_bits(n, LR)]; r.im = -r.im; p[base + n * q] = r; }
      } else {
        dft_dif<T, R>(u);
#pragma unroll
        for (int n = 0; n < R; ++n) p[base + n * q] = u[brev_bits(n, LR)];
      }
    }
    __syncthreads();
    s = sS;
  }
}

// ---- the same decimation-in-time transform on an INTERLEAVED tile -------------------------------------------------------------
// Layout [block][row][column]: nblk * 2^lc transforms of M points, element i of transform (blk, c) at
// buf[((blk << log2m) + i) << lc | c] -- the order in which a tile of 2^lc adjacent columns ARRIVES when its rows are
// fetched as contiguous pieces (LDS-DMA writes 64 lanes x 16 bytes contiguously, so a tile loaded by global_load_lds keeps
// its arrival order; ira_fftlong.hip, cols_inv_glds_kernel).  Neighbouring lanes take neighbouring columns of the same
// butterfly (contiguous 16-byte accesses).  Barriers are `lds_barrier()`: s_waitcnt lgkmcnt(0) + s_barrier, WITHOUT the
// vmcnt(0) of __syncthreads(), so that LDS-DMA loads of the next tile stay in flight across the passes.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int LR = 4, bool SPLIT = false>
__device__ __forceinline__ void lds_fft_dit_rc(cplx<T>* buf, int log2m, const cplx<T>* __restrict__ tw, unsigned tw_per_m,
                                               bool conj_tw, int tid, int nt, int nblk, int lc) {
  constexpr int R = 1 << LR;
  const unsigned M = 1u << log2m;
  const unsigned half = (M * tw_per_m) >> 1;
  const unsigned cm = (1u << lc) - 1u;
  const int rem = log2m % LR;
  int s = 0;
  if (rem & 1) {
    const unsigned per = M >> 1;
    for (unsigned g = tid; g < ((per * (unsigned)nblk) << lc); g += nt) {
      const unsigned c = g & cm, h = g >> lc, which = h >> (log2m - 1), b = h & (per - 1);
      cplx<T>* p = buf + ((which << log2m) << lc) + c;
      const cplx<T> a0 = p[(2 * b) << lc], a1 = p[(2 * b + 1) << lc];
      p[(2 * b) << lc] = cadd(a0, a1);
      p[(2 * b + 1) << lc] = csub(a0, a1);
    }
    lds_barrier();
    s = 1;
  }
  while (s < rem) {
    const unsigned q = 1u << s;
    const int sS = s + 2;
    const unsigned step = tw_per_m << (log2m - sS);
    const unsigned per = M >> 2;
    for (unsigned g = tid; g < ((per * (unsigned)nblk) << lc); g += nt) {
      const unsigned c = g & cm, h = g >> lc, which = h >> (log2m - 2), b = h & (per - 1);
      cplx<T>* p = buf + ((which << log2m) << lc) + c;
      const unsigned j = b & (q - 1);
      const unsigned base = ((b >> s) << sS) + j;
      const cplx<T> a0 = p[base << lc];
      cplx<T> t1 = p[(base + q) << lc], t2 = p[(base + 2 * q) << lc], t3 = p[(base + 3 * q) << lc];
      if (j != 0) {
        cplx<T> w1 = tw_get<T, SPLIT>(tw, j * step);
        cplx<T> w2 = tw_get<T, SPLIT>(tw, 2 * j * step);
        cplx<T> w3 = tw_lookup<T, SPLIT>(tw, 3 * j * step, half);
        if (conj_tw) { w1.im = -w1.im; w2.im = -w2.im; w3.im = -w3.im; }
        t1 = cmul(t1, w2);
        t2 = cmul(t2, w1);
        t3 = cmul(t3, w3);
      }
      const cplx<T> s01 = cadd(a0, t1), d01 = csub(a0, t1);
      const cplx<T> s23 = cadd(t2, t3);
      const cplx<T> d23 = conj_tw ? cmul_pi(csub(t2, t3)) : cmul_mi(csub(t2, t3));
      p[base << lc] = cadd(s01, s23);
      p[(base + q) << lc] = cadd(d01, d23);
      p[(base + 2 * q) << lc] = csub(s01, s23);
      p[(base + 3 * q) << lc] = csub(d01, d23);
    }
    lds_barrier();
    s = sS;
  }
  while (s < log2m) {
    const unsigned q = 1u << s;
    const int sS = s + LR;
    const unsigned step = tw_per_m << (log2m - sS);
    const unsigned per = M >> LR;
    for (unsigned g = tid; g < ((per * (unsigned)nblk) << lc); g += nt) {
      const unsigned c = g & cm, h = g >> lc, which = h >> (log2m - LR), b = h & (per - 1);
      cplx<T>* p = buf + ((which << log2m) << lc) + c;
      const unsigned j = b & (q - 1);
      const unsigned base = ((b >> s) << sS) + j;
      cplx<T> u[R];
#pragma unroll
      for (int k = 0; k < R; ++k) u[k] = p[(base + brev_bits(k, LR) * q) << lc];
      if (j != 0) {
        cplx<T> w = tw_get<T, SPLIT>(tw, j * step);
        if (conj_tw) w.im = -w.im;
        twiddle_r<T, LR, false>(u, w);
      }
      if (conj_tw) {
#pragma unroll
        for (int k = 0; k < R; ++k) u[k].im = -u[k].im;
        dft_dif<T, R>(u);
#pragma unroll
        for (int n = 0; n < R; ++n) { cplx<T> r = u[brev_bits(n, LR)]; r.im = -r.im; p[(base + n * q) << lc] = r; }
      } else {
        dft_dif<T, R>(u);
#pragma unroll
        for (int n = 0; n < R; ++n) p[(base + n * q) << lc] = u[brev_bits(n, LR)];
      }
    }
    lds_barrier();
    s = sS;
  }
}

}  // namespace ira
