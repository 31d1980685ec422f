// STFT v4: float64 / n_fft = 8192 (the reference's modal-cloud default, modalcloud.py:121-158), frame-major output.
//
// v2 (ira_stft2.hip) keeps a whole 4096-point float64 exchange (64 KB + padding) per 2-wave team plus the transposing
// output tile: 143 KB per workgroup, ONE workgroup = 4 waves per CU.  v4 applies the two ideas of the float32 kernel
// (ira_stft3.hip) to this configuration:
//   * half-size LDS exchanges (33 KB per team) -> one team per workgroup, four workgroups = 8 waves per CU;
//   * frame-major (T, F) output -> every lane stores its 32 values of the frame directly, no tile, 512-byte runs.
// Transform: packed real FFT, z[n] = xw[2n] + i xw[2n+1], M = 4096 = 16 * 16 * 16, DIF, n = n1*256 + n2*16 + n3,
// k = k1 + 16 k2 + 256 k3, 128 lanes (q), 32 complex values per lane:
//   step 1  lane m = q + 128 h (h = 0, 1): 16-point DFT over n1 from global memory, twiddle W_M^(k1 m)
//   E1      half h at a time: [16 k1][128 m'] complex, row stride 129     -> (k1 = q & 15, n3 = (q >> 4) + 8 hb) reads n2
//   step 2  two 16-point DFTs over n2 (hb = 0, 1), twiddle W_M^(16 k2 n3)
//   E2      half hb at a time: k1 + 16 k2 + 256 n3' complex               -> row r = q + 128 hh = k1 + 16 k2 reads n3
//   step 3  two 16-point DFTs over n3 -> lane holds Z[r + 256 k3]
//   E3      natural order, real parts then imaginary parts through one 4096-double buffer
//   post    (Z[k], Z[M-k]) -> |X[k]|, |X[N/2-k]| in dB (table log2, ira_log.h), stored at out[t*F + k]
// 16-byte LDS accesses are served a quarter wave at a time: in every exchange the 16 lanes of a quarter differ in k1
// (or in consecutive m), which the strides 129 and 1 map to distinct 16-byte bank groups: conflict free.
#include <cmath>
#include <cstdlib>

#include "ira_fft_reg.h"
#include "ira_log.h"

namespace {

using ira::brev_bits;
using ira::cplx;
using ira::dft_dif;
using ira::powers16;

typedef cplx<double> cdd;

constexpr int M4 = 4096, F4 = M4 + 1, TL4 = 128;
constexpr int ROW4 = 129;                 // E1 half: row stride (complex)
constexpr int E2N4 = 256;                 // E2 half: n3' stride (complex)
constexpr int EXC4 = 16 * ROW4;           // complex slots per workgroup: max(16*129, 8*256, 4096 doubles / 2)
static_assert(EXC4 >= 8 * E2N4 && EXC4 * 2 >= M4, "exchange buffer too small");

__device__ __forceinline__ float db_of4(double re, double im, double floor_pow, float floor_db,
                                        const ira::LogTabEntry* tab) {
  // |X| <= 8192 * max|x| < 3e42 for float32 samples, so p < 1e85 is always a normal double unless the frame holds an
  // infinity or a NaN -- and those frames are flagged as a whole (bad_frame) and never reach this value.  (Round 2 carried
  // a hypot + log10 path for p >= 1e300 here: dead for this kernel's inputs, and 40 % of its code size.)
  const double p = fma(re, re, im * im);
  if (!(p > floor_pow)) return floor_db;                                       // also catches NaN
  return (float)(3.0102999566398120 * ira::log2_table<4>(p, tab));
}

// Linear magnitude the reference's aggregation starts from: 10^(float32(dB)/20) with dB = 20 log10 max(|X|, floor)
// (modalcloud.py:186-190 applied to the float32 STFT of :150-156).  With dB = 10 log10 p in float64 and d = float32(dB) - dB
// (|d| <= 4e-6 dB) this is sqrt(p) * 10^(d/20) = sqrt(p) (1 + t + t^2/2), t = d ln(10)/20 <= 5e-7 (next term 2e-20):
// a square root and two fused multiply-adds instead of a float64 exp10 (~50 instructions), same value to ~1e-15.
__device__ __forceinline__ double lin_of4(double re, double im, double floor_pow, float floor_db, double floor_lin32,
                                          const ira::LogTabEntry* tab) {
  const double p = fma(re, re, im * im);
  if (!(p > floor_pow)) return floor_lin32;                                    // also catches NaN, like db_of4
  const double db = 3.0102999566398120 * ira::log2_table<4>(p, tab);           // (p < 1e85: see db_of4)
  const double t = ((double)(float)db - db) * 0.11512925464970228;
  // sqrt(p) for a normal p (floor_pow < p < 1e85: no scaling needed): hardware reciprocal square root (~26 bits) and ONE
  // coupled Newton step (Goldschmidt form): relative error 1.5 e0^2 ~ 3e-16 -- the mean of these values is rounded to a
  // float32 dB value, 1e-12 would do (round 2 ran two steps).  6 instructions instead of the ~25 of the library sqrt.
  const double y0 = __builtin_amdgcn_rsq(p);
  double g = p * y0;
  const double h = 0.5 * y0;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  return g * fma(t, fma(t, 0.5, 1.0), 1.0);
}

__global__ __launch_bounds__(TL4) void stft4_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int32_t* __restrict__ nframes, int hop,
    const double* __restrict__ window, const cdd* __restrict__ tw, double floor_lin, float floor_db,
    float* __restrict__ out, const int64_t* __restrict__ out_off, const int32_t* __restrict__ frame_sel,
    const int64_t* __restrict__ sel_off, int lb_nbins, int lb_kbase, const int32_t* __restrict__ lb_first,
    const int32_t* __restrict__ lb_count) {
  __shared__ __attribute__((aligned(16))) cdd ex[EXC4];
  __shared__ ira::LogTabEntry ltab[ira::LOGTAB_N];
  // XCD-aware bijective remap: consecutive frames of a segment (which share 15/16 of their samples) meet in one L2
  const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
  const unsigned orig = blockIdx.y * gx + blockIdx.x;
  const unsigned xq = nwg / 8, xr = nwg % 8, xcd = orig % 8;
  const unsigned wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;
  const int seg = (int)(wg / gx);
  const int col = (int)(wg % gx);
  const int T_out = nframes[seg];
  if (col >= T_out) return;
  const int q = threadIdx.x;
  double* exd = reinterpret_cast<double*>(ex);
  ira::build_log_table(ltab, q);

  const int64_t frame = frame_sel ? (int64_t)frame_sel[sel_off[seg] + col] : (int64_t)col;
  const float* fx = x + off[seg] + frame * hop;
  const int k1l = q & 15, n3a = q >> 4;

  // ---- step 1 -------------------------------------------------------------------------------------------------
  cdd a1[16];   // half h = 1, held in registers until E1 is free again
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int m = q + TL4 * h;
    float xa[16], xb[16];
    double wa[16], wb[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int n = n1 * 256 + m;
      xa[n1] = fx[2 * n]; xb[n1] = fx[2 * n + 1];
      wa[n1] = window[2 * n]; wb[n1] = window[2 * n + 1];
    }
    __builtin_amdgcn_sched_barrier(0);
    cdd v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = {(double)xa[n1] * wa[n1], (double)xb[n1] * wb[n1]};
    dft_dif<double, 16>(v);
    cdd p[16];
    powers16<double>(tw[2 * m], p);                      // W_M^m = W_N^(2m)
    if (h == 0) {
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        const cdd a = v[brev_bits(k1, 4)];
        ex[k1 * ROW4 + q] = (k1 == 0) ? a : ira::cmul(a, p[k1]);
      }
    } else {
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        const cdd a = v[brev_bits(k1, 4)];
        a1[k1] = (k1 == 0) ? a : ira::cmul(a, p[k1]);
      }
    }
  }
  __syncthreads();

  // ---- E1 -> step-2 operands: n2 = 0..7 come from half 0, n2 = 8..15 from half 1 ------------------------------------
  cdd b2[2][16];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) b2[hb][n2] = ex[k1l * ROW4 + n2 * 16 + n3a + 8 * hb];
  __syncthreads();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) ex[k1 * ROW4 + q] = a1[k1];
  __syncthreads();
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) b2[hb][8 + n2] = ex[k1l * ROW4 + n2 * 16 + n3a + 8 * hb];
  __syncthreads();

  // ---- step 2 and E2 (half hb = n3 in [8hb, 8hb + 8)) -> step-3 operands ---------------------------------------------
  cdd z3[2][16];
  {
    cdd p[16];
    dft_dif<double, 16>(b2[0]);
    powers16<double>(tw[32 * n3a], p);                   // W_M^(16 n3) = W_N^(32 n3)
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      const cdd a = b2[0][brev_bits(k2, 4)];
      ex[k1l + 16 * k2 + E2N4 * n3a] = (k2 == 0) ? a : ira::cmul(a, p[k2]);
    }
    dft_dif<double, 16>(b2[1]);
    powers16<double>(tw[32 * (n3a + 8)], p);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) b2[1][brev_bits(k2, 4)] = ira::cmul(b2[1][brev_bits(k2, 4)], p[k2]);
  }
  __syncthreads();
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) z3[hh][n3] = ex[k1l + 16 * (n3a + 8 * hh) + E2N4 * n3];
  __syncthreads();
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) ex[k1l + 16 * k2 + E2N4 * n3a] = b2[1][brev_bits(k2, 4)];
  __syncthreads();
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) z3[hh][8 + n3] = ex[k1l + 16 * (n3a + 8 * hh) + E2N4 * n3];
  __syncthreads();

  // ---- step 3: lane holds Z[r + 256 k3], r = q + 128 hh -----------------------------------------------------------------
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) dft_dif<double, 16>(z3[hh]);

  // ---- E3: real parts, then imaginary parts, natural order ------------------------------------------------------------
  double zkr[16], zpr[16], zki[16], zpi[16], midr, midi;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) exd[q + TL4 * hh + 256 * k3] = z3[hh][brev_bits(k3, 4)].re;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = q + TL4 * i;
    zkr[i] = exd[k];
    zpr[i] = exd[(M4 - k) & (M4 - 1)];
  }
  midr = exd[M4 / 2];
  __syncthreads();
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) exd[q + TL4 * hh + 256 * k3] = z3[hh][brev_bits(k3, 4)].im;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = q + TL4 * i;
    zki[i] = exd[k];
    zpi[i] = exd[(M4 - k) & (M4 - 1)];
  }
  midi = exd[M4 / 2];

  // ---- post: X[k] = E + P, X[M-k] = conj(E - P) with E = (Zk + conj Zp)/2, P = W_N^k (-i)(Zk - conj Zp)/2 ---------------
  const double floor_pow = floor_lin * floor_lin;
  const cdd wlane = tw[q];
  // A NaN (or infinite) sample anywhere in the frame makes every bin of numpy's rfft NaN (modalcloud.py:150): it shows in
  // Z[0] = sum of the packed inputs (lane 0, first pair; 0 * NaN at the Hann end points is NaN too).  One check per frame.
  __shared__ int frame_bad;
  if (q == 0) frame_bad = !((zkr[0] - zkr[0]) + (zki[0] - zki[0]) == 0.0) ? 1 : 0;
  __syncthreads();
  const bool bad_frame = frame_bad != 0;
  const float qnan32 = __uint_as_float(0x7fc00000u);
  if (lb_nbins <= 0) {
    // ---- frame-major dB matrix ---------------------------------------------------------------------------------------
    float* fo = out + out_off[seg] + (int64_t)col * F4;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = q + TL4 * i;
      const cdd e = {0.5 * (zkr[i] + zpr[i]), 0.5 * (zki[i] - zpi[i])};
      const cdd d = {0.5 * (zkr[i] - zpr[i]), 0.5 * (zki[i] + zpi[i])};
      const cdd o = {d.im, -d.re};
      const cdd wk = ira::cmul(wlane, tw[TL4 * i]);        // W_N^k = W_N^q W_N^(128 i); second factor wave-uniform
      const cdd pp = ira::cmul(wk, o);
      fo[k] = bad_frame ? qnan32 : db_of4(e.re + pp.re, e.im + pp.im, floor_pow, floor_db, ltab);
      fo[M4 - k] = bad_frame ? qnan32 : db_of4(e.re - pp.re, e.im - pp.im, floor_pow, floor_db, ltab);   // k = 0 -> bin M
    }
    if (q == 0) fo[M4 / 2] = bad_frame ? qnan32 : db_of4(midr, midi, floor_pow, floor_db, ltab);
    return;
  }

  // ---- fused modal-cloud aggregation (reference modalcloud.py:176-207): the frame's dB values never leave the CU.
  // float32 dB (the reference's STFT output type) -> linear magnitude 10^(dB/20) in float64 -> LDS; then log bin b is
  // the mean of its rows, added in ascending order, -> 20 log10(max(., 1e-30)) -> float32 at out[b * T + frame].
  // Only the rows some log bin reads are converted: 20 Hz .. 20 kHz is rows 4 .. 3413 of 4097, a sixth of the
  // conversions (the costliest part of the frame) is skipped.
  __shared__ int lb_range[2];
  if (q < 2) lb_range[q] = q == 0 ? F4 : 0;
  __syncthreads();                                          // every lane has finished reading E3 (and sees lb_range)
  {
    int lo = F4, hi = 0;
    for (int b = q; b < lb_nbins; b += TL4) {
      const int c = lb_count[b];
      if (c > 0) {
        const int f0 = lb_kbase + lb_first[b];
        lo = f0 < lo ? f0 : lo;
        hi = f0 + c > hi ? f0 + c : hi;
      }
    }
    atomicMin(&lb_range[0], lo);
    atomicMax(&lb_range[1], hi);
  }
  __syncthreads();
  const int k_lo = lb_range[0], k_hi = lb_range[1];
  const double floor_lin32 = exp10((double)floor_db * 0.05);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = q + TL4 * i;
    const bool need_a = k >= k_lo && k < k_hi, need_b = (M4 - k) >= k_lo && (M4 - k) < k_hi;
    if (!need_a && !need_b) continue;
    const cdd e = {0.5 * (zkr[i] + zpr[i]), 0.5 * (zki[i] - zpi[i])};
    const cdd d = {0.5 * (zkr[i] - zpr[i]), 0.5 * (zki[i] + zpi[i])};
    const cdd o = {d.im, -d.re};
    const cdd wk = ira::cmul(wlane, tw[TL4 * i]);
    const cdd pp = ira::cmul(wk, o);
    if (need_a) exd[k] = lin_of4(e.re + pp.re, e.im + pp.im, floor_pow, floor_db, floor_lin32, ltab);
    if (need_b) exd[M4 - k] = lin_of4(e.re - pp.re, e.im - pp.im, floor_pow, floor_db, floor_lin32, ltab);
  }
  if (q == 0 && M4 / 2 >= k_lo && M4 / 2 < k_hi) exd[M4 / 2] = lin_of4(midr, midi, floor_pow, floor_db, floor_lin32, ltab);
  __syncthreads();
  float* co = out + out_off[seg];
  for (int b = q; b < lb_nbins; b += TL4) {
    const int c = lb_count[b];
    float v = __uint_as_float(0x7fc00000u);
    if (c > 0) {
      const double* r = exd + lb_kbase + lb_first[b];
      double acc = r[0];
      for (int k0 = 1; k0 < c; k0 += 8) {
        double v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = (k0 + u < c) ? r[k0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < c) acc += v8[u];
      }
      v = (float)(20.0 * log10(fmax(acc / (double)c, 1e-30)));
    }
    co[(int64_t)b * T_out + col] = (bad_frame && c > 0) ? qnan32 : v;      // mean of NaN magnitudes (modalcloud.py:186-200)
  }
}

// ------------------------------------------------------------------------------------------------------------
// STFT v5: the same transform with ONE FRAME ON FOUR WAVES (256 lanes, 16 complex values per lane).
// Counters of v4 (profiles/r02_stft4_counters.txt): 3624 VALU instructions per wave at ~4.2 cycles each (float64 issues
// over 4 cycles), VALU active 30 % of a wave's life, 34 % parked at waitcnt / barriers; with 2 waves per SIMD (228
// VGPRs, 35 KB of LDS per 2-wave workgroup) the SIMDs' float64 pipes are busy 60 % of the time.  The frame's LDS budget
// (one half-size exchange buffer) does not depend on how many lanes share it, the register budget does: 16 values per
// lane instead of 32 fit ~128 VGPRs, so the same four frames per CU now bring 16 waves instead of 8 and a wave that waits
// at a barrier has three others on its SIMD to cover for it.
//   step 1  lane m = q: 16-point DFT over n1 from global memory, twiddle W_M^(k1 m)
//   E1      lanes 0..127 write [16 k1][128 m'] (stride 129), all lanes read n2 = 0..7 at (k1 = q & 15, n3 = q >> 4);
//           then lanes 128..255 write and all read n2 = 8..15
//   step 2  16-point DFT over n2, twiddle W_M^(16 k2 n3)
//   E2      lanes with n3 < 8 (q < 128) write k1 + 16 k2 + 256 n3, lane r = q = k1 + 16 k2 reads n3 = 0..7; then n3 >= 8
//   step 3  16-point DFT over n3 -> lane r holds Z[r + 256 k3]
//   E3      natural order, real parts then imaginary parts through one 4096-double buffer; lane pairs k = q + 256 i, i < 8
// ------------------------------------------------------------------------------------------------------------
constexpr int TL5 = 256;

__global__ __launch_bounds__(TL5) __attribute__((amdgpu_waves_per_eu(4, 4))) void stft5_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int32_t* __restrict__ nframes, int hop,
    const double* __restrict__ window, const cdd* __restrict__ tw, double floor_lin, float floor_db,
    float* __restrict__ out, const int64_t* __restrict__ out_off, const int32_t* __restrict__ frame_sel,
    const int64_t* __restrict__ sel_off, int lb_nbins, int lb_kbase, const int32_t* __restrict__ lb_first,
    const int32_t* __restrict__ lb_count, int ablate) {
  // ablate (IRA_STFT5_ABLATE, tuning build, timing only): 1 no window loads, 2 no sample loads, 4 no dB -> linear conversion
  __shared__ __attribute__((aligned(16))) cdd ex[EXC4];
  __shared__ ira::LogTabEntry ltab[ira::LOGTAB_N];
  __shared__ int lb_range[2];
  __shared__ int frame_bad;
  const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
  const unsigned orig = blockIdx.y * gx + blockIdx.x;
  const unsigned xq = nwg / 8, xr = nwg % 8, xcd = orig % 8;
  const unsigned wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;
  const int seg = (int)(wg / gx);
  const int col = (int)(wg % gx);
  const int T_out = nframes[seg];
  if (col >= T_out) return;
  const int q = threadIdx.x;
  double* exd = reinterpret_cast<double*>(ex);
  ira::build_log_table(ltab, q);
  if (q < 2) lb_range[q] = q == 0 ? F4 : 0;

  const int64_t frame = frame_sel ? (int64_t)frame_sel[sel_off[seg] + col] : (int64_t)col;
  const float* fx = x + off[seg] + frame * hop;
  const int k1l = q & 15, n3l = q >> 4;            // step-2 role: (k1, n3)
  const bool lower = q < TL4;                      // lanes whose step-1 / step-2 results go through the buffer first
  // The eight wave-uniform factors W_N^(256 i) of the post step, requested HERE: they compile to scalar loads, and placed
  // at their use each one stalled its wave for a scalar-cache round trip (s_waitcnt lgkmcnt(0), which also drains the LDS
  // queue) in the middle of the conversion loop -- eight serial stalls per frame and wave.
  cdd wuni[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wuni[i] = tw[TL5 * i];

  // ---- step 1 -------------------------------------------------------------------------------------------------
  cdd v[16];
  {
    float xa[16], xb[16];
    double wa[16], wb[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int n = n1 * 256 + q;
      if (IRA_ABL(ablate & 2)) { xa[n1] = (float)(n & 7) * 0.125f; xb[n1] = (float)(q & 3); }
      else { xa[n1] = fx[2 * n]; xb[n1] = fx[2 * n + 1]; }
      if (IRA_ABL(ablate & 1)) { wa[n1] = 0.5 + 1e-4 * n1; wb[n1] = 0.25; }
      else { wa[n1] = window[2 * n]; wb[n1] = window[2 * n + 1]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = {(double)xa[n1] * wa[n1], (double)xb[n1] * wb[n1]};
  }
  dft_dif<double, 16>(v);
  ira::twiddle16<double, true>(v, tw[2 * q]);                 // W_M^(k1 q) = W_N^(2 q k1), k1 at v[brev(k1)]

  // ---- E1: half-size exchange, lower lanes first ----------------------------------------------------------------
  cdd b[16];
  if (lower) {
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) ex[k1 * ROW4 + q] = v[brev_bits(k1, 4)];
  }
  __syncthreads();
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) b[n2] = ex[k1l * ROW4 + n2 * 16 + n3l];
  __syncthreads();
  if (!lower) {
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) ex[k1 * ROW4 + (q - TL4)] = v[brev_bits(k1, 4)];
  }
  __syncthreads();
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) b[8 + n2] = ex[k1l * ROW4 + n2 * 16 + n3l];
  __syncthreads();

  // ---- step 2 and E2 ---------------------------------------------------------------------------------------------
  dft_dif<double, 16>(b);
  ira::twiddle16<double, true>(b, tw[32 * n3l]);              // W_M^(16 k2 n3) = W_N^(32 n3 k2)
  if (lower) {                                                 // n3 < 8
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) ex[k1l + 16 * k2 + E2N4 * n3l] = b[brev_bits(k2, 4)];
  }
  __syncthreads();
#pragma unroll
  for (int n3 = 0; n3 < 8; ++n3) v[n3] = ex[q + E2N4 * n3];
  __syncthreads();
  if (!lower) {
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) ex[k1l + 16 * k2 + E2N4 * (n3l - 8)] = b[brev_bits(k2, 4)];
  }
  __syncthreads();
#pragma unroll
  for (int n3 = 0; n3 < 8; ++n3) v[8 + n3] = ex[q + E2N4 * n3];
  __syncthreads();

  // ---- step 3: lane r = q holds Z[r + 256 k3] at v[brev(k3)] ----------------------------------------------------------
  dft_dif<double, 16>(v);

  // ---- E3: the mirror partners only.  The post step pairs Z[k] with Z[M - k]; lane q keeps k = q + 256 i, i < 8, which it
  // already holds (v[brev(i)]), and M - k = (256 - q) + 256 (15 - i) is entry k3 = 15 - i >= 8 of lane 256 - q: every lane
  // publishes its upper eight values and reads eight of its partner's -- 256 bytes per lane through LDS and one barrier pair
  // instead of the 512 bytes and two of the round-2 natural-order exchange (real parts, then imaginary parts, own values
  // included).  Lane 0 pairs with itself: M - 256 i = 256 (16 - i), its own entry 16 - i; i = 0 pairs Z[0] with Z[0].
  double zkr[8], zpr[8], zki[8], zpi[8], midr, midi;
#pragma unroll
  for (int j = 0; j < 8; ++j) ex[j * TL5 + q] = v[brev_bits(8 + j, 4)];       // k3 = 8 + j
  __syncthreads();
  {
    const int partner = (TL5 - q) & (TL5 - 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // partner entry k3 = 15 - i (slot 7 - i); lane 0: k3 = 16 - i (slot 8 - i), i = 0 -> Z[0] itself
      const int slot = q == 0 ? 8 - i : 7 - i;
      cdd zp = v[0];                                                             // Z[0] (lane 0, i = 0)
      if (!(q == 0 && i == 0)) zp = ex[slot * TL5 + partner];
      zkr[i] = v[brev_bits(i, 4)].re; zki[i] = v[brev_bits(i, 4)].im;
      zpr[i] = zp.re; zpi[i] = zp.im;
    }
  }
  {
    const cdd mid = ex[0];                                                       // Z[2048]: lane 0, k3 = 8
    midr = mid.re; midi = mid.im;
  }
  if (q == 0) frame_bad = !((zkr[0] - zkr[0]) + (zki[0] - zki[0]) == 0.0) ? 1 : 0;   // NaN / infinity in the frame (see v4)
  __syncthreads();                                            // E3 fully read; frame_bad, lb_range visible
  const bool bad_frame = frame_bad != 0;
  const float qnan32 = __uint_as_float(0x7fc00000u);

  // ---- post ---------------------------------------------------------------------------------------------------
  const double floor_pow = floor_lin * floor_lin;
  const cdd wlane = tw[q];
  if (lb_nbins <= 0) {
    float* fo = out + out_off[seg] + (int64_t)col * F4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = q + TL5 * i;
      const cdd e = {0.5 * (zkr[i] + zpr[i]), 0.5 * (zki[i] - zpi[i])};
      const cdd d = {0.5 * (zkr[i] - zpr[i]), 0.5 * (zki[i] + zpi[i])};
      const cdd o = {d.im, -d.re};
      const cdd wk = ira::cmul(wlane, wuni[i]);              // W_N^k = W_N^q W_N^(256 i); second factor wave-uniform
      const cdd pp = ira::cmul(wk, o);
      fo[k] = bad_frame ? qnan32 : db_of4(e.re + pp.re, e.im + pp.im, floor_pow, floor_db, ltab);
      fo[M4 - k] = bad_frame ? qnan32 : db_of4(e.re - pp.re, e.im - pp.im, floor_pow, floor_db, ltab);   // k = 0 -> bin M
    }
    if (q == 0) fo[M4 / 2] = bad_frame ? qnan32 : db_of4(midr, midi, floor_pow, floor_db, ltab);
    return;
  }
  // fused modal-cloud aggregation, as in v4
  {
    int lo = F4, hi = 0;
    for (int bb = q; bb < lb_nbins; bb += TL5) {
      const int c = lb_count[bb];
      if (c > 0) {
        const int f0 = lb_kbase + lb_first[bb];
        lo = f0 < lo ? f0 : lo;
        hi = f0 + c > hi ? f0 + c : hi;
      }
    }
    atomicMin(&lb_range[0], lo);
    atomicMax(&lb_range[1], hi);
  }
  __syncthreads();
  const int k_lo = lb_range[0], k_hi = lb_range[1];
  const double floor_lin32 = exp10((double)floor_db * 0.05);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = q + TL5 * i;
    const bool need_a = k >= k_lo && k < k_hi, need_b = (M4 - k) >= k_lo && (M4 - k) < k_hi;
    if (!need_a && !need_b) continue;
    const cdd e = {0.5 * (zkr[i] + zpr[i]), 0.5 * (zki[i] - zpi[i])};
    const cdd d = {0.5 * (zkr[i] - zpr[i]), 0.5 * (zki[i] + zpi[i])};
    const cdd o = {d.im, -d.re};
    const cdd wk = ira::cmul(wlane, wuni[i]);
    const cdd pp = ira::cmul(wk, o);
    if (IRA_ABL(ablate & 4)) {
      if (need_a) exd[k] = e.re + pp.re;
      if (need_b) exd[M4 - k] = e.im - pp.im;
      continue;
    }
    if (need_a) exd[k] = lin_of4(e.re + pp.re, e.im + pp.im, floor_pow, floor_db, floor_lin32, ltab);
    if (need_b) exd[M4 - k] = lin_of4(e.re - pp.re, e.im - pp.im, floor_pow, floor_db, floor_lin32, ltab);
  }
  if (q == 0 && M4 / 2 >= k_lo && M4 / 2 < k_hi) exd[M4 / 2] = lin_of4(midr, midi, floor_pow, floor_db, floor_lin32, ltab);
  __syncthreads();
  float* co = out + out_off[seg];
  for (int bb = q; bb < lb_nbins; bb += TL5) {
    const int c = lb_count[bb];
    float val = qnan32;
    if (c > 0) {
      const double* r = exd + lb_kbase + lb_first[bb];
      double acc = r[0];
      for (int k0 = 1; k0 < c; k0 += 8) {
        double v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = (k0 + u < c) ? r[k0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < c) acc += v8[u];
      }
      // 20 log10 m through the table log2 (series to r^6: a few 1e-16 relative, invisible after the float32 rounding) --
      // the library log10 was ~130 of the ~1800 instructions of every wave although only 240 lanes of a frame use it
      val = (float)(6.0205999132796239 * ira::log2_table<6>(fmax(acc / (double)c, 1e-30), ltab));
    }
    co[(int64_t)bb * T_out + col] = (bad_frame && c > 0) ? qnan32 : val;
  }
}

// ------------------------------------------------------------------------------------------------------------
// STFT v8: ONE FRAME ON EIGHT WAVES (512 lanes, 8 complex values per lane), M = 4096 = 8 * 8 * 8 * 8 (round 4).
// v5's own ablation (profiles/r03_stft5_isa_segments.txt): the transform + exchange part of a frame is a latency chain
// (1360 instructions, 12 workgroup barriers) that four waves per SIMD hide only half of; registers (122) and LDS (35 KB per
// frame) both stop at four frames per CU.  With eight values per lane a frame needs <= 64 registers, so the same four
// frames per CU bring EIGHT waves per SIMD -- and the radix-8 decomposition makes the first digit of the output index the
// WAVE number, which turns two of the three exchanges into wave-private transposes without any workgroup barrier:
//   n = 512 n1 + 64 n2 + 8 n3 + n4,  k = k1 + 8 k2 + 64 k3 + 512 k4
//   step 1  wave n2, lane 8 n3 + n4 (m = 64 n2 + lane: contiguous loads): 8-point DFT over n1 of the windowed packed samples,
//           twiddle W_M^(k1 m)
//   E1      the only cross-wave exchange, half the frame at a time: rows k1 of 256 complex; wave k1 reads the n2 = 0..3
//           then 4..7 entries of ITS row, lane-linear both ways (conflict free)                        -- 4 barriers
//   step 2  wave k1, lane m' = 8 n3 + n4: DFT over n2, twiddle W_512^(k2 m')
//   E2      wave-private 8 x 8 transpose (k2 <-> n3) through the wave's own 4.6 KB, real then imaginary parts: rows k2 of
//           72 doubles; reader lane 8 k2 + n4 gathers n3 = 0..7; both ways bank-conflict free            -- no barrier
//   step 3  wave k1, lane 8 k2 + n4: DFT over n3, twiddle W_64^(k3 n4)
//   E3      wave-private transpose (k3 <-> n4): lane stride 9 doubles; reader lane 8 k2 + k3             -- no barrier
//   step 4  wave k1, lane 8 k2 + k3: DFT over n4 -> the lane holds Z[r + 512 k4], r = k1 + 8 k2 + 64 k3
//   mirror  as v5: every lane publishes its upper four values at its own (wave, lane) slot; the partner of r is
//           wave 8 - k1, lane 63 - lane (wave 0: a permutation of its own lanes)                       -- 3 barriers
//   post    as v5 (dB, or linear magnitudes -> log-bin means)
// LDS per frame: 36 864 B of exchange + the 2 KB log table: four frames per CU.
// MEASURED (round 4, 256 x 10 s, modal cloud; profiles/r04_stft8_ab.txt): correct (every modal / waterfall golden), 64 registers
// + 17 spilled, and SLOWER -- 5.80 ms against v5's 4.32 ms.  The instruction budget says why (tools/isa_budget.py): per FRAME
// 8 x 1509 = 12.1 k VALU instructions against v5's 4 x 2508 = 10.0 k (three twiddle stages of eight lanes' worth of powers
// instead of two, address arithmetic and exec-mask bookkeeping per lane on twice the lanes), and v5 already keeps the
// float64 pipes 70 % busy: occupancy cannot buy back 20 % more instructions.  Kept as the A/B (IRA_STFT_V8, tuning build);
// the product runs v5.
// ------------------------------------------------------------------------------------------------------------
constexpr int TL8 = 512;
constexpr int REG8 = 576;                      // doubles of one wave's transpose region: 64 lanes x 9 = 8 rows x 72
constexpr int EX8 = 8 * REG8;                  // doubles per workgroup
static_assert(EX8 >= 2 * 8 * 256 && EX8 >= 2 * 4 * TL8 && EX8 >= F4, "exchange buffer too small");

__global__ __launch_bounds__(TL8) __attribute__((amdgpu_waves_per_eu(8, 8))) void stft8_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int32_t* __restrict__ nframes, int hop,
    const double* __restrict__ window, const cdd* __restrict__ tw, double floor_lin, float floor_db,
    float* __restrict__ out, const int64_t* __restrict__ out_off, const int32_t* __restrict__ frame_sel,
    const int64_t* __restrict__ sel_off, int lb_nbins, int lb_kbase, const int32_t* __restrict__ lb_first,
    const int32_t* __restrict__ lb_count) {
  __shared__ __attribute__((aligned(16))) double exd[EX8];
  __shared__ ira::LogTabEntry ltab[ira::LOGTAB_N];
  __shared__ int lb_range[2];
  __shared__ int frame_bad;
  cdd* ex = reinterpret_cast<cdd*>(exd);
  const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
  const unsigned orig = blockIdx.y * gx + blockIdx.x;
  const unsigned xq = nwg / 8, xr = nwg % 8, xcd = orig % 8;
  const unsigned wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;
  const int seg = (int)(wg / gx);
  const int col = (int)(wg % gx);
  const int T_out = nframes[seg];
  if (col >= T_out) return;
  const int q = threadIdx.x, wave = q >> 6, lane = q & 63;
  ira::build_log_table(ltab, q);
  if (q < 2) lb_range[q] = q == 0 ? F4 : 0;
  const int64_t frame = frame_sel ? (int64_t)frame_sel[sel_off[seg] + col] : (int64_t)col;
  const float* fx = x + off[seg] + frame * hop;
  cdd wuni[4];                                                   // wave-uniform post factors W_N^(512 i): scalar loads, up front
#pragma unroll
  for (int i = 0; i < 4; ++i) wuni[i] = tw[TL8 * i];

  // ---- step 1: lane (n3 = lane >> 3, n4 = lane & 7) of wave n2 transforms m = 64 n2 + lane (contiguous loads) --------------------
  cdd v[8];
  {
    const int m = q;
    float2 xs[8];
    double2 ws[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
      const int n = n1 * 512 + m;
      xs[n1] = *reinterpret_cast<const float2*>(fx + 2 * n);
      ws[n1] = *reinterpret_cast<const double2*>(window + 2 * n);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) v[n1] = {(double)xs[n1].x * ws[n1].x, (double)xs[n1].y * ws[n1].y};
    dft_dif<double, 8>(v);
    ira::twiddle8<double, true>(v, tw[2 * m]);                   // W_M^(k1 m) = W_N^(2 m k1), k1 at v[brev3(k1)]
  }

  // ---- E1: rows k1 of 256 complex, the writers' lane order kept; wave k1 reads its row ----------------------------------------
  cdd b[8];
  const bool lower = q < 256;
  if (lower) {
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) ex[k1 * 256 + q] = v[brev_bits(k1, 3)];
  }
  __syncthreads();
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) b[n2] = ex[wave * 256 + n2 * 64 + lane];
  __syncthreads();
  if (!lower) {
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) ex[k1 * 256 + (q - 256)] = v[brev_bits(k1, 3)];
  }
  __syncthreads();
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) b[4 + n2] = ex[wave * 256 + n2 * 64 + lane];
  __syncthreads();                                               // E1 fully read: the wave regions below overlap it

  // ---- step 2 (wave = k1, lane = m' = 8 n3 + n4) and the wave-private transpose k2 <-> n3 -------------------------------------
  double* reg = exd + wave * REG8;
  dft_dif<double, 8>(b);
  ira::twiddle8<double, true>(b, tw[16 * lane]);                 // W_512^(k2 m') = W_N^(16 m' k2)
  cdd c[8];
  {
    // rows k2 of 72 doubles, column = producer lane 8 n3 + n4; reader lane 8 k2 + n4 gathers n3 = 0..7
    const int rd = (lane >> 3) * 72 + (lane & 7);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) reg[k2 * 72 + lane] = b[brev_bits(k2, 3)].re;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) c[n3].re = reg[rd + n3 * 8];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) reg[k2 * 72 + lane] = b[brev_bits(k2, 3)].im;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) c[n3].im = reg[rd + n3 * 8];
    __builtin_amdgcn_wave_barrier();
  }

  // ---- step 3 (lane = 8 k2 + n4) and the wave-private transpose k3 <-> n4 ---------------------------------------------------
  dft_dif<double, 8>(c);
  ira::twiddle8<double, true>(c, tw[128 * (lane & 7)]);         // W_64^(k3 n4) = W_N^(128 n4 k3)
  cdd d[8];
  {
    // producer lane 8 k2 + n4 writes its eight k3 values 9 doubles apart; reader lane 8 k2 + k3 gathers n4 = 0..7
    const int rd = (lane >> 3) * 72 + (lane & 7);               // (8 k2 + n4) * 9 + k3 at n4 = 0
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) reg[lane * 9 + k3] = c[brev_bits(k3, 3)].re;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n4 = 0; n4 < 8; ++n4) d[n4].re = reg[rd + n4 * 9];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) reg[lane * 9 + k3] = c[brev_bits(k3, 3)].im;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n4 = 0; n4 < 8; ++n4) d[n4].im = reg[rd + n4 * 9];
  }

  // ---- step 4: lane (k2 = lane >> 3, k3 = lane & 7) of wave k1 holds Z[r + 512 k4] at d[brev3(k4)] --------------------------------
  dft_dif<double, 8>(d);
  const int r = wave + 8 * (lane >> 3) + 64 * (lane & 7);

  // ---- mirror exchange: the upper four values of every lane at its own slot; partner of r = (512 - r) mod 512 -------------------
  __syncthreads();                                               // every wave has left its transpose region
#pragma unroll
  for (int j = 0; j < 4; ++j) ex[j * TL8 + q] = d[brev_bits(4 + j, 3)];
  __syncthreads();
  double zkr[4], zki[4], zpr[4], zpi[4], midr, midi;
  {
    int pos;
    if (wave > 0) {
      pos = (8 - wave) * 64 + (63 - lane);
    } else {
      const int k2 = lane >> 3, k3 = lane & 7;
      pos = k2 > 0 ? (8 - k2) * 8 + (7 - k3) : ((8 - k3) & 7);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // partner entry k4 = 7 - i (slot 3 - i); r = 0 pairs with itself: k4 = 8 - i (slot 4 - i), i = 0 -> Z[0] itself
      const int slot = r == 0 ? 4 - i : 3 - i;
      cdd zp = d[0];
      if (!(r == 0 && i == 0)) zp = ex[slot * TL8 + pos];
      zkr[i] = d[brev_bits(i, 3)].re; zki[i] = d[brev_bits(i, 3)].im;
      zpr[i] = zp.re; zpi[i] = zp.im;
    }
    const cdd mid = ex[0];                                       // Z[2048]: r = 0, k4 = 4
    midr = mid.re; midi = mid.im;
  }
  if (q == 0) frame_bad = !((zkr[0] - zkr[0]) + (zki[0] - zki[0]) == 0.0) ? 1 : 0;   // NaN / infinity in the frame (see v4)
  __syncthreads();                                               // mirror slots fully read; frame_bad, lb_range visible
  const bool bad_frame = frame_bad != 0;
  const float qnan32 = __uint_as_float(0x7fc00000u);

  // ---- post ---------------------------------------------------------------------------------------------------
  const double floor_pow = floor_lin * floor_lin;
  const cdd wlane = tw[r];
  if (lb_nbins <= 0) {
    float* fo = out + out_off[seg] + (int64_t)col * F4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = r + TL8 * i;
      const cdd e = {0.5 * (zkr[i] + zpr[i]), 0.5 * (zki[i] - zpi[i])};
      const cdd dd = {0.5 * (zkr[i] - zpr[i]), 0.5 * (zki[i] + zpi[i])};
      const cdd o = {dd.im, -dd.re};
      const cdd wk = ira::cmul(wlane, wuni[i]);                  // W_N^k = W_N^r W_N^(512 i); second factor wave-uniform
      const cdd pp = ira::cmul(wk, o);
      fo[k] = bad_frame ? qnan32 : db_of4(e.re + pp.re, e.im + pp.im, floor_pow, floor_db, ltab);
      fo[M4 - k] = bad_frame ? qnan32 : db_of4(e.re - pp.re, e.im - pp.im, floor_pow, floor_db, ltab);   // k = 0 -> bin M
    }
    if (q == 0) fo[M4 / 2] = bad_frame ? qnan32 : db_of4(midr, midi, floor_pow, floor_db, ltab);
    return;
  }
  // fused modal-cloud aggregation, as in v4 / v5
  {
    int lo = F4, hi = 0;
    for (int bb = q; bb < lb_nbins; bb += TL8) {
      const int cn = lb_count[bb];
      if (cn > 0) {
        const int f0 = lb_kbase + lb_first[bb];
        lo = f0 < lo ? f0 : lo;
        hi = f0 + cn > hi ? f0 + cn : hi;
      }
    }
    if (q < lb_nbins) { atomicMin(&lb_range[0], lo); atomicMax(&lb_range[1], hi); }
  }
  __syncthreads();
  const int k_lo = lb_range[0], k_hi = lb_range[1];
  const double floor_lin32 = exp10((double)floor_db * 0.05);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = r + TL8 * i;
    const bool need_a = k >= k_lo && k < k_hi, need_b = (M4 - k) >= k_lo && (M4 - k) < k_hi;
    if (!need_a && !need_b) continue;
    const cdd e = {0.5 * (zkr[i] + zpr[i]), 0.5 * (zki[i] - zpi[i])};
    const cdd dd = {0.5 * (zkr[i] - zpr[i]), 0.5 * (zki[i] + zpi[i])};
    const cdd o = {dd.im, -dd.re};
    const cdd wk = ira::cmul(wlane, wuni[i]);
    const cdd pp = ira::cmul(wk, o);
    if (need_a) exd[k] = lin_of4(e.re + pp.re, e.im + pp.im, floor_pow, floor_db, floor_lin32, ltab);
    if (need_b) exd[M4 - k] = lin_of4(e.re - pp.re, e.im - pp.im, floor_pow, floor_db, floor_lin32, ltab);
  }
  if (q == 0 && M4 / 2 >= k_lo && M4 / 2 < k_hi) exd[M4 / 2] = lin_of4(midr, midi, floor_pow, floor_db, floor_lin32, ltab);
  __syncthreads();
  float* co = out + out_off[seg];
  for (int bb = q; bb < lb_nbins; bb += TL8) {
    const int cn = lb_count[bb];
    float val = qnan32;
    if (cn > 0) {
      const double* rr = exd + lb_kbase + lb_first[bb];
      double acc = rr[0];
      for (int k0 = 1; k0 < cn; k0 += 8) {
        double v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = (k0 + u < cn) ? rr[k0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < cn) acc += v8[u];
      }
      val = (float)(6.0205999132796239 * ira::log2_table<6>(fmax(acc / (double)cn, 1e-30), ltab));
    }
    co[(int64_t)bb * T_out + col] = (bad_frame && cn > 0) ? qnan32 : val;
  }
}

}  // namespace

// float64 / n_fft 8192, frame-major output only; anything else returns IRA_E_UNSUPPORTED.
int32_t ira_stft4_dispatch_tf(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                              int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                              int32_t precision, double floor_db, float* out, const int64_t* out_off,
                              const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st) {
  if (precision != 64 || n_fft != 8192) return IRA_E_UNSUPPORTED;
  const double floor_lin = std::pow(10.0, floor_db / 20.0);
  dim3 grid(max_frames, nseg);
  if (ira_tune_flag("IRA_STFT_V4"))
    stft4_kernel<<<grid, TL4, 0, st>>>(x, off, nframes, hop, static_cast<const double*>(window),
                                       static_cast<const cdd*>(tw), floor_lin, (float)floor_db, out, out_off, frame_sel,
                                       sel_off, 0, 0, nullptr, nullptr);
  else if (ira_tune_flag("IRA_STFT_V8"))                     // A/B (tuning build): one frame on eight waves, see stft8_kernel
    stft8_kernel<<<grid, TL8, 0, st>>>(x, off, nframes, hop, static_cast<const double*>(window),
                                       static_cast<const cdd*>(tw), floor_lin, (float)floor_db, out, out_off, frame_sel,
                                       sel_off, 0, 0, nullptr, nullptr);
  else
    stft5_kernel<<<grid, TL5, (size_t)ira_tune_int("IRA_STFT5_LDS_PAD", 0), st>>>(x, off, nframes, hop, static_cast<const double*>(window),
                                       static_cast<const cdd*>(tw), floor_lin, (float)floor_db, out, out_off, frame_sel,
                                       sel_off, 0, 0, nullptr, nullptr, ira_tune_int("IRA_STFT5_ABLATE", 0));
  IRA_RETURN_LAUNCH();
}

// STFT + log-bin aggregation in one kernel: out[e] is the (nbins, T_e) curve matrix of ira_logbin_aggregate.
int32_t ira_stft4_dispatch_logbin(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                                  int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                                  int32_t precision, double floor_db, int32_t k_base, const int32_t* first,
                                  const int32_t* count, int32_t nbins, float* curves, const int64_t* curves_off,
                                  hipStream_t st) {
  if (precision != 64 || n_fft != 8192) return IRA_E_UNSUPPORTED;
  const double floor_lin = std::pow(10.0, floor_db / 20.0);
  dim3 grid(max_frames, nseg);
  if (ira_tune_flag("IRA_STFT_V4"))
    stft4_kernel<<<grid, TL4, 0, st>>>(x, off, nframes, hop, static_cast<const double*>(window),
                                       static_cast<const cdd*>(tw), floor_lin, (float)floor_db, curves, curves_off,
                                       nullptr, nullptr, nbins, k_base, first, count);
  else if (ira_tune_flag("IRA_STFT_V8"))
    stft8_kernel<<<grid, TL8, 0, st>>>(x, off, nframes, hop, static_cast<const double*>(window),
                                       static_cast<const cdd*>(tw), floor_lin, (float)floor_db, curves, curves_off,
                                       nullptr, nullptr, nbins, k_base, first, count);
  else
    stft5_kernel<<<grid, TL5, (size_t)ira_tune_int("IRA_STFT5_LDS_PAD", 0), st>>>(x, off, nframes, hop, static_cast<const double*>(window),
                                       static_cast<const cdd*>(tw), floor_lin, (float)floor_db, curves, curves_off,
                                       nullptr, nullptr, nbins, k_base, first, count, ira_tune_int("IRA_STFT5_ABLATE", 0));
  IRA_RETURN_LAUNCH();
}
