// Where does the float32 STFT frame's issue rate go?    tools/micro/dft16_rate.hip showed the butterflies
// issuing at 1.8 wave-instructions per CU-cycle and the whole frame at 1.23; this file separates the frame into
//   0 nopost     steps 1-3 of the frame (window, 2 x dft16 + powers16 twiddles, 2 x dft16 + twiddles, 4 x dft8), registers only
//   1 post       the untangle + dB epilogue of 16 bin pairs + the middle bin exactly as the kernel has it (stft6_kernel)
//   2 post_nolog the same with v_log_f32 replaced by a multiply (what the quarter-rate transcendental costs)
//   3 post_new   the restructured epilogue: halves folded into the dB constant, floor as v_max in the dB domain, no per-bin
//                NaN select (the frame's flag is wave-uniform: a branch)
//   4 frame_new  steps 1-3 + the restructured epilogue
//   5 frame      steps 1-3 + the kernel's epilogue (= dft16_rate variant 3)
//   6 post_poly  variant 3 with v_log_f32 replaced by v_frexp_mant / v_frexp_exp + a degree-6 polynomial (full-rate instructions)
//   7 frame_poly steps 1-3 + variant 6
//   8 / 9 / 10   variants 0 / 7 / 5 with the kernel's LDS exchanges E1 / E2 / E3 between the steps (wave-private buffers)
//   11 / 12      variants 8 / (steps + restructured epilogue) with E3 replaced by a lane permutation (ds_bpermute_b32)
// All on synthetic registers, no memory; 1..4 waves per SIMD, one 16-wave workgroup per CU at most.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I audio_analysis_amd/csrc tools/micro/stft_epilogue_rate.hip -o /tmp/stft_epilogue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#include "ira_fft_reg.h"

using ira::brev_bits;
using ira::cplx;
using ira::dft_dif;
using ira::powers16;
typedef cplx<float> cf;

#define PIN(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ float db_of(float re, float im, float floor_pow, float floor_db) {
  const float p = re * re + im * im;
  if (!(p > floor_pow)) return floor_db;
  return 3.0102999566398120f * __log2f(p);
}

constexpr int M3 = 2048, ROWH = 66, E2N3 = 272, EXC = 1072;
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// steps 1-3 with the kernel's half-size LDS exchanges E1 / E2 / E3 (wave-private buffers, the kernel's strides)
template <bool PERM = false>
__device__ __forceinline__ void steps123_lds(cf (&v)[16], cf w1, cf w2, cf w3, cf wl, float (&zkr)[16], float (&zpr)[16],
                                             float (&zki)[16], float (&zpi)[16], cf* ex, int q) {
  float* exf = reinterpret_cast<float*>(ex);
  const int k1l = q & 15, n3a = q >> 4;
  cf a1[16];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    cf u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = {v[h].re * w2.re + (float)(i + 1), v[h].im * w2.im - (float)(i + 1)};
    dft_dif<float, 16>(u);
    cf p[16];
    powers16<float>(h ? w2 : w1, p);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      const cf a = u[brev_bits(k1, 4)];
      const cf r = (k1 == 0) ? a : ira::cmul(a, p[k1]);
      if (h == 0) ex[k1 * ROWH + q] = r; else a1[k1] = r;
    }
  }
  cf b2[2][16];
  wave_sync();
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) b2[hb][n2] = ex[k1l * ROWH + n2 * 8 + n3a + 4 * hb];
  wave_sync();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) ex[k1 * ROWH + q] = a1[k1];
  wave_sync();
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) b2[hb][8 + n2] = ex[k1l * ROWH + n2 * 8 + n3a + 4 * hb];
  wave_sync();
  cf z3[4][8];
  {
    cf p[16];
    dft_dif<float, 16>(b2[0]);
    powers16<float>(w3, p);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      const cf a = b2[0][brev_bits(k2, 4)];
      ex[k1l + 16 * k2 + E2N3 * n3a] = (k2 == 0) ? a : ira::cmul(a, p[k2]);
    }
    dft_dif<float, 16>(b2[1]);
    powers16<float>(wl, p);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) b2[1][brev_bits(k2, 4)] = ira::cmul(b2[1][brev_bits(k2, 4)], p[k2]);
  }
  wave_sync();
#pragma unroll
  for (int hh = 0; hh < 4; ++hh)
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) z3[hh][n3] = ex[k1l + 16 * (n3a + 4 * hh) + E2N3 * n3];
  wave_sync();
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) ex[k1l + 16 * k2 + E2N3 * n3a] = b2[1][brev_bits(k2, 4)];
  wave_sync();
#pragma unroll
  for (int hh = 0; hh < 4; ++hh)
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) z3[hh][4 + n3] = ex[k1l + 16 * (n3a + 4 * hh) + E2N3 * n3];
  wave_sync();
#pragma unroll
  for (int hh = 0; hh < 4; ++hh) dft_dif<float, 8>(z3[hh]);
  if (PERM) {
    // E3 without LDS: the lane already holds Z[q + 64 i'], i' = hh + 4 k3 < 32; the mirror Z[2048 - k], k = q + 64 i (i < 16),
    // sits in lane (64 - q) & 63 as its register 31 - i -- except for lane 0, whose mirror is its OWN register 32 - i (i >= 1)
    // or register 0 (i = 0).  32 ds_bpermute_b32 (the LDS crossbar, no memory) + 32 selects instead of 130 LDS accesses.
    const int src = ((64 - q) & 63) << 2;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const cf own = z3[i & 3][brev_bits(i >> 2, 3)];
      const int ip = 31 - i, il = (32 - i) & 31;             // partner register of lanes q > 0 / of lane 0
      const cf pv = z3[ip & 3][brev_bits(ip >> 2, 3)], lv = z3[il & 3][brev_bits(il >> 2, 3)];
      const float pr = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(pv.re)));
      const float pi = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(pv.im)));
      zkr[i] = own.re; zki[i] = own.im;
      zpr[i] = q == 0 ? lv.re : pr;
      zpi[i] = q == 0 ? lv.im : pi;
    }
    return;
  }
#pragma unroll
  for (int hh = 0; hh < 4; ++hh)
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) exf[q + 64 * hh + 256 * k3] = z3[hh][brev_bits(k3, 3)].re;
  wave_sync();
#pragma unroll
  for (int i = 0; i < 16; ++i) { const int kk = q + 64 * i; zkr[i] = exf[kk]; zpr[i] = exf[(M3 - kk) & (M3 - 1)]; }
  wave_sync();
#pragma unroll
  for (int hh = 0; hh < 4; ++hh)
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) exf[q + 64 * hh + 256 * k3] = z3[hh][brev_bits(k3, 3)].im;
  wave_sync();
#pragma unroll
  for (int i = 0; i < 16; ++i) { const int kk = q + 64 * i; zki[i] = exf[kk]; zpi[i] = exf[(M3 - kk) & (M3 - 1)]; }
  wave_sync();
}

// steps 1-3 on registers (as dft16_rate variant 3): v -> zkr / zpr / zki / zpi
__device__ __forceinline__ void steps123(cf (&v)[16], cf w1, cf w2, cf w3, cf wl, float (&zkr)[16], float (&zpr)[16],
                                         float (&zki)[16], float (&zpi)[16]) {
  cf a0[16], a1[16];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    cf u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = {v[h].re * w2.re + (float)(i + 1), v[h].im * w2.im - (float)(i + 1)};
    dft_dif<float, 16>(u);
    cf p[16];
    powers16<float>(h ? w2 : w1, p);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      const cf a = u[brev_bits(k1, 4)];
      const cf r = (k1 == 0) ? a : ira::cmul(a, p[k1]);
      if (h == 0) a0[k1] = r; else a1[k1] = r;
    }
  }
  cf b2[2][16];
#pragma unroll
  for (int i = 0; i < 8; ++i) { b2[0][i] = a0[2 * i]; b2[1][i] = a0[2 * i + 1]; b2[0][8 + i] = a1[2 * i]; b2[1][8 + i] = a1[2 * i + 1]; }
  cf z3[4][8];
  {
    cf p[16];
    dft_dif<float, 16>(b2[0]);
    powers16<float>(w3, p);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) b2[0][brev_bits(k2, 4)] = ira::cmul(b2[0][brev_bits(k2, 4)], p[k2]);
    dft_dif<float, 16>(b2[1]);
    powers16<float>(wl, p);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) b2[1][brev_bits(k2, 4)] = ira::cmul(b2[1][brev_bits(k2, 4)], p[k2]);
  }
#pragma unroll
  for (int hh = 0; hh < 4; ++hh)
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) { z3[hh][n3] = b2[0][4 * hh + n3]; z3[hh][4 + n3] = b2[1][4 * hh + n3]; }
#pragma unroll
  for (int hh = 0; hh < 4; ++hh) dft_dif<float, 8>(z3[hh]);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    zkr[i] = z3[i >> 2][2 * (i & 3)].re; zpr[i] = z3[i >> 2][2 * (i & 3) + 1].re;
    zki[i] = z3[i >> 2][2 * (i & 3)].im; zpi[i] = z3[i >> 2][2 * (i & 3) + 1].im;
  }
}

// the kernel's epilogue; LOG = false: v_log_f32 replaced by a multiply
template <bool LOG>
__device__ __forceinline__ void post_old(const float (&zkr)[16], const float (&zpr)[16], const float (&zki)[16], const float (&zpi)[16],
                                         cf wl, const cf* __restrict__ wuni, float& sl, float& sh) {
  const float floor_pow = 1e-12f, floor_db = -120.0f, qn = __uint_as_float(0x7fc00000u);
  const float z0 = (zkr[0] - zkr[0]) + (zki[0] - zki[0]);
  const bool bad = __shfl(z0, 0, 64) != 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const cf e = {0.5f * (zkr[i] + zpr[i]), 0.5f * (zki[i] - zpi[i])};
    const cf d = {0.5f * (zkr[i] - zpr[i]), 0.5f * (zki[i] + zpi[i])};
    const cf o = {d.im, -d.re};
    const cf wk = ira::cmul(wl, wuni[i]);
    const cf pp = ira::cmul(wk, o);
    float lo, hi;
    if (LOG) {
      lo = bad ? qn : db_of(e.re + pp.re, e.im + pp.im, floor_pow, floor_db);
      hi = bad ? qn : db_of(e.re - pp.re, e.im - pp.im, floor_pow, floor_db);
    } else {
      const float a = e.re + pp.re, b = e.im + pp.im, c = e.re - pp.re, dd = e.im - pp.im;
      const float p1 = a * a + b * b, p2 = c * c + dd * dd;
      lo = bad ? qn : (!(p1 > floor_pow) ? floor_db : 3.0102999566398120f * (p1 * 1.0001f));
      hi = bad ? qn : (!(p2 > floor_pow) ? floor_db : 3.0102999566398120f * (p2 * 1.0001f));
    }
    sl += lo; sh += hi;
  }
}

// restructured: X = (E' + P') / 2 with E' = Zk + conj Zp, P' = W (-i)(Zk - conj Zp); |X|^2 = |E' + P'|^2 / 4, so
// dB = 3.0103 log2 |E' + P'|^2 - 6.0206, floored by one v_max in the dB domain (NaN-safe: v_max returns the other operand)
__device__ __forceinline__ float db_new(float re, float im, float floor_db) {
  const float p = re * re + im * im;
  const float db = fmaf(3.0102999566398120f, __log2f(p), -6.0205999132796240f);
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(db), "v"(floor_db));      // (fmaxf adds a canonicalising v_max per operand)
  return r;
}
// ... and without the transcendental unit: p = m 2^e (v_frexp_*), 10 log10(p) - 6.0206 = 3.0103 e + Q(m - 3/4), Q of degree 6
// on [1/2, 1): 7e-6 dB (tools: Chebyshev interpolation, evaluated in float32 Horner form)
__device__ __forceinline__ float db_poly(float re, float im, float floor_db) {
  const float p = re * re + im * im;
  const float m = __builtin_amdgcn_frexp_mantf(p) - 0.75f;
  const float e = (float)__builtin_amdgcn_frexp_expf(p);
  float q = -4.7333541814e+00f;
  q = fmaf(q, m, 4.2295899082e+00f);
  q = fmaf(q, m, -3.4099165525e+00f);
  q = fmaf(q, m, 3.4130717282e+00f);
  q = fmaf(q, m, -3.8605656273e+00f);
  q = fmaf(q, m, 5.7907383526e+00f);
  q = fmaf(q, m, -7.2699872794e+00f);
  const float db = fmaf(e, 3.0102999566398120f, q);
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(db), "v"(floor_db));
  return r;
}
template <bool POLY>
__device__ __forceinline__ void post_new(const float (&zkr)[16], const float (&zpr)[16], const float (&zki)[16], const float (&zpi)[16],
                                         cf wl, const cf* __restrict__ wuni, float& sl, float& sh) {
  const float floor_db = -120.0f, qn = __uint_as_float(0x7fc00000u);
  const float z0 = (zkr[0] - zkr[0]) + (zki[0] - zki[0]);
  const bool bad = __shfl(z0, 0, 64) != 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const cf e = {zkr[i] + zpr[i], zki[i] - zpi[i]};
    const cf o = {zki[i] + zpi[i], zpr[i] - zkr[i]};        // (-i) (Zk - conj Zp)
    const cf wk = ira::cmul(wl, wuni[i]);
    const cf pp = ira::cmul(wk, o);
    sl += POLY ? db_poly(e.re + pp.re, e.im + pp.im, floor_db) : db_new(e.re + pp.re, e.im + pp.im, floor_db);
    sh += POLY ? db_poly(e.re - pp.re, e.im - pp.im, floor_db) : db_new(e.re - pp.re, e.im - pp.im, floor_db);
  }
  if (bad) { sl = qn; sh = qn; }                               // (the kernel: a wave-uniform branch to a NaN store path)
}

template <int VAR>
__global__ __launch_bounds__(1024) void k(float* out, const cf* __restrict__ tw, unsigned long long* cyc, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, q = tid & 63;
  cf* ex = reinterpret_cast<cf*>(smem) + (size_t)(tid >> 6) * EXC;
  cf w1 = tw[q], w2 = tw[64 + q], w3 = tw[128 + (q >> 4)], wl = tw[192 + q];
  cf wuni[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) wuni[i] = tw[64 * i];       // wave-uniform: scalar registers, as in the kernel
  cf v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = {seed * (float)(tid + i), seed * (float)(i + 1)};
  float zkr[16], zpr[16], zki[16], zpi[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { zkr[i] = seed * (float)(tid + i); zpr[i] = seed * (float)(i + 3); zki[i] = seed * (float)(tid - i); zpi[i] = seed * (float)(2 * i + 1); }
  float acc = 0.0f;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    PIN(w1.re); PIN(w1.im); PIN(w2.re); PIN(w2.im); PIN(w3.re); PIN(w3.im); PIN(wl.re); PIN(wl.im);
    float sl = seed, sh = seed;
    if (VAR == 0 || VAR == 4 || VAR == 5 || VAR == 7) steps123(v, w1, w2, w3, wl, zkr, zpr, zki, zpi);
    else if (VAR == 11 || VAR == 12) steps123_lds<true>(v, w1, w2, w3, wl, zkr, zpr, zki, zpi, ex, q);
    else if (VAR >= 8) steps123_lds(v, w1, w2, w3, wl, zkr, zpr, zki, zpi, ex, q);
    else {
#pragma unroll
      for (int i = 0; i < 16; ++i) { PIN(zkr[i]); PIN(zpr[i]); PIN(zki[i]); PIN(zpi[i]); }
    }
    if (VAR == 0 || VAR == 8 || VAR == 11) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { sl += zkr[i] + zpr[i]; sh += zki[i] + zpi[i]; }    // keep the 64 results live (64 adds)
    } else if (VAR == 1 || VAR == 5 || VAR == 10) post_old<true>(zkr, zpr, zki, zpi, wl, wuni, sl, sh);
    else if (VAR == 2) post_old<false>(zkr, zpr, zki, zpi, wl, wuni, sl, sh);
    else if (VAR == 6 || VAR == 7 || VAR == 9) post_new<true>(zkr, zpr, zki, zpi, wl, wuni, sl, sh);
    else post_new<false>(zkr, zpr, zki, zpi, wl, wuni, sl, sh);
    v[0] = {sl * 1e-3f, sh * 1e-3f};
    v[1] = {sh * 1e-3f, sl * 1e-3f};
    PIN(v[0].re); PIN(v[0].im); PIN(v[1].re); PIN(v[1].im);
    acc += sl;
  }
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float s = acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i].re + v[i].im;
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

template <int VAR>
void run(const char* name, int threads, int valu_per_iter, int iters) {
  float* out; unsigned long long* cyc; cf* tw;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 16); hipMalloc(&tw, 2048 * sizeof(cf));
  cf* h = (cf*)malloc(2048 * sizeof(cf));
  for (int i = 0; i < 2048; ++i) { h[i].re = (float)cos(-2.0 * M_PI * i / 4096.0); h[i].im = (float)sin(-2.0 * M_PI * i / 4096.0); }
  hipMemcpy(tw, h, 2048 * sizeof(cf), hipMemcpyHostToDevice);
  free(h);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = (size_t)(threads / 64) * EXC * sizeof(cf);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<VAR><<<256, threads, lds>>>(out, tw, cyc, iters, 1e-3f);
  hipEventRecord(e0);
  k<VAR><<<256, threads, lds>>>(out, tw, cyc, iters, 1e-3f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hh[2]; hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost);
  const double mhz = 100.0 * (double)hh[0] / (double)hh[1];
  const double winstr = 256.0 * (threads / 64) * (double)iters * valu_per_iter;
  const double per_cu_cycle = winstr / (ms * 1e-3) / 256.0 / (mhz * 1e6);
  // SIMD-cycles one wave's iteration occupies its SIMD for = 4 SIMDs x cycles / (waves x iterations)
  const double simd_cycles_per_iter = 4.0 * (ms * 1e-3) * (mhz * 1e6) / ((threads / 64) * (double)iters);
  printf("%-10s waves/SIMD %d  VALU/iter %5d  %8.3f ms  clock %4.0f MHz  %.3f wave-instr per CU-cycle  (%.2f SIMD-cycles per instruction; %7.0f SIMD-cycles per iteration)\n",
         name, threads / 256, valu_per_iter, ms, mhz, per_cu_cycle, 4.0 / per_cu_cycle, simd_cycles_per_iter);
  hipFree(out); hipFree(cyc); hipFree(tw);
}

int main(int argc, char** argv) {
  int cnt[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (argc > 1) {
    FILE* f = fopen(argv[1], "r");
    if (f) { for (int i = 0; i < 13; ++i) if (fscanf(f, "%d", &cnt[i]) != 1) break; fclose(f); }
  }
  for (int th : {512, 1024}) {
    run<0>("nopost", th, cnt[0], 400);
    run<1>("post", th, cnt[1], 1600);
    run<2>("post_nolog", th, cnt[2], 1600);
    run<3>("post_new", th, cnt[3], 1600);
    run<4>("frame_new", th, cnt[4], 400);
    run<5>("frame", th, cnt[5], 400);
    run<6>("post_poly", th, cnt[6], 1600);
    run<7>("frame_poly", th, cnt[7], 400);
    run<8>("nopost+lds", th, cnt[8], 400);
    run<9>("f_poly+lds", th, cnt[9], 400);
    run<10>("frame+lds", th, cnt[10], 400);
    run<11>("nopost+perm", th, cnt[11], 400);
    run<12>("f_new+perm", th, cnt[12], 400);
  }
  return 0;
}
