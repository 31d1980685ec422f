// Issue-rate microbenchmark of the float32 STFT's OWN instruction stream: the register-resident
// dft_dif<float,16> + powers16 + cmul body of stft3_kernel (ira_stft3.hip) on synthetic registers, no global memory,
// at 1..4 waves per SIMD, next to a stream of independent v_fma_f32 (tools/pk_f32_rate.hip measured 1.43-1.52
// wave-instructions per CU-cycle for those).
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I audio_analysis_amd/csrc tools/micro/dft16_rate.hip -o tools/micro/dft16_rate
//   hipcc ... -S -o /tmp/dft16_rate.s  (device assembly) + tools/micro/dft16_count.py -> VALU instructions per loop body
//   tools/micro/dft16_rate <counts.txt>
//
// Variants (one loop iteration each; the loop is not unrolled):
//   0 fma       256 independent v_fma_f32 (8 accumulators x 32)                                  -- the ceiling
//   1 dft16     dft_dif<float,16> on 16 complex registers                                         -- butterflies only
//   2 step1     window multiply + dft_dif<16> + powers16 + 15 complex twiddle multiplies          -- step 1 of the kernel
//   3 frame     the whole arithmetic of one frame (2 x step 1, 2 x step 2, 4 x dft8, untangle + dB of 33 bins), data
//               passed between the steps in registers (no LDS): the transform's own instruction stream
//   4 frame+lds variant 3 with the kernel's half-size LDS exchanges E1 / E2 / E3 (wave-private buffers, same strides)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ira_fft_reg.h"

using ira::brev_bits;
using ira::cplx;
using ira::dft_dif;
using ira::powers16;
typedef cplx<float> cf;

constexpr int M3 = 2048;
constexpr int ROWH = 66, E2N3 = 272, EXC = 1072;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float db_of(float re, float im, float floor_pow, float floor_db) {
  const float p = re * re + im * im;
  if (!(p > floor_pow)) return floor_db;
  return 3.0102999566398120f * __log2f(p);
}

#define PIN(x) asm volatile("" : "+v"(x))

template <int VAR>
__global__ __launch_bounds__(1024) void k(float* out, const cf* __restrict__ tw, unsigned long long* cyc, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, q = tid & 63, team = tid >> 6;
  cf* ex = reinterpret_cast<cf*>(smem) + (size_t)team * EXC;
  float* exf = reinterpret_cast<float*>(ex);
  cf w1 = tw[q], w2 = tw[64 + q], w3 = tw[128 + (q >> 4)], wl = tw[192 + q];
  cf v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = {seed * (float)(tid + i), seed * (float)(i + 1)};
  float acc = 0.0f;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    // the kernel computes its twiddle powers once per frame: keep the optimiser from hoisting them out of this loop
    PIN(w1.re); PIN(w1.im); PIN(w2.re); PIN(w2.im); PIN(w3.re); PIN(w3.im); PIN(wl.re); PIN(wl.im);
    if (VAR == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i].re) : "v"(w1.re), "v"(w1.im));
    } else if (VAR == 1) {
      dft_dif<float, 16>(v);
#pragma unroll
      for (int i = 0; i < 16; ++i) { PIN(v[i].re); PIN(v[i].im); }
    } else if (VAR == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = {v[i].re * w2.re, v[i].im * w2.im};
      dft_dif<float, 16>(v);
      cf p[16];
      powers16<float>(w1, p);
      cf o[16];
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        const cf a = v[brev_bits(k1, 4)];
        o[k1] = (k1 == 0) ? a : ira::cmul(a, p[k1]);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { v[i] = o[i]; PIN(v[i].re); PIN(v[i].im); }
    } else {
      constexpr bool LDS = VAR == 4;
      const int k1l = q & 15, n3a = q >> 4;
      // ---- step 1, two halves --------------------------------------------------------------------------------------------
      cf a0[16], a1[16];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        cf u[16];                      // "sample x window": two multiplies per element, like the kernel's step 1
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = {v[h].re * w2.re + (float)(i + 1), v[h].im * w2.im - (float)(i + 1)};
        dft_dif<float, 16>(u);
        cf p[16];
        powers16<float>(h ? w2 : w1, p);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
          const cf a = u[brev_bits(k1, 4)];
          const cf r = (k1 == 0) ? a : ira::cmul(a, p[k1]);
          if (h == 0) { if (LDS) ex[k1 * ROWH + q] = r; else a0[k1] = r; } else a1[k1] = r;
        }
      }
      cf b2[2][16];
      if (LDS) {
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
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { b2[0][i] = a0[2 * i]; b2[1][i] = a0[2 * i + 1]; b2[0][8 + i] = a1[2 * i]; b2[1][8 + i] = a1[2 * i + 1]; }
      }
      // ---- step 2 ------------------------------------------------------------------------------------------------------
      cf z3[4][8];
      {
        cf p[16];
        dft_dif<float, 16>(b2[0]);
        powers16<float>(w3, p);
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2) b2[0][brev_bits(k2, 4)] = ira::cmul(b2[0][brev_bits(k2, 4)], p[k2]);
        if (LDS) {
#pragma unroll
          for (int k2 = 0; k2 < 16; ++k2) ex[k1l + 16 * k2 + E2N3 * n3a] = b2[0][brev_bits(k2, 4)];
        }
        dft_dif<float, 16>(b2[1]);
        powers16<float>(wl, p);
#pragma unroll
        for (int k2 = 1; k2 < 16; ++k2) b2[1][brev_bits(k2, 4)] = ira::cmul(b2[1][brev_bits(k2, 4)], p[k2]);
      }
      if (LDS) {
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
      } else {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
          for (int n3 = 0; n3 < 4; ++n3) { z3[hh][n3] = b2[0][4 * hh + n3]; z3[hh][4 + n3] = b2[1][4 * hh + n3]; }
      }
      // ---- step 3 ------------------------------------------------------------------------------------------------------
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) dft_dif<float, 8>(z3[hh]);
      float zkr[16], zpr[16], zki[16], zpi[16], midr, midi;
      if (LDS) {
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
          for (int k3 = 0; k3 < 8; ++k3) exf[q + 64 * hh + 256 * k3] = z3[hh][brev_bits(k3, 3)].re;
        wave_sync();
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int kk = q + 64 * i; zkr[i] = exf[kk]; zpr[i] = exf[(M3 - kk) & (M3 - 1)]; }
        midr = exf[M3 / 2];
        wave_sync();
#pragma unroll
        for (int hh = 0; hh < 4; ++hh)
#pragma unroll
          for (int k3 = 0; k3 < 8; ++k3) exf[q + 64 * hh + 256 * k3] = z3[hh][brev_bits(k3, 3)].im;
        wave_sync();
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int kk = q + 64 * i; zki[i] = exf[kk]; zpi[i] = exf[(M3 - kk) & (M3 - 1)]; }
        midi = exf[M3 / 2];
        wave_sync();
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          zkr[i] = z3[i >> 2][2 * (i & 3)].re; zpr[i] = z3[i >> 2][2 * (i & 3) + 1].re;
          zki[i] = z3[i >> 2][2 * (i & 3)].im; zpi[i] = z3[i >> 2][2 * (i & 3) + 1].im;
        }
        midr = zkr[3]; midi = zki[5];
      }
      // ---- post: untangle + dB --------------------------------------------------------------------------------------------
      const float floor_pow = 1e-12f, floor_db = -120.0f;
      float lo[16], hi[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const cf e = {0.5f * (zkr[i] + zpr[i]), 0.5f * (zki[i] - zpi[i])};
        const cf d = {0.5f * (zkr[i] - zpr[i]), 0.5f * (zki[i] + zpi[i])};
        const cf o = {d.im, -d.re};
        const cf wk = ira::cmul(wl, tw[64 * i]);
        const cf pp = ira::cmul(wk, o);
        lo[i] = db_of(e.re + pp.re, e.im + pp.im, floor_pow, floor_db);
        hi[i] = db_of(e.re - pp.re, e.im - pp.im, floor_pow, floor_db);
      }
      const float mid = db_of(midr, midi, floor_pow, floor_db);
      // feed the next iteration: every result stays live through a sum (32 adds; the kernel has 33 stores there)
      float sl = mid, sh = seed;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sl += lo[i]; sh += hi[i]; }
      v[0] = {sl * 1e-3f, sh * 1e-3f};
      v[1] = {sh * 1e-3f, sl * 1e-3f};
      PIN(v[0].re); PIN(v[0].im); PIN(v[1].re); PIN(v[1].im);
      acc += mid;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float s = acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i].re + v[i].im;
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

static double g_clock_mhz = 0.0;

template <int VAR>
void run(const char* name, int threads, int valu_per_iter, int iters) {
  float* out; unsigned long long* cyc; cf* tw;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 16); hipMalloc(&tw, 2048 * sizeof(cf));
  cf* h = (cf*)malloc(2048 * sizeof(cf));
  for (int i = 0; i < 2048; ++i) { h[i].re = (float)cos(-2.0 * M_PI * i / 4096.0); h[i].im = (float)sin(-2.0 * M_PI * i / 4096.0); }
  hipMemcpy(tw, h, 2048 * sizeof(cf), hipMemcpyHostToDevice);
  free(h);
  const size_t lds = (size_t)(threads / 64) * EXC * sizeof(cf);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<VAR><<<256, threads, lds>>>(out, tw, cyc, iters, 1e-3f);
  hipEventRecord(e0);
  k<VAR><<<256, threads, lds>>>(out, tw, cyc, iters, 1e-3f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hh[2]; hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost);
  const double mhz = 100.0 * (double)hh[0] / (double)hh[1];     // s_memrealtime ticks at 100 MHz
  g_clock_mhz = mhz;
  const double winstr = 256.0 * (threads / 64) * (double)iters * valu_per_iter;
  // per CU-cycle at the shader clock measured inside this launch
  const double per_cu_cycle = winstr / (ms * 1e-3) / 256.0 / (mhz * 1e6);
  printf("%-10s waves/SIMD %d  VALU/iter %5d  %8.3f ms  clock %4.0f MHz  %7.1f wave-instr/ns  %.3f per CU-cycle  (%.2f SIMD-cycles per instruction)\n",
         name, threads / 256, valu_per_iter, ms, mhz, winstr / (ms * 1e6), per_cu_cycle, 4.0 / per_cu_cycle);
  hipFree(out); hipFree(cyc); hipFree(tw);
}

int main(int argc, char** argv) {
  // VALU instructions per loop iteration of each variant, from the device assembly (tools/micro/dft16_count.py)
  int cnt[5] = {256, 0, 0, 0, 0};
  if (argc > 1) {
    FILE* f = fopen(argv[1], "r");
    if (f) { for (int i = 0; i < 5; ++i) if (fscanf(f, "%d", &cnt[i]) != 1) break; fclose(f); }
  }
  for (int th : {256, 512, 768, 1024}) {
    run<0>("fma", th, cnt[0], 4000);
    run<1>("dft16", th, cnt[1], 8000);
    run<2>("step1", th, cnt[2], 4000);
    run<3>("frame", th, cnt[3], 400);
    run<4>("frame+lds", th, cnt[4], 400);
  }
  return 0;
}
