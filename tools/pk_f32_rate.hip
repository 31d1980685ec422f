// Issue-rate microbenchmark: scalar vs packed f32 VALU on gfx950 (one wave per SIMD and two waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/pk_f32_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = {(float)threadIdx.x + i, (float)i};
  f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        if (MODE == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b.x));
        if (MODE == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
        if (MODE == 6) asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a[i]) : "v"(b));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 16);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e0);
  k<MODE><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long hh[2]; hipMemcpy(hh, cyc, 16, hipMemcpyDeviceToHost); unsigned long long h = hh[0];
  printf("[shader clock %.0f MHz by memtime/memrealtime] ", 100.0 * (double)hh[0] / (double)hh[1]);
  const double winstr = 256.0 * (threads / 64) * iters * 32.0;      // wave-instructions in the launch
  printf("[%.3f ms, %.1f ticks/us, %.2f wave-instr/ns chip = %.3f per CU-cycle@2.4GHz] ", ms, h / (ms * 1e3),
         winstr / (ms * 1e6), winstr / (ms * 1e6) / 256 / 2.4);
  // s_memtime counts at 100 MHz; convert with an assumed 2.4 GHz shader clock
  printf("%-34s waves/SIMD %d: %.2f memtime-ticks per instr (x24 = %.1f shader cycles @2.4GHz)\n", name, threads / 256,
         (double)h / (iters * 32.0), 24.0 * (double)h / (iters * 32.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int th : {256, 512, 768, 1024}) {
    run<0>("v_fma_f32", th); run<1>("v_pk_fma_f32", th); run<2>("v_add_f32", th); run<3>("v_pk_add_f32", th);
    run<4>("v_pk_mul_f32", th); run<5>("v_pk_fma_f32 op_sel swap + neg_lo", th); run<6>("v_pk_add_f32 neg (sub)", th);
  }
  return 0;
}
