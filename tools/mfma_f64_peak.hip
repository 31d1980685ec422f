// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on this GPU (independent accumulators, operands in
// registers).  Used to price the AR Gram kernel against a measured ceiling rather than a datasheet number.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  d4 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
  }
  double s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int waves_per_block, int iters) {
  double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, 64 * waves_per_block>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<blocks, 64 * waves_per_block>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * waves_per_block * iters * NACC * 2048.0;
  printf("NACC=%d blocks=%d waves/block=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, waves_per_block, iters, ms,
         flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<4>(1024, 4, 20000); run<10>(1024, 4, 8000); run<10>(2048, 4, 8000); run<16>(1024, 4, 5000); run<10>(1024, 1, 8000);
  return 0;
}
