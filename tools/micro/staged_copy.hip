// Micro-benchmark: what can a "tile through LDS" kernel move at all?  The skeleton of every long-FFT pass of libira:
//   load a contiguous tile (T bytes, 16 bytes per lane and load, U loads in flight) -> LDS -> barrier [-> P extra rounds of
//   LDS read / write / barrier standing in for radix passes] -> read back -> contiguous store.
// Sweeps the tile size (= workgroups per CU through the LDS footprint), the number of stand-in passes and, for reference,
// a plain copy with no LDS.  Build: hipcc --offload-arch=gfx950 -O3 -o staged_copy staged_copy.hip ; run: ./staged_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct alignas(16) V { double a, b; };

template <int U>
__global__ __launch_bounds__(256) void staged_kernel(const V* __restrict__ in, V* __restrict__ out, int tile_elems, int passes,
                                                     int use_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  V* lds = reinterpret_cast<V*>(smem);
  const int tid = threadIdx.x;
  const long long base = (long long)blockIdx.x * tile_elems;
  for (int b = 0; b < tile_elems; b += 256 * U) {
    V raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int i = b + tid + 256 * u;
      i = i < tile_elems ? i : tile_elems - 1;
      raw[u] = in[base + i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int i = b + tid + 256 * u;
      i = i < tile_elems ? i : tile_elems - 1;
      if (use_lds) lds[i] = raw[u];
      else { V v = raw[u]; v.a += 1.0; if (b + tid + 256 * u < tile_elems) out[base + i] = v; }
    }
  }
  if (!use_lds) return;
  __syncthreads();
  for (int p = 0; p < passes; ++p) {                      // stand-in for a radix pass: every element read, changed, written
    for (int i = tid; i < tile_elems; i += 256) {
      V v = lds[(i * 17 + p) % tile_elems];
      v.a = v.a * 1.0000001 + v.b;
      lds[(i * 17 + p) % tile_elems] = v;
    }
    __syncthreads();
  }
  for (int i = tid; i < tile_elems; i += 256) {
    V v = lds[i];
    v.a += 1.0;
    out[base + i] = v;
  }
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30);
  const size_t n = bytes / sizeof(V);
  V *in, *out;
  hipMalloc(&in, n * sizeof(V)); hipMalloc(&out, n * sizeof(V));
  hipMemset(in, 0, n * sizeof(V)); hipMemset(out, 0, n * sizeof(V));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%.2f GB read + %.2f GB written per launch\n", n * 16 / 1e9, n * 16 / 1e9);
  const int tiles_kb[] = {8, 16, 20, 24, 32, 48, 64};
  for (int use_lds = 0; use_lds <= 1; ++use_lds)
    for (int passes : {0, 3, 6}) {
      if (!use_lds && passes) continue;
      for (int kb : tiles_kb) {
        const int tile_elems = kb * 1024 / 16;
        const unsigned grid = (unsigned)(n / tile_elems);
        const size_t lds = use_lds ? (size_t)kb * 1024 : 0;
        hipFuncSetAttribute(reinterpret_cast<const void*>(staged_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        for (int w = 0; w < 2; ++w) staged_kernel<8><<<grid, 256, lds>>>(in, out, tile_elems, passes, use_lds);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) staged_kernel<8><<<grid, 256, lds>>>(in, out, tile_elems, passes, use_lds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%s passes %d tile %2d KB (%2d workgroups per CU by LDS): %.3f ms  %.2f TB/s (read + write)\n",
               use_lds ? "through LDS" : "plain copy ", passes, kb, use_lds ? (int)(160 / (kb + 0.001)) : 8, ms,
               2.0 * n * 16 / (ms * 1e-3) / 1e12);
      }
    }
  return 0;
}
