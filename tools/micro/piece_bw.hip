// Micro-benchmark: what does the four-step FFT's column access pattern cost by itself?
// A "job" is a rows x cols matrix of 16-byte values (cols*16 bytes per row).  A workgroup moves a tile of `rows` x C values:
//   mode 0: strided read (pieces of C*16 bytes, one per row) -> contiguous write
//   mode 1: contiguous read -> strided write
//   mode 2: strided read -> strided write (other tiling, like pass 1's output)
// `lds_kb` of dynamic LDS per workgroup only limits residency (as the FFT kernels' tiles do).
// Build: hipcc --offload-arch=gfx950 -O3 -o piece_bw piece_bw.hip ; run: ./piece_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(16) V { double a, b; };

template <int U>
__global__ __launch_bounds__(256) void move_kernel(const V* __restrict__ in, V* __restrict__ out, int rows, int cols, int C,
                                                   int mode, int remap) {
  extern __shared__ unsigned char smem[];
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (remap) {
    const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
    const unsigned orig = blockIdx.y * gx + blockIdx.x;
    const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    const unsigned wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    bx = wg % gx; by = wg / gx;
  }
  const long long job = (long long)by * rows * cols;
  const int total = rows * C, tid = threadIdx.x;
  const int c0 = bx * C;
  for (int base = 0; base < total; base += 256 * U) {
    V raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int i = base + tid + 256 * u;
      i = i < total ? i : total - 1;
      const long long src = (mode == 1) ? job + (long long)bx * total + i : job + (long long)(i / C) * cols + c0 + i % C;
      raw[u] = in[src];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + tid + 256 * u;
      if (i >= total) continue;
      const long long dst = (mode == 0) ? job + (long long)bx * total + i : job + (long long)(i / C) * cols + c0 + i % C;
      V v = raw[u];
      v.a += 1.0;
      out[dst] = v;
    }
  }
  if (smem[0] == 123 && tid == 999) out[0].a = 0;   // keep the LDS allocation
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 640, cols = argc > 2 ? atoi(argv[2]) : 750, jobs = argc > 3 ? atoi(argv[3]) : 384;
  const size_t n = (size_t)rows * cols * jobs;
  V *in, *out;
  hipMalloc(&in, n * sizeof(V)); hipMalloc(&out, n * sizeof(V));
  hipMemset(in, 0, n * sizeof(V)); hipMemset(out, 0, n * sizeof(V));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("rows %d cols %d jobs %d: %.2f GB each way\n", rows, cols, jobs, n * 16 / 1e9);
  for (int mode = 0; mode < 3; ++mode)
    for (int lds_kb : {0, 20, 32, 48})
      for (int C : {1, 2, 4, 8, 5, 10, 25}) {
        if (cols % C) continue;
        for (int remap = 0; remap < 2; ++remap) {
          dim3 grid(cols / C, jobs);
          float best = 1e9;
          for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0);
            move_kernel<5><<<grid, 256, lds_kb * 1024>>>(in, out, rows, cols, C, mode, remap);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it && ms < best) best = ms;
          }
          printf("mode %d lds %2d KB C %2d (%3d B pieces) remap %d: %.3f ms  %.2f TB/s (read+write)\n", mode, lds_kb, C, C * 16,
                 remap, best, 2.0 * n * 16 / best / 1e9);
          fflush(stdout);
        }
      }
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
