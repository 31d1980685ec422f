// Peak pick, Schroeder energy-decay curve, threshold crossings and decay-line fits.
// Compiled with -ffp-contract=off: the f64 interpolation/regression arithmetic must round like
// NumPy's (no fused multiply-add), see reference analyse/decay.py:173-260.
#include "ira_common.h"
#include "ira_log.h"

namespace {

// ------------------------------------------------------------------------------------------------
// a2: argmax |x| with first-maximum-wins.  Key = (bits(|x|) << 32) | (0xFFFFFFFF - index):
// unsigned 64-bit max picks the largest magnitude, then the smallest index.
// ------------------------------------------------------------------------------------------------
constexpr int PEAK_THREADS = 256;
constexpr int PEAK_CHUNK = 16384;  // samples per workgroup

__global__ __launch_bounds__(PEAK_THREADS) void peak_partial_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int64_t* __restrict__ len,
    unsigned long long* __restrict__ keys) {
  const int s = blockIdx.y;
  const int64_t n = len[s];
  const int64_t c0 = (int64_t)blockIdx.x * PEAK_CHUNK;
  if (c0 >= n) return;
  const int64_t c1 = (c0 + PEAK_CHUNK < n) ? c0 + PEAK_CHUNK : n;
  const float* p = x + off[s];
  unsigned long long best = 0ull;
  for (int64_t i = c0 + threadIdx.x; i < c1; i += PEAK_THREADS) {
    const float a = fabsf(p[i]);
    const unsigned long long k =
        ((unsigned long long)__float_as_uint(a) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
    best = k > best ? k : best;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(best, o, 64);
    best = other > best ? other : best;
  }
  __shared__ unsigned long long wbest[PEAK_THREADS / IRA_WAVE];
  if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < PEAK_THREADS / IRA_WAVE; ++w) best = wbest[w] > best ? wbest[w] : best;
    atomicMax(&keys[s], best);
  }
}

__global__ void peak_decode_kernel(unsigned long long* __restrict__ keys, float* __restrict__ peak_abs,
                                   int nseg) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const unsigned long long k = keys[s];
  // an all-zero (or empty) segment leaves key 0 or (0<<32 | ~i): index 0 by construction
  const uint32_t inv = (uint32_t)(k & 0xFFFFFFFFull);
  const int64_t idx = (k == 0ull) ? 0 : (int64_t)(0xFFFFFFFFu - inv);
  if (peak_abs) peak_abs[s] = __uint_as_float((uint32_t)(k >> 32));
  reinterpret_cast<int64_t*>(keys)[s] = idx;
}

// ------------------------------------------------------------------------------------------------
// a3: Schroeder EDC.  1024-thread workgroups walk 4096-sample tiles from the END of the segment towards its
// start (the direction numpy.cumsum(e[::-1]) accumulates in) with a wave-shuffle suffix scan per tile; a first
// pass collects per-chunk sums, a second re-scans with the carries, now knowing edc[0], and emits
// 10*log10(max(edc,eps)/edc[0]) floored, as float32.  Both passes run the SAME scan code.
// ------------------------------------------------------------------------------------------------
constexpr int EDC_THREADS = 1024;
constexpr int EDC_PER_THREAD = 4;
constexpr int EDC_TILE = EDC_THREADS * EDC_PER_THREAD;

struct EdcShared {
  double wave_tot[EDC_THREADS / IRA_WAVE];
  double total;
};

typedef float edc_f4 __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access, 4-byte alignment

// Suffix sums of one tile.  local index i in [0, tile_len); thread t owns i = 4t..4t+3 and reads them with one
// 16-byte load (consecutive threads = consecutive 16-byte pieces: fully coalesced, no LDS staging).
// On return s[0..3] hold the inclusive suffix sums (within the tile) at the thread's four positions and the
// function result is the suffix sum at local index 0 (thread 0's s[0], broadcast), i.e. the tile total in
// exactly the association order the emit pass uses.  Two barriers per tile.
__device__ __forceinline__ double tile_suffix_scan(const float* __restrict__ src, int tile_len, EdcShared& sh,
                                                   double s[EDC_PER_THREAD]) {
  const int t = threadIdx.x;
  const int i0 = EDC_PER_THREAD * t;
  float x[EDC_PER_THREAD];
  if (i0 + EDC_PER_THREAD <= tile_len) {
    const edc_f4 v = *reinterpret_cast<const edc_f4*>(src + i0);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
#pragma unroll
    for (int r = 0; r < EDC_PER_THREAD; ++r) x[r] = (i0 + r < tile_len) ? src[i0 + r] : 0.0f;
  }
  double e[EDC_PER_THREAD];
#pragma unroll
  for (int r = 0; r < EDC_PER_THREAD; ++r) {
    const double v = (double)x[r];
    e[r] = v * v;
  }
  s[3] = e[3];
  s[2] = e[2] + s[3];
  s[1] = e[1] + s[2];
  s[0] = e[0] + s[1];
  // inclusive suffix scan of thread totals across the wave (towards higher lanes)
  const int lane = t & 63, wave = t >> 6;
  double incl = s[0];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_down(incl, o, 64);
    if (lane + o < 64) incl += up;
  }
  double excl = __shfl_down(incl, 1, 64);  // sum over the lanes after this one
  if (lane == 63) excl = 0.0;
  if (lane == 0) sh.wave_tot[wave] = incl;
  __syncthreads();
  double later_waves = 0.0;
  for (int w = EDC_THREADS / IRA_WAVE - 1; w > wave; --w) later_waves += sh.wave_tot[w];
  excl += later_waves;
#pragma unroll
  for (int r = 0; r < EDC_PER_THREAD; ++r) s[r] += excl;
  if (t == 0) sh.total = s[0];
  __syncthreads();
  // No trailing barrier: the next tile writes wave_tot only after every thread has passed the barrier above (its
  // wave_tot reads precede it), and rewrites total only after its own first barrier (this read precedes that).
  return sh.total;
}

// The scan is split over many workgroups so that a small batch still fills the chip:
//   edc_sums_kernel   (chunks x segments)  per 16384-sample chunk (4 tiles, counted from the END of the segment):
//                     chunk total, plus the last tile's total and the local carry in front of it
//   edc_carry_kernel  (1 thread / segment) sequential carries over the chunks and the normaliser edc[0]
//   edc_emit_kernel   (chunks x segments)  re-scan with the carries and emit the dB curve
// Every value is formed as  s + (local_run + chunk_carry)  in all three kernels, so the normaliser is bit-identical
// to the value the emit pass produces at index 0 (=> edc_db[0] is exactly 0 dB, which the 0 dB crossing needs).
constexpr int EDC_CHUNK_TILES = 4;
constexpr int EDC_MAX_CHUNKS = IRA_EDC_SCRATCH_DOUBLES / 4 - 1;   // scratch: totals | last-tile totals | local carries | carries(+norm)

__global__ __launch_bounds__(EDC_THREADS) void edc_sums_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int64_t* __restrict__ len,
    double* __restrict__ scratch) {
  __shared__ EdcShared sh;
  const int seg = blockIdx.y, chunk = blockIdx.x;
  const int64_t n = len[seg];
  const int64_t ntiles = (n + EDC_TILE - 1) / EDC_TILE;
  const int64_t t0 = (int64_t)chunk * EDC_CHUNK_TILES;
  if (t0 >= ntiles) return;
  const float* src = x + off[seg];
  double* sc = scratch + (int64_t)seg * IRA_EDC_SCRATCH_DOUBLES;
  double run = 0.0, before = 0.0, tot = 0.0;
  double s[EDC_PER_THREAD];
  for (int64_t j = t0; j < t0 + EDC_CHUNK_TILES && j < ntiles; ++j) {
    const int64_t hi = n - j * EDC_TILE;
    const int64_t lo = hi - EDC_TILE > 0 ? hi - EDC_TILE : 0;
    before = run;
    tot = tile_suffix_scan(src + lo, (int)(hi - lo), sh, s);
    run = tot + run;
  }
  if (threadIdx.x == 0) {
    const int q = IRA_EDC_SCRATCH_DOUBLES / 4;
    sc[chunk] = run; sc[q + chunk] = tot; sc[2 * q + chunk] = before;
  }
}

__global__ void edc_carry_kernel(const int64_t* __restrict__ len, int nseg, double eps, double* __restrict__ scratch) {
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= nseg) return;
  const int64_t n = len[seg];
  if (n <= 0) return;
  const int q = IRA_EDC_SCRATCH_DOUBLES / 4;
  double* sc = scratch + (int64_t)seg * IRA_EDC_SCRATCH_DOUBLES;
  const int64_t ntiles = (n + EDC_TILE - 1) / EDC_TILE;
  const int nchunks = (int)((ntiles + EDC_CHUNK_TILES - 1) / EDC_CHUNK_TILES);
  double carry = 0.0;
  for (int c = 0; c < nchunks; ++c) {
    sc[3 * q + c] = carry;
    if (c == nchunks - 1) {
      // edc[0] exactly as the emit pass forms it: last tile's local-0 value + (local carry + chunk carry)
      const double v = sc[q + c] + (sc[2 * q + c] + carry);
      sc[4 * q - 1] = fmax(v, eps);
    }
    carry = sc[c] + carry;
  }
}

__global__ __launch_bounds__(EDC_THREADS) void edc_emit_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int64_t* __restrict__ len, double eps,
    double floor_db, float* __restrict__ out, double* __restrict__ out64, const int64_t* __restrict__ out_off,
    const double* __restrict__ scratch) {
  __shared__ EdcShared sh;
  __shared__ ira::LogTabEntry ltab[ira::LOGTAB_N];
  const int seg = blockIdx.y, chunk = blockIdx.x;
  const int64_t n = len[seg];
  const int64_t ntiles = (n + EDC_TILE - 1) / EDC_TILE;
  const int64_t t0 = (int64_t)chunk * EDC_CHUNK_TILES;
  if (t0 >= ntiles) return;
  ira::build_log_table(ltab, threadIdx.x);
  __syncthreads();
  const float* src = x + off[seg];
  float* dst = out ? out + out_off[seg] : nullptr;
  double* dst64 = out64 ? out64 + out_off[seg] : nullptr;
  const int q = IRA_EDC_SCRATCH_DOUBLES / 4;
  const double* sc = scratch + (int64_t)seg * IRA_EDC_SCRATCH_DOUBLES;
  const double chunk_carry = sc[3 * q + chunk];
  const double norm = sc[4 * q - 1];
  // 10 log10(v / norm) = 10 log10(2) (log2 v - log2 norm) with the table log2 (ira_log.h): ~30 instructions per sample
  // instead of ~110 for an f64 divide + log10, same value to ~1e-14 dB.  v == norm at index 0 gives exactly 0 dB.
  const bool fast = norm > 1e-300 && norm < 1e300;
  const double lnorm = fast ? ira::log2_table(norm, ltab) : 0.0;
  double run = 0.0;
  double s[EDC_PER_THREAD];
  for (int64_t j = t0; j < t0 + EDC_CHUNK_TILES && j < ntiles; ++j) {
    const int64_t hi = n - j * EDC_TILE;
    const int64_t lo = hi - EDC_TILE > 0 ? hi - EDC_TILE : 0;
    const int tl = (int)(hi - lo);
    const double tot = tile_suffix_scan(src + lo, tl, sh, s);
    const double c = run + chunk_carry;
    const int i0 = EDC_PER_THREAD * threadIdx.x;
    float o4[EDC_PER_THREAD];
#pragma unroll
    for (int r = 0; r < EDC_PER_THREAD; ++r) {
      const int i = i0 + r;
      const double v = fmax(s[r] + c, eps);
      double db;
      if (fast && v > 1e-300 && v < 1e300) db = 3.0102999566398120 * (ira::log2_table(v, ltab) - lnorm);
      else db = 10.0 * log10(v / norm);
      if (dst64 && i < tl) dst64[lo + i] = db;  // unfloored f64 (host-side optional smoothing, decay.py:161-164)
      o4[r] = (float)fmax(db, floor_db);
    }
    if (dst) {
      if (i0 + EDC_PER_THREAD <= tl) {
        const edc_f4 v = {o4[0], o4[1], o4[2], o4[3]};
        *reinterpret_cast<edc_f4*>(dst + lo + i0) = v;
      } else {
#pragma unroll
        for (int r = 0; r < EDC_PER_THREAD; ++r)
          if (i0 + r < tl) dst[lo + i0 + r] = o4[r];
      }
    }
    run = tot + run;
  }
}

// ------------------------------------------------------------------------------------------------
// a4/a5/a16: crossings + line fits on float32 dB curves.
// ------------------------------------------------------------------------------------------------
constexpr int FIT_MAX_RANGES = 4;
constexpr int FIT_MAX_CROSS = 4;
constexpr int FIT_MAX_TARGETS = 2 * FIT_MAX_RANGES + FIT_MAX_CROSS;
constexpr int FIT_U = 8;   // independent loads in flight per thread in the regression passes

struct FitParams {
  double hi[FIT_MAX_RANGES];
  double lo[FIT_MAX_RANGES];
  double cross[FIT_MAX_CROSS];
  int nranges, ncross, min_points, rel_to_peak;
  double floor_db, min_peak_above_floor;
  float t_mul, t_div;
  const float* t_axis;
};

// Time axis: either an explicit float32 array (t_axis) or the analytic axis of the reference,
// float32(i) * t_mul / t_div evaluated as two correctly rounded float32 operations.
struct TimeAxis {
  const float* axis;
  float t_mul, t_div;
  __device__ __forceinline__ float at(long long i) const {
    return axis ? axis[i] : ((float)i * t_mul) / t_div;
  }
};

struct FitShared {
  double red[3][16];
  long long idx[FIT_MAX_TARGETS];
  double tgt[FIT_MAX_TARGETS];
};

__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, FitShared& sh) {
  a = ira::wave_sum(a); b = ira::wave_sum(b); c = ira::wave_sum(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) { sh.red[0][wave] = a; sh.red[1][wave] = b; sh.red[2][wave] = c; }
  __syncthreads();
  a = b = c = 0.0;
  for (int w = 0; w < nw; ++w) { a += sh.red[0][w]; b += sh.red[1][w]; c += sh.red[2][w]; }
}

// first index i with y_rel[i] <= target  (float32 compare, like `curve <= target` under NEP 50)
__device__ __forceinline__ double crossing_time_from_index(const float* y, float peak, long long idx, long long n,
                                                           double target, const TimeAxis& ta) {
  if (idx >= n) return __longlong_as_double(0x7ff8000000000000ll);  // NaN = "no crossing"
  if (idx == 0) return (double)ta.at(0);
  const double t0 = (double)ta.at(idx - 1);
  const double t1 = (double)ta.at(idx);
  const double y0 = (double)(y[idx - 1] - peak);
  const double y1 = (double)(y[idx] - peak);
  if (y1 == y0) return t1;
  double frac = (target - y0) / (y1 - y0);
  frac = fmin(fmax(frac, 0.0), 1.0);
  return t0 + frac * (t1 - t0);
}

// Long curves (decay / band EDCs of whole files): the first-crossing search is done by MANY workgroups per curve
// (grid chunks x curves), each taking the minimum crossing index of its chunk into an index slot per target with
// atomicMin; the per-curve kernel below then starts from those indices.  One workgroup sweeping a 480 k-sample
// curve whose -35 dB point is never reached (band-limited EDCs flatten on their wrapped pre-ringing) was the
// latency-bound worst case of curve_fit_kernel.  The slots live in the first ntargets doubles of the curve's
// fit_out record (cross_out when there are no ranges); they are memset to 0x7f.. (a huge positive int64 = "none").
constexpr int XS_THREADS = 256;
constexpr int XS_PER_THREAD = 32;
constexpr int XS_CHUNK = XS_THREADS * XS_PER_THREAD;

__global__ __launch_bounds__(XS_THREADS) void crossing_search_kernel(const float* __restrict__ ybase,
                                                                     const int64_t* __restrict__ off,
                                                                     const int64_t* __restrict__ len, FitParams P,
                                                                     unsigned long long* __restrict__ slots,
                                                                     int slot_stride) {
  const int c = blockIdx.y;
  const long long n = len[c];
  const long long base = (long long)blockIdx.x * XS_CHUNK;
  if (base >= n) return;
  const float* y = ybase + off[c];
  const int tid = threadIdx.x;
  const int ntargets = 2 * P.nranges + P.ncross;
  float targets[FIT_MAX_TARGETS];
#pragma unroll
  for (int k = 0; k < FIT_MAX_TARGETS; ++k) {
    double tv = 0.0;
    if (k < 2 * P.nranges) tv = (k & 1) ? P.lo[k >> 1] : P.hi[k >> 1];
    else if (k < ntargets) tv = P.cross[k - 2 * P.nranges];
    targets[k] = (float)tv;
  }
  long long first[FIT_MAX_TARGETS];
#pragma unroll
  for (int k = 0; k < FIT_MAX_TARGETS; ++k) first[k] = n;
  float v[XS_PER_THREAD];
#pragma unroll
  for (int u = 0; u < XS_PER_THREAD; ++u) {
    const long long i = base + tid + (long long)XS_THREADS * u;
    v[u] = i < n ? y[i] : INFINITY;
  }
#pragma unroll
  for (int u = XS_PER_THREAD - 1; u >= 0; --u) {        // descending: the smallest index wins without a compare
    const long long i = base + tid + (long long)XS_THREADS * u;
#pragma unroll
    for (int k = 0; k < FIT_MAX_TARGETS; ++k)
      if (k < ntargets && v[u] <= targets[k]) first[k] = i;
  }
  unsigned long long* sl = slots + (long long)c * slot_stride;
#pragma unroll
  for (int k = 0; k < FIT_MAX_TARGETS; ++k) {
    if (k < ntargets && __any(first[k] < n)) {          // wave-uniform: crossings are rare events
      long long f = first[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const long long other = __shfl_xor(f, o, 64);
        f = other < f ? other : f;
      }
      if ((tid & 63) == 0) atomicMin(&sl[k], (unsigned long long)f);
    }
  }
}

__global__ void curve_fit_kernel(const float* __restrict__ ybase, const int64_t* __restrict__ off,
                                 const int64_t* __restrict__ len, FitParams P, double* __restrict__ fit_out,
                                 double* __restrict__ cross_out, int pre_searched) {
  __shared__ FitShared sh;
  const int c = blockIdx.x;
  const long long n = len[c];
  const float* y = ybase + off[c];
  const int tid = threadIdx.x, nt = blockDim.x;
  const double qnan = __longlong_as_double(0x7ff8000000000000ll);
  const TimeAxis ta{P.t_axis, P.t_mul, P.t_div};
  double* fo = fit_out + (int64_t)c * P.nranges * IRA_FIT_DOUBLES;
  double* co = cross_out ? cross_out + (int64_t)c * P.ncross : nullptr;

  // ---- optional normalisation to the curve's own peak (modal cloud) -------------------------------------
  float peak = 0.0f;
  bool usable = n > 0;
  if (P.rel_to_peak) {
    float m = -INFINITY;
    int bad = 0;
    for (long long i = tid; i < n; i += nt) {
      const float v = y[i];
      if (!isfinite(v)) bad = 1;
      m = fmaxf(m, v);
    }
    m = ira::wave_max(m);
    bad = __any(bad) ? 1 : 0;
    if ((tid & 63) == 0) { sh.red[0][tid >> 6] = (double)m; sh.red[1][tid >> 6] = (double)bad; }
    __syncthreads();
    float pk = -INFINITY;
    int anybad = 0;
    for (int w = 0; w < ((nt + 63) >> 6); ++w) {
      pk = fmaxf(pk, (float)sh.red[0][w]);
      anybad |= (sh.red[1][w] != 0.0);
    }
    __syncthreads();
    peak = pk;
    if (anybad) usable = false;
    if (usable && ((double)peak - P.floor_db) < P.min_peak_above_floor) usable = false;
  }
  if (!usable) {
    if (tid == 0) {
      for (int r = 0; r < P.nranges; ++r)
        for (int k = 0; k < IRA_FIT_DOUBLES; ++k) fo[r * IRA_FIT_DOUBLES + k] = (k == 0) ? 0.0 : qnan;
      for (int j = 0; j < P.ncross && co; ++j) co[j] = qnan;
    }
    return;
  }

  // ---- first-crossing indices for every target, one sweep ----------------------------------------------
  // Fixed-size, fully unrolled arrays so they stay in registers (runtime-indexed arrays go to scratch).
  const int ntargets = 2 * P.nranges + P.ncross;
  if (tid < FIT_MAX_TARGETS) {
    double tv = 0.0;
    if (tid < 2 * P.nranges) tv = (tid & 1) ? P.lo[tid >> 1] : P.hi[tid >> 1];
    else if (tid < ntargets) tv = P.cross[tid - 2 * P.nranges];
    sh.tgt[tid] = tv;
    long long start = n;
    if (pre_searched && tid < ntargets) {
      // indices found by crossing_search_kernel (slots alias the head of this curve's output record)
      const unsigned long long* sl = reinterpret_cast<const unsigned long long*>(P.nranges > 0 ? fo : co);
      const unsigned long long f = sl[tid];
      start = f < (unsigned long long)n ? (long long)f : n;
    }
    sh.idx[tid] = start;
  }
  __syncthreads();
  float targets[FIT_MAX_TARGETS];
#pragma unroll
  for (int k = 0; k < FIT_MAX_TARGETS; ++k) targets[k] = (float)sh.tgt[k];
  // Swept in blocks with an early exit: once every target has a crossing, later samples cannot change any FIRST
  // crossing index, so the rest of the curve need not be read (for an EDC the -35 dB point sits in the first
  // few percent of a 10 s curve).  A target that is never reached still scans to the end, like the reference.
  constexpr int SWEEP_U = 8;
  for (long long base = 0; base < n && !pre_searched; base += (long long)nt * SWEEP_U) {
    long long first[FIT_MAX_TARGETS];
#pragma unroll
    for (int k = 0; k < FIT_MAX_TARGETS; ++k) first[k] = n;
#pragma unroll
    for (int u = 0; u < SWEEP_U; ++u) {
      const long long i = base + tid + (long long)nt * u;
      if (i < n) {
        const float v = y[i] - peak;
#pragma unroll
        for (int k = 0; k < FIT_MAX_TARGETS; ++k)
          if (k < ntargets && v <= targets[k] && i < first[k]) first[k] = i;
      }
    }
#pragma unroll
    for (int k = 0; k < FIT_MAX_TARGETS; ++k) {
      if (k < ntargets) {
        long long f = first[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const long long other = __shfl_xor(f, o, 64);
          f = other < f ? other : f;
        }
        if ((tid & 63) == 0 && f < n) atomicMin(&sh.idx[k], f);
      }
    }
    __syncthreads();
    bool all_found = true;
    for (int k = 0; k < ntargets; ++k) all_found = all_found && (sh.idx[k] < n);
    __syncthreads();
    if (all_found) break;
  }
  const double* targets_d = sh.tgt;

  if (co && tid == 0) {
    for (int j = 0; j < P.ncross; ++j) {
      const int k = 2 * P.nranges + j;
      co[j] = crossing_time_from_index(y, peak, sh.idx[k], n, targets_d[k], ta);
    }
  }

  // ---- per range: crossing times -> float32 mask -> two-pass regression ----------------------------------
  for (int r = 0; r < P.nranges; ++r) {
    const long long i_hi = sh.idx[2 * r], i_lo = sh.idx[2 * r + 1];
    const double ts = crossing_time_from_index(y, peak, i_hi, n, targets_d[2 * r], ta);
    const double te = crossing_time_from_index(y, peak, i_lo, n, targets_d[2 * r + 1], ta);
    double* o = fo + r * IRA_FIT_DOUBLES;
    bool ok = !(isnan(ts) || isnan(te) || te <= ts);
    if (!ok) {
      if (tid == 0) { o[0] = 0.0; o[1] = ts; o[2] = te; for (int k = 3; k < IRA_FIT_DOUBLES; ++k) o[k] = qnan; }
      continue;
    }
    const float ts32 = (float)ts, te32 = (float)te;
    long long a0 = i_hi - 2; if (a0 < 0) a0 = 0;
    long long a1 = i_lo + 2; if (a1 > n - 1) a1 = n - 1;
    // pass 1: count, sum t, sum y
    // The three passes read y in batches of FIT_U independent loads per thread: with one load per iteration a single
    // workgroup streaming a long range is bound by one memory latency per element.
    double cnt = 0.0, st = 0.0, sy = 0.0;
    for (long long b0 = a0; b0 <= a1; b0 += (long long)nt * FIT_U) {
      float yv[FIT_U];
#pragma unroll
      for (int u = 0; u < FIT_U; ++u) {
        const long long i = b0 + tid + (long long)nt * u;
        yv[u] = i <= a1 ? y[i] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < FIT_U; ++u) {
        const long long i = b0 + tid + (long long)nt * u;
        if (i > a1) continue;
        const float tf = ta.at(i);
        if (tf >= ts32 && tf <= te32) { cnt += 1.0; st += (double)tf; sy += (double)(yv[u] - peak); }
      }
    }
    block_sum3(cnt, st, sy, sh);
    const long long npts = (long long)cnt;
    if (npts < P.min_points) {
      if (tid == 0) { o[0] = 0.0; o[1] = ts; o[2] = te; for (int k = 3; k < 7; ++k) o[k] = qnan; o[7] = (double)npts; }
      continue;
    }
    const double tm = st / cnt, ym = sy / cnt;
    // pass 2: centred second moments
    double stt = 0.0, sty = 0.0, syy = 0.0;
    for (long long b0 = a0; b0 <= a1; b0 += (long long)nt * FIT_U) {
      float yv[FIT_U];
#pragma unroll
      for (int u = 0; u < FIT_U; ++u) {
        const long long i = b0 + tid + (long long)nt * u;
        yv[u] = i <= a1 ? y[i] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < FIT_U; ++u) {
        const long long i = b0 + tid + (long long)nt * u;
        if (i > a1) continue;
        const float tf = ta.at(i);
        if (tf >= ts32 && tf <= te32) {
          const double dt = (double)tf - tm, dy = (double)(yv[u] - peak) - ym;
          stt += dt * dt; sty += dt * dy; syy += dy * dy;
        }
      }
    }
    block_sum3(stt, sty, syy, sh);
    const double slope = sty / stt;
    const double icpt = ym - slope * tm;
    // pass 3: residual sum of squares against the fitted line (decay.py:244-247)
    double sres = 0.0, d1 = 0.0, d2 = 0.0;
    for (long long b0 = a0; b0 <= a1; b0 += (long long)nt * FIT_U) {
      float yv[FIT_U];
#pragma unroll
      for (int u = 0; u < FIT_U; ++u) {
        const long long i = b0 + tid + (long long)nt * u;
        yv[u] = i <= a1 ? y[i] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < FIT_U; ++u) {
        const long long i = b0 + tid + (long long)nt * u;
        if (i > a1) continue;
        const float tf = ta.at(i);
        if (tf >= ts32 && tf <= te32) {
          const double e = (double)(yv[u] - peak) - (slope * (double)tf + icpt);
          sres += e * e;
        }
      }
    }
    block_sum3(sres, d1, d2, sh);
    if (tid == 0) {
      const bool neg = slope < 0.0;  // also false for NaN (stt == 0)
      o[0] = neg ? 1.0 : 0.0;
      o[1] = ts; o[2] = te; o[3] = slope; o[4] = icpt;
      o[5] = syy > 0.0 ? 1.0 - sres / syy : 0.0;
      o[6] = -60.0 / slope;
      o[7] = (double)npts;
    }
  }
}

}  // namespace

extern "C" int32_t ira_peak_index(const float* x_dev, const int64_t* off_dev, const int64_t* len_dev, int32_t nseg,
                                  int64_t* peak_dev, float* peak_abs_dev, void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(len_dev); IRA_CHECK_PTR(peak_dev);
  if (nseg <= 0) return nseg == 0 ? IRA_OK : IRA_E_SIZE;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(peak_dev, 0, sizeof(int64_t) * (size_t)nseg, st);
  if (e != hipSuccess) return ira_hip_status(e);
  // chunk count is sized for the longest supported segment (2^31 samples would be 131072 chunks); the host
  // passes lengths on the device only, so launch a fixed generous grid and let empty chunks exit at once.
  const int max_chunks = 2048;  // 33.5 M samples per segment
  peak_partial_kernel<<<dim3(max_chunks, nseg), PEAK_THREADS, 0, st>>>(
      x_dev, off_dev, len_dev, reinterpret_cast<unsigned long long*>(peak_dev));
  peak_decode_kernel<<<(nseg + 255) / 256, 256, 0, st>>>(reinterpret_cast<unsigned long long*>(peak_dev),
                                                           peak_abs_dev, nseg);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_edc_db(const float* x_dev, const int64_t* off_dev, const int64_t* len_dev, int32_t nseg,
                              int64_t max_len, double eps, double floor_db, float* edc_db_dev, double* edc_db64_dev,
                              const int64_t* edc_off_dev, double* scratch_dev, void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(len_dev);
  if (edc_db_dev == nullptr && edc_db64_dev == nullptr) return IRA_E_NULL;
  IRA_CHECK_PTR(edc_off_dev); IRA_CHECK_PTR(scratch_dev);
  if (nseg <= 0) return nseg == 0 ? IRA_OK : IRA_E_SIZE;
  if (max_len <= 0 || max_len > (int64_t)EDC_MAX_CHUNKS * EDC_CHUNK_TILES * EDC_TILE) return IRA_E_SIZE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = (max_len + EDC_TILE - 1) / EDC_TILE;
  const int nchunks = (int)((ntiles + EDC_CHUNK_TILES - 1) / EDC_CHUNK_TILES);
  if (nseg > 65535) return IRA_E_SIZE;
  edc_sums_kernel<<<dim3(nchunks, nseg), EDC_THREADS, 0, st>>>(x_dev, off_dev, len_dev, scratch_dev);
  edc_carry_kernel<<<(nseg + 63) / 64, 64, 0, st>>>(len_dev, nseg, eps, scratch_dev);
  edc_emit_kernel<<<dim3(nchunks, nseg), EDC_THREADS, 0, st>>>(x_dev, off_dev, len_dev, eps, floor_db, edc_db_dev,
                                                               edc_db64_dev, edc_off_dev, scratch_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_curve_fits(const float* y_dev, const int64_t* off_dev, const int64_t* len_dev,
                                  int32_t ncurves, int32_t max_len, float t_mul, float t_div,
                                  const float* t_axis_dev, const double* ranges_hi_lo, int32_t nranges, int32_t min_points,
                                  const double* cross_targets, int32_t ncross, int32_t rel_to_peak, double floor_db,
                                  double min_peak_above_floor, double* fit_out_dev, double* cross_out_dev,
                                  void* stream) {
  IRA_CHECK_PTR(y_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(len_dev);
  if (nranges < 0 || nranges > FIT_MAX_RANGES || ncross < 0 || ncross > FIT_MAX_CROSS) return IRA_E_SIZE;
  if (nranges > 0) { IRA_CHECK_PTR(ranges_hi_lo); IRA_CHECK_PTR(fit_out_dev); }
  if (ncross > 0) { IRA_CHECK_PTR(cross_targets); IRA_CHECK_PTR(cross_out_dev); }
  if (ncurves <= 0) return ncurves == 0 ? IRA_OK : IRA_E_SIZE;
  FitParams P{};
  for (int r = 0; r < nranges; ++r) { P.hi[r] = ranges_hi_lo[2 * r]; P.lo[r] = ranges_hi_lo[2 * r + 1]; }
  for (int j = 0; j < ncross; ++j) P.cross[j] = cross_targets[j];
  P.nranges = nranges; P.ncross = ncross; P.min_points = min_points; P.rel_to_peak = rel_to_peak;
  P.floor_db = floor_db; P.min_peak_above_floor = min_peak_above_floor; P.t_mul = t_mul; P.t_div = t_div; P.t_axis = t_axis_dev;
  const int threads = max_len <= 2048 ? 64 : (max_len <= 32768 ? 256 : 1024);
  hipStream_t st = (hipStream_t)stream;
  const int ntargets = 2 * nranges + ncross;
  int pre = 0;
  if (max_len > 4 * XS_CHUNK && !rel_to_peak && ntargets > 0 && ncurves <= 65535) {
    // parallel crossing search for long curves; slots = head of each curve's output record (see the kernel comment)
    double* slot_base = nranges > 0 ? fit_out_dev : cross_out_dev;
    const int slot_stride = nranges > 0 ? nranges * IRA_FIT_DOUBLES : ncross;
    hipError_t e = hipMemsetAsync(slot_base, 0x7f, sizeof(double) * (size_t)slot_stride * (size_t)ncurves, st);
    if (e != hipSuccess) return ira_hip_status(e);
    crossing_search_kernel<<<dim3((max_len + XS_CHUNK - 1) / XS_CHUNK, ncurves), XS_THREADS, 0, st>>>(
        y_dev, off_dev, len_dev, P, reinterpret_cast<unsigned long long*>(slot_base), slot_stride);
    pre = 1;
  }
  curve_fit_kernel<<<ncurves, threads, 0, st>>>(y_dev, off_dev, len_dev, P, fit_out_dev,
                                               ncross > 0 ? cross_out_dev : nullptr, pre);
  IRA_RETURN_LAUNCH();
}
