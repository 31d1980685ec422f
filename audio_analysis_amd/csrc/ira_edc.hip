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

__device__ __forceinline__ unsigned long long peak_key(float v, int64_t i) {
  return ((unsigned long long)__float_as_uint(fabsf(v)) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
}

// Sixteen-byte loads, all of a thread's loads in flight together: the chunk is cut into a scalar head up to the first
// 16-byte boundary (segments start anywhere: peak-trimmed offsets), PEAK_CHUNK / 4 - 1 aligned float4 values and a scalar
// tail.  (Round 3 read one float per lane and iteration: 2.9 TB/s on a kernel that does nothing but read 4N bytes.)
__global__ __launch_bounds__(PEAK_THREADS) void peak_partial_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int64_t* __restrict__ len,
    unsigned long long* __restrict__ keys) {
  const int s = blockIdx.y;
  const int64_t n = len[s];
  const int64_t c0 = (int64_t)blockIdx.x * PEAK_CHUNK;
  if (c0 >= n) return;
  const int64_t c1 = (c0 + PEAK_CHUNK < n) ? c0 + PEAK_CHUNK : n;
  const float* p = x + off[s];
  const int tid = threadIdx.x;
  const int cnt = (int)(c1 - c0);
  const float* q = p + c0;
  int head = (int)(((16u - (unsigned)((uintptr_t)q & 15u)) & 15u) >> 2);
  head = head < cnt ? head : cnt;
  const int nvec = (cnt - head) >> 2;
  const float4* v4 = reinterpret_cast<const float4*>(q + head);
  unsigned long long best = 0ull;
  constexpr int PU = PEAK_CHUNK / 4 / PEAK_THREADS;          // 16 vector loads per thread
  if (nvec > 0) {                                             // (workgroup-uniform: a chunk of fewer than four aligned samples
    float4 a[PU];                                             //  has no vector part and must not touch memory past its end)
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int j = tid + PEAK_THREADS * u;
      a[u] = v4[j < nvec ? j : nvec - 1];                     // clamped: unconditional loads, all in flight together
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int j = tid + PEAK_THREADS * u;
      if (j < nvec) {
        const int64_t i = c0 + head + 4 * (int64_t)j;
        unsigned long long k;
        k = peak_key(a[u].x, i);     best = k > best ? k : best;
        k = peak_key(a[u].y, i + 1); best = k > best ? k : best;
        k = peak_key(a[u].z, i + 2); best = k > best ? k : best;
        k = peak_key(a[u].w, i + 3); best = k > best ? k : best;
      }
    }
  }
  if (tid < head) {
    const unsigned long long k = peak_key(q[tid], c0 + tid);
    best = k > best ? k : best;
  }
  {
    const int t0 = head + 4 * nvec + tid;                     // at most three tail samples
    if (t0 < cnt) {
      const unsigned long long k = peak_key(q[t0], c0 + t0);
      best = k > best ? k : best;
    }
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
// a3: Schroeder EDC.  A segment is cut into 4096-sample tiles counted from its END (the direction
// numpy.cumsum(e[::-1]) accumulates in); one 256-thread workgroup scans one tile: every thread owns 16 consecutive
// samples (four 16-byte loads issued together: one memory round trip per tile), forms their suffix sums serially,
// the wave combines thread totals with shuffles and the four waves meet through LDS -- one barrier per tile.
//   edc_sums_kernel   (tiles x segments)   tile totals
//   edc_carry_kernel  (1 wave / segment)   carry[j] = energy behind tile j, and the normaliser edc[0]
//   edc_emit_kernel   (tiles x segments)   re-scan, add the carry, 10*log10(max(edc,eps)/edc[0]) floored -> float32
// Every kernel (the fused fit kernel below included) runs the SAME scan code and forms a value as
// (suffix sum inside the tile) + carry[j], so the normaliser is bit-identical to the value the emit pass produces at
// index 0 (=> edc_db[0] is exactly 0 dB, which the 0 dB crossing needs) and the fit kernel sees the emitted curve.
// ------------------------------------------------------------------------------------------------
constexpr int EDC_THREADS = 256;
constexpr int EDC_PER_THREAD = 16;
constexpr int EDC_TILE = EDC_THREADS * EDC_PER_THREAD;
constexpr int EDC_WAVES = EDC_THREADS / IRA_WAVE;
// scratch per segment: tot[0 .. T) | carry[0 .. T) | ... | norm (last double); T <= EDC_MAX_TILES
constexpr int EDC_MAX_TILES = IRA_EDC_SCRATCH_DOUBLES / 2 - 1;
static_assert(EDC_TILE == 4096, "the host sizes its checks with 4096-sample tiles");

struct EdcShared {
  double wave_tot[2][EDC_WAVES];     // double-buffered by tile parity: one barrier per tile is enough
};

typedef float edc_f4 __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access, 4-byte alignment

// The thread's 16 samples of a tile of tile_len valid samples (zeros beyond): local indices 16t .. 16t+15.
__device__ __forceinline__ void tile_load(const float* __restrict__ src, int tile_len, float x[EDC_PER_THREAD]) {
  const int i0 = EDC_PER_THREAD * threadIdx.x;
  if (i0 + EDC_PER_THREAD <= tile_len) {
#pragma unroll
    for (int j = 0; j < EDC_PER_THREAD / 4; ++j) {
      const edc_f4 v = *reinterpret_cast<const edc_f4*>(src + i0 + 4 * j);
      x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < EDC_PER_THREAD; ++r) x[r] = (i0 + r < tile_len) ? src[i0 + r] : 0.0f;
  }
}

// Inclusive suffix sums (within the tile) of the squared samples at the thread's 16 positions.  parity = tile
// counter & 1 of the calling loop (selects the LDS buffer).  s[0] of thread 0 is the tile total.
__device__ __forceinline__ void tile_suffix_scan(const float x[EDC_PER_THREAD], EdcShared& sh, int parity,
                                                 double s[EDC_PER_THREAD]) {
  const int t = threadIdx.x;
  {
    const double v = (double)x[EDC_PER_THREAD - 1];
    s[EDC_PER_THREAD - 1] = v * v;
  }
#pragma unroll
  for (int r = EDC_PER_THREAD - 2; r >= 0; --r) {
    const double v = (double)x[r];
    s[r] = v * v + s[r + 1];
  }
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
  if (lane == 0) sh.wave_tot[parity][wave] = incl;
  __syncthreads();
  double later_waves = 0.0;
#pragma unroll
  for (int w = EDC_WAVES - 1; w > 0; --w)
    if (w > wave) later_waves += sh.wave_tot[parity][w];
  excl += later_waves;
#pragma unroll
  for (int r = 0; r < EDC_PER_THREAD; ++r) s[r] += excl;
}

__global__ __launch_bounds__(EDC_THREADS) void edc_sums_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int64_t* __restrict__ len,
    double* __restrict__ scratch, const int64_t* __restrict__ part_off) {
  __shared__ EdcShared sh;
  const int seg = blockIdx.y;
  const int64_t j = blockIdx.x;
  const int64_t n = len[seg];
  const int64_t ntiles = (n + EDC_TILE - 1) / EDC_TILE;
  if (j >= ntiles) return;
  // segments whose producer delivered tile energies (edc_tiles_from_parts_kernel) scan only the tile that holds their first
  // sample: the normaliser edc[0] = tot[T-1] + carry[T-1] must be the value the emit / fit scans form at index 0
  if (part_off != nullptr && part_off[seg] >= 0 && j < ntiles - 1) return;
  const int64_t hi = n - j * EDC_TILE;
  const int64_t lo = hi - EDC_TILE > 0 ? hi - EDC_TILE : 0;
  float xv[EDC_PER_THREAD];
  tile_load(x + off[seg] + lo, (int)(hi - lo), xv);
  double s[EDC_PER_THREAD];
  tile_suffix_scan(xv, sh, 0, s);
  if (threadIdx.x == 0) scratch[(int64_t)seg * IRA_EDC_SCRATCH_DOUBLES + j] = s[0];
}

// Tile totals from the partial energies the band inverse's second pass left behind (ira_fftsmooth.hip, band_tile_partials):
// part[part_off[seg] + w * tiles + j], w < wgs, is workgroup w's share of the energy of the signal's tile j (tiles of EDC_TILE
// samples counted from the END of the signal -- which is the end of the segment, so segment tile j IS signal tile j for every
// tile but the one that holds the segment's first sample; `tiles` = the SIGNAL's tile count, which the segment's last sample
// index gives: the signal is the segment plus what was trimmed in front of it ... the caller passes it per segment).
// One thread per tile adds the wgs partials in ascending order (coalesced: the threads of a workgroup read neighbouring
// tiles of the same producer workgroup): the total does not depend on scheduling.
__global__ __launch_bounds__(128) void edc_tiles_from_parts_kernel(const int64_t* __restrict__ len,
                                                                   const double* __restrict__ part,
                                                                   const int64_t* __restrict__ part_off,
                                                                   const int32_t* __restrict__ part_wgs,
                                                                   const int32_t* __restrict__ part_tiles,
                                                                   double* __restrict__ scratch) {
  const int seg = blockIdx.y;
  const int64_t n = len[seg];
  const int64_t ntiles = (n + EDC_TILE - 1) / EDC_TILE;
  const int64_t po = part_off[seg];
  if (po < 0) return;
  const int wgs = part_wgs[seg], tiles = part_tiles[seg];
  const double* p = part + po;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ntiles - 1; j += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int w = 0; w < wgs; ++w) acc += p[(int64_t)w * tiles + j];
    scratch[(int64_t)seg * IRA_EDC_SCRATCH_DOUBLES + j] = acc;
  }
}

// numpy.maximum semantics: a NaN operand gives NaN (fmax would drop it).  A NaN sample makes the reference's whole
// curve NaN (decay.py:151-166), an infinite one NaN before it and the floor after it.
__device__ __forceinline__ double np_max(double a, double b) { return (a != a) ? a : fmax(a, b); }

// One wave per segment: exclusive prefix sums of the tile totals (in blocks of 64 with a shuffle scan; any fixed
// association will do, every later kernel reads THESE carries) and the normaliser edc[0] = tot[T-1] + carry[T-1],
// which is exactly how the emit pass forms the value at index 0.
__global__ __launch_bounds__(IRA_WAVE) void edc_carry_kernel(const int64_t* __restrict__ len, int nseg, double eps,
                                                             double* __restrict__ scratch) {
  const int seg = blockIdx.x;
  if (seg >= nseg) return;
  const int64_t n = len[seg];
  if (n <= 0) return;
  const int q = IRA_EDC_SCRATCH_DOUBLES / 2;
  double* sc = scratch + (int64_t)seg * IRA_EDC_SCRATCH_DOUBLES;
  const int ntiles = (int)((n + EDC_TILE - 1) / EDC_TILE);
  const int lane = threadIdx.x;
  double base = 0.0;
  for (int j0 = 0; j0 < ntiles; j0 += IRA_WAVE) {
    const int j = j0 + lane;
    const double tot = j < ntiles ? sc[j] : 0.0;
    double incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double dn = __shfl_up(incl, o, 64);
      if (lane >= o) incl += dn;
    }
    double excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 0.0;
    const double carry = base + excl;
    if (j < ntiles) {
      sc[q + j] = carry;
      if (j == ntiles - 1) sc[IRA_EDC_SCRATCH_DOUBLES - 1] = np_max(tot + carry, eps);
    }
    base = base + __shfl(incl, 63, 64);
  }
}

// dB value of one suffix sum: 10 log10(max(sum, eps) / norm) through the table log2 (ira_log.h): ~30 instructions per
// sample instead of ~110 for an f64 divide + log10, same value to ~1e-14 dB; sum == norm gives exactly 0 dB.
// Shared by the emit pass and the fused fit kernel so that both see the same float32 curve bit for bit.
__device__ __forceinline__ double edc_db64(double sum, double eps, double norm, double lnorm, bool fast,
                                           const ira::LogTabEntry* ltab) {
  const double v = np_max(sum, eps);
  if (fast && v > 1e-300 && v < 1e300) return 3.0102999566398120 * (ira::log2_table(v, ltab) - lnorm);
  return 10.0 * log10(v / norm);
}

__global__ __launch_bounds__(EDC_THREADS) void edc_emit_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int64_t* __restrict__ len, double eps,
    double floor_db, float* __restrict__ out, double* __restrict__ out64, const int64_t* __restrict__ out_off,
    const double* __restrict__ scratch) {
  __shared__ EdcShared sh;
  __shared__ ira::LogTabEntry ltab[ira::LOGTAB_N];
  const int seg = blockIdx.y;
  const int64_t j = blockIdx.x;
  const int64_t n = len[seg];
  const int64_t ntiles = (n + EDC_TILE - 1) / EDC_TILE;
  if (j >= ntiles) return;
  const int64_t hi = n - j * EDC_TILE;
  const int64_t lo = hi - EDC_TILE > 0 ? hi - EDC_TILE : 0;
  const int tl = (int)(hi - lo);
  float xv[EDC_PER_THREAD];
  tile_load(x + off[seg] + lo, tl, xv);
  ira::build_log_table(ltab, threadIdx.x);
  const double* sc = scratch + (int64_t)seg * IRA_EDC_SCRATCH_DOUBLES;
  const double carry = sc[IRA_EDC_SCRATCH_DOUBLES / 2 + j];
  const double norm = sc[IRA_EDC_SCRATCH_DOUBLES - 1];
  double s[EDC_PER_THREAD];
  tile_suffix_scan(xv, sh, 0, s);                      // its barrier also publishes the log table
  const bool fast = norm > 1e-300 && norm < 1e300;
  const double lnorm = fast ? ira::log2_table(norm, ltab) : 0.0;
  float* dst = out ? out + out_off[seg] + lo : nullptr;
  double* dst64 = out64 ? out64 + out_off[seg] + lo : nullptr;
  const int i0 = EDC_PER_THREAD * threadIdx.x;
  float o[EDC_PER_THREAD];
#pragma unroll
  for (int r = 0; r < EDC_PER_THREAD; ++r) {
    const double db = edc_db64(s[r] + carry, eps, norm, lnorm, fast, ltab);
    if (dst64 && i0 + r < tl) dst64[i0 + r] = db;      // unfloored f64 (optional dB smoothing, decay.py:161-164)
    o[r] = (float)np_max(db, floor_db);
  }
  if (dst) {
    if (i0 + EDC_PER_THREAD <= tl) {
#pragma unroll
      for (int jj = 0; jj < EDC_PER_THREAD / 4; ++jj) {
        const edc_f4 v = {o[4 * jj], o[4 * jj + 1], o[4 * jj + 2], o[4 * jj + 3]};
        *reinterpret_cast<edc_f4*>(dst + i0 + 4 * jj) = v;
      }
    } else {
#pragma unroll
      for (int r = 0; r < EDC_PER_THREAD; ++r)
        if (i0 + r < tl) dst[i0 + r] = o[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Optional dB smoothing of the EDC (decay.py:159-166, default off): numpy.convolve(edc_db, ones(w)/w, mode="same") on the
// UNFLOORED float64 curve, then the floor and the float32 cast.  "same" keeps the centre of the full convolution: output i
// sums the inputs i - (w-1-h) .. i + h with h = (w-1)/2 (integer division), zero beyond the ends (the edges sag).
// Each product a[j] * (1/w) is rounded like numpy's, the sum runs in ascending j.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edc_box_smooth_kernel(const double* __restrict__ in64,
                                                             const int64_t* __restrict__ off,
                                                             const int64_t* __restrict__ len, int window, double floor_db,
                                                             float* __restrict__ out) {
  const int seg = blockIdx.y;
  const long long n = len[seg];
  const double* a = in64 + off[seg];
  float* o = out + off[seg];
  const double inv_w = 1.0 / (double)window;
  const long long h = (window - 1) / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    long long j0 = i + h - (window - 1), j1 = i + h;
    if (j0 < 0) j0 = 0;
    if (j1 > n - 1) j1 = n - 1;
    double acc = 0.0;
    for (long long j = j0; j <= j1; ++j) acc += a[j] * inv_w;
    o[i] = (float)np_max(acc, floor_db);
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
  if (frac == frac) frac = fmin(fmax(frac, 0.0), 1.0);       // numpy.clip keeps NaN (a NaN neighbour: decay.py:192-196)
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

// ------------------------------------------------------------------------------------------------
// a3-a6 fused (ira_edc_fits): crossings and decay-line fits straight from the SAMPLES.
//
// After edc_sums / edc_carry every 4096-sample tile of a segment knows the energy behind it (its carry), i.e. a
// lower bound of every EDC value inside it -- and an EDC is monotone.  So the tile that holds a dB crossing is
// known WITHOUT the curve, and the regression only needs the curve between the two crossings of a range.  One
// 256-thread workgroup per segment (four to five of them share a CU and hide each other's memory round trips):
//   phase A  the carries are staged in LDS; each target level binary-searches the first tile (in time) that is not
//            certainly above it; tiles are then re-scanned (the emit pass's own scan code and dB conversion -> the
//            identical float32 values) until the first index with edc_db <= target is found; the tile's dB values sit
//            in LDS, so y[idx-1], y[idx] for the interpolation come from there;
//   phase B  the tiles between the crossings are re-scanned once and shifted first and second moments of every
//            range accumulated in float64 (one sweep instead of the three passes of curve_fit_kernel).
// Nothing reads an EDC array: a band EDC that only feeds its fits (rt60bands.py:272-321) is never written, and
// the decay block's curve is written by the emit pass for the caller but not read back here.
//
// Three launches.  Phase A is serial per segment (a handful of tiles); phase B is not: a low band's fit range spans
// the whole file (118 tiles of a 10 s IR), and as ONE workgroup per segment it was a 1 ms tail at 5 % VALU utilisation
// (profiles/r02: 192 workgroups on 256 CUs).  So:
//   edc_fit_kernel      (segments)            phase A; crossings out; range records + chunking into the segment's scratch
//   edc_moments_kernel  (chunks x segments)   phase B for one chunk of K consecutive tiles: partial moments -> scratch
//   edc_line_kernel     (segments x ranges)   partial moments summed in chunk order (deterministic), line fit, records out
// The records live in the first half of the segment's scratch (the tile totals, dead once edc_carry has run).
// ------------------------------------------------------------------------------------------------
constexpr int FITREC_HDR = 8;                                   // j_first, j_last, K (tiles per chunk), nchunks
constexpr int FITREC_RANGE = 8;                                 // ts, te, tmid, ymid, a0, a1, ok, -
constexpr int FITREC_PART = FITREC_HDR + FITREC_RANGE * 4;      // partial moments: [chunk][range][6]
constexpr int FIT_MAX_CHUNKS = 80;
constexpr int FIT_MIN_CHUNK_TILES = 4;                          // 16 k samples per chunk at least
static_assert(FITREC_PART + FIT_MAX_CHUNKS * 4 * 6 <= IRA_EDC_SCRATCH_DOUBLES / 2, "fit records fit the dead half of the scratch");

struct EdcFitRange {
  double ts, te, tmid, ymid;
  long long a0, a1;
  float ts32, te32;
  int ok, pad;
};

struct EdcFitShared {
  EdcShared scan;
  ira::LogTabEntry ltab[ira::LOGTAB_N];
  double carry[EDC_MAX_TILES + 1];
  float db[EDC_TILE];
  EdcFitRange rng[FIT_MAX_RANGES];
  double part[EDC_WAVES][FIT_MAX_RANGES][6];
  unsigned long long found[FIT_MAX_TARGETS];      // local index of the first crossing in the tile in LDS
  long long idx[FIT_MAX_TARGETS];                 // first index with edc_db <= target; n = never
  float y_at[FIT_MAX_TARGETS], y_prev[FIT_MAX_TARGETS];
  float tgt32[FIT_MAX_TARGETS];
  int first_tile[FIT_MAX_TARGETS];                // tile (counted from the END) where the search starts; -1 = never
};

struct EdcMomentsShared {
  EdcShared scan;
  ira::LogTabEntry ltab[ira::LOGTAB_N];
  float db[EDC_TILE];
  EdcFitRange rng[FIT_MAX_RANGES];
  double part[EDC_WAVES][FIT_MAX_RANGES][6];
};

// Samples of tile j (counted from the end of the segment) into registers: issued one tile AHEAD of its use (phase B walks
// consecutive tiles), so that the memory round trip of tile j-1 runs under the scan and the logarithms of tile j.
__device__ __forceinline__ void edc_tile_fetch(const float* __restrict__ src, long long n, int j, float xv[EDC_PER_THREAD]) {
  const long long hi = n - (long long)j * EDC_TILE;
  const long long lo = hi - EDC_TILE > 0 ? hi - EDC_TILE : 0;
  tile_load(src + lo, (int)(hi - lo), xv);
}

// dB curve of tile j from its pre-fetched samples into sh.db, in time order.  Same scan, same carry and same conversion
// as edc_emit_kernel.  The caller alternates `parity` between consecutive calls.
__device__ __forceinline__ void edc_tile_to_lds(const float xv[EDC_PER_THREAD], long long n, int j, double eps,
                                                double floor_db, double norm, double lnorm, bool fast, int parity,
                                                EdcFitShared& sh, long long& tstart, int& tlen) {
  const long long hi = n - (long long)j * EDC_TILE;
  tstart = hi - EDC_TILE > 0 ? hi - EDC_TILE : 0;
  tlen = (int)(hi - tstart);
  double s[EDC_PER_THREAD];
  tile_suffix_scan(xv, sh.scan, parity, s);
  const double carry = sh.carry[j];
  const int i0 = EDC_PER_THREAD * threadIdx.x;
#pragma unroll
  for (int r = 0; r < EDC_PER_THREAD; ++r)
    if (i0 + r < tlen) sh.db[i0 + r] = (float)np_max(edc_db64(s[r] + carry, eps, norm, lnorm, fast, sh.ltab), floor_db);
  __syncthreads();
}

__device__ __forceinline__ double crossing_time_from_values(long long idx, long long n, float y_prev, float y_at,
                                                            double target, const TimeAxis& ta) {
  if (idx >= n) return __longlong_as_double(0x7ff8000000000000ll);  // NaN = "no crossing"
  if (idx == 0) return (double)ta.at(0);
  const double t0 = (double)ta.at(idx - 1);
  const double t1 = (double)ta.at(idx);
  const double y0 = (double)y_prev;
  const double y1 = (double)y_at;
  if (y1 == y0) return t1;
  double frac = (target - y0) / (y1 - y0);
  if (frac == frac) frac = fmin(fmax(frac, 0.0), 1.0);       // numpy.clip keeps NaN (a NaN neighbour: decay.py:192-196)
  return t0 + frac * (t1 - t0);
}

__global__ __launch_bounds__(EDC_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void edc_fit_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int64_t* __restrict__ len, double eps,
    double floor_db, FitParams P, double* __restrict__ scratch, double* __restrict__ fit_out,
    double* __restrict__ cross_out) {
  extern __shared__ __align__(16) unsigned char ef_smem[];
  EdcFitShared& sh = *reinterpret_cast<EdcFitShared*>(ef_smem);
  const int seg = blockIdx.x;
  const long long n = len[seg];
  const int tid = threadIdx.x;
  const double qnan = __longlong_as_double(0x7ff8000000000000ll);
  const TimeAxis ta{nullptr, P.t_mul, P.t_div};
  double* fo = fit_out ? fit_out + (int64_t)seg * P.nranges * IRA_FIT_DOUBLES : nullptr;
  double* co = cross_out ? cross_out + (int64_t)seg * P.ncross : nullptr;
  const int ntargets = 2 * P.nranges + P.ncross;
  if (n <= 0) {
    if (tid == 0) {
      for (int r = 0; r < P.nranges; ++r)
        for (int k = 0; k < IRA_FIT_DOUBLES; ++k) fo[r * IRA_FIT_DOUBLES + k] = (k == 0) ? 0.0 : qnan;
      for (int j = 0; j < P.ncross && co; ++j) co[j] = qnan;
      scratch[(int64_t)seg * IRA_EDC_SCRATCH_DOUBLES + 3] = -1.0;      // nchunks < 0: records already written
    }
    return;
  }
  ira::build_log_table(sh.ltab, tid);
  const float* src = x + off[seg];
  double* sc = scratch + (int64_t)seg * IRA_EDC_SCRATCH_DOUBLES;
  const int ntiles = (int)((n + EDC_TILE - 1) / EDC_TILE);
  for (int j = tid; j < ntiles; j += EDC_THREADS) sh.carry[j] = sc[IRA_EDC_SCRATCH_DOUBLES / 2 + j];
  const double norm = sc[IRA_EDC_SCRATCH_DOUBLES - 1];
  const bool fast = norm > 1e-300 && norm < 1e300;
  __syncthreads();
  const double lnorm = fast ? ira::log2_table(norm, sh.ltab) : 0.0;

  // ---- where each target's search starts -------------------------------------------------------------------------
  if (tid < FIT_MAX_TARGETS) {
    double tv = 0.0;
    if (tid < 2 * P.nranges) tv = (tid & 1) ? P.lo[tid >> 1] : P.hi[tid >> 1];
    else if (tid < ntargets) tv = P.cross[tid - 2 * P.nranges];
    const float t32 = (float)tv;
    sh.tgt32[tid] = t32;
    sh.idx[tid] = n;
    sh.y_at[tid] = 0.0f; sh.y_prev[tid] = 0.0f;
    int first = -1;
    if (tid < ntargets && !(t32 < (float)floor_db)) {      // the floored curve never goes below float32(floor_db)
      // A value v has float32(dB(v)) > target for certain once dB(v) exceeds the target by two float32 ulps; in the
      // linear domain: v > thr_hi.  Every value of tile j is >= its carry (sums of non-negative terms are monotone
      // in floating point), so carry > thr_hi rules the whole tile out; eps clamps from below the same way.  Carries
      // grow with j (towards the start of the segment): the first tile IN TIME that is not ruled out is the LARGEST j
      // with !(carry[j] > thr_hi) -- binary search; a NaN normaliser rules nothing out (search from the start).
      const double ulp2 = fmax(fabs((double)t32) * 2.384185791015625e-07, 1e-9);
      const double thr_hi = norm * pow(10.0, ((double)t32 + ulp2) / 10.0) * (1.0 + 1e-12);
      if (!(eps > thr_hi)) {
        if (!(thr_hi == thr_hi)) {
          first = ntiles - 1;
        } else {
          int lo_j = 0, hi_j = ntiles - 1;                  // carry[0] = 0 is never above a positive threshold
          while (lo_j < hi_j) {
            const int mid = (lo_j + hi_j + 1) >> 1;
            if (!(sh.carry[mid] > thr_hi)) lo_j = mid; else hi_j = mid - 1;
          }
          first = lo_j;
        }
      }
    }
    sh.first_tile[tid] = first;
  }
  __syncthreads();

  // ---- phase A: first-crossing indices ---------------------------------------------------------------------------
  int parity = 0;
  int j_top = -1;
  for (int k = 0; k < ntargets; ++k) j_top = sh.first_tile[k] > j_top ? sh.first_tile[k] : j_top;
  for (int j = j_top; j >= 0; --j) {
    bool need = false, open = false;
    for (int k = 0; k < ntargets; ++k) {
      const bool unfound = sh.first_tile[k] >= 0 && sh.idx[k] >= n;
      open = open || unfound;
      need = need || (unfound && sh.first_tile[k] >= j);
    }
    if (!open) break;
    if (!need) continue;                                   // uniform: every thread reads the same shared state
    long long tstart; int tlen;
    {
      float xv[EDC_PER_THREAD];
      edc_tile_fetch(src, n, j, xv);
      edc_tile_to_lds(xv, n, j, eps, floor_db, norm, lnorm, fast, parity, sh, tstart, tlen);
    }
    parity ^= 1;
    if (tid < FIT_MAX_TARGETS) sh.found[tid] = 0xffffffffffffffffull;
    __syncthreads();
    for (int k = 0; k < ntargets; ++k) {
      if (!(sh.first_tile[k] >= j && sh.idx[k] >= n)) continue;      // uniform
      const float t32 = sh.tgt32[k];
      int f = tlen;
      for (int i = tid; i < tlen; i += EDC_THREADS)
        if (sh.db[i] <= t32) { f = i; break; }             // ascending per thread: its first hit is its smallest
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(f, o, 64);
        f = other < f ? other : f;
      }
      if ((tid & 63) == 0 && f < tlen) atomicMin(&sh.found[k], (unsigned long long)f);
    }
    __syncthreads();
    if (tid < ntargets && sh.first_tile[tid] >= j && sh.idx[tid] >= n && sh.found[tid] != 0xffffffffffffffffull) {
      const int f = (int)sh.found[tid];
      sh.idx[tid] = tstart + f;
      sh.y_at[tid] = sh.db[f];
      if (f > 0) sh.y_prev[tid] = sh.db[f - 1];
      else if (tstart > 0) {
        // the sample before this tile is the LAST sample of tile j+1: its suffix sum is its own energy + carry[j+1]
        const double v = (double)src[tstart - 1];
        sh.y_prev[tid] = (float)np_max(edc_db64(v * v + sh.carry[j + 1], eps, norm, lnorm, fast, sh.ltab), floor_db);
      }
    }
    __syncthreads();
  }

  if (co && tid == 0) {
    for (int jc = 0; jc < P.ncross; ++jc) {
      const int k = 2 * P.nranges + jc;
      co[jc] = crossing_time_from_values(sh.idx[k], n, sh.y_prev[k], sh.y_at[k], P.cross[jc], ta);
    }
  }
  if (P.nranges == 0) return;

  // ---- the ranges: crossing times, index window, mid-range shifts; and the chunking of phase B --------------------------
  if (tid < FIT_MAX_RANGES) {
    EdcFitRange g;
    g.ok = 0; g.ts = g.te = qnan; g.tmid = g.ymid = 0.0; g.ts32 = g.te32 = 0.0f; g.a0 = 0; g.a1 = -1; g.pad = 0;
    if (tid < P.nranges) {
      const int r = tid;
      const long long i_hi = sh.idx[2 * r], i_lo = sh.idx[2 * r + 1];
      g.ts = crossing_time_from_values(i_hi, n, sh.y_prev[2 * r], sh.y_at[2 * r], P.hi[r], ta);
      g.te = crossing_time_from_values(i_lo, n, sh.y_prev[2 * r + 1], sh.y_at[2 * r + 1], P.lo[r], ta);
      g.ok = !(isnan(g.ts) || isnan(g.te) || g.te <= g.ts) ? 1 : 0;
      if (g.ok) {
        g.a0 = i_hi - 2 > 0 ? i_hi - 2 : 0;
        g.a1 = i_lo + 2 < n - 1 ? i_lo + 2 : n - 1;
        g.tmid = 0.5 * (g.ts + g.te);
        g.ymid = 0.5 * (P.hi[r] + P.lo[r]);
      }
    }
    sh.rng[tid] = g;
    double* rec = sc + FITREC_HDR + FITREC_RANGE * tid;
    rec[0] = g.ts; rec[1] = g.te; rec[2] = g.tmid; rec[3] = g.ymid;
    rec[4] = (double)g.a0; rec[5] = (double)g.a1; rec[6] = (double)g.ok; rec[7] = 0.0;
  }
  __syncthreads();
  if (tid == 0) {
    long long lo_all = n, hi_all = -1;
    for (int r = 0; r < P.nranges; ++r) {
      if (!sh.rng[r].ok) continue;
      lo_all = sh.rng[r].a0 < lo_all ? sh.rng[r].a0 : lo_all;
      hi_all = sh.rng[r].a1 > hi_all ? sh.rng[r].a1 : hi_all;
    }
    int j_first = 0, j_last = 0, K = FIT_MIN_CHUNK_TILES, nchunks = 0;
    if (hi_all >= lo_all) {
      j_first = (int)((n - 1 - lo_all) / EDC_TILE);              // earliest tile in time (largest number)
      j_last = (int)((n - 1 - hi_all) / EDC_TILE);
      const int T = j_first - j_last + 1;
      const int kk = (T + FIT_MAX_CHUNKS - 1) / FIT_MAX_CHUNKS;
      K = kk > FIT_MIN_CHUNK_TILES ? kk : FIT_MIN_CHUNK_TILES;
      nchunks = (T + K - 1) / K;
    }
    sc[0] = (double)j_first; sc[1] = (double)j_last; sc[2] = (double)K; sc[3] = (double)nchunks;
  }
}

// Phase B for chunk blockIdx.x of segment blockIdx.y: tiles j_first - c K ... down to j_last, at most K of them.
__global__ __launch_bounds__(EDC_THREADS) void edc_moments_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int64_t* __restrict__ len, double eps,
    double floor_db, FitParams P, double* __restrict__ scratch) {
  __shared__ EdcMomentsShared sh;
  const int seg = blockIdx.y, c = blockIdx.x;
  double* sc = scratch + (int64_t)seg * IRA_EDC_SCRATCH_DOUBLES;
  // the segment's header and range records: all loads first, then the moves to scalar registers (one round trip)
  const double h0 = sc[0], h1 = sc[1], h2 = sc[2], h3 = sc[3];
  const long long n = len[seg];
  const long long so = off[seg];
  const double nrm = sc[IRA_EDC_SCRATCH_DOUBLES - 1];
  const int nchunks = (int)ira::uniform(h3);
  if (c >= nchunks) return;
  const int j_first = (int)ira::uniform(h0), j_last = (int)ira::uniform(h1), K = (int)ira::uniform(h2);
  const int tid = threadIdx.x;
  const TimeAxis ta{nullptr, P.t_mul, P.t_div};
  ira::build_log_table(sh.ltab, tid);
  if (tid < FIT_MAX_RANGES) {
    const double* rec = sc + FITREC_HDR + FITREC_RANGE * tid;
    EdcFitRange g;
    g.ts = rec[0]; g.te = rec[1]; g.tmid = rec[2]; g.ymid = rec[3];
    g.a0 = (long long)rec[4]; g.a1 = (long long)rec[5]; g.ok = tid < P.nranges ? (int)rec[6] : 0; g.pad = 0;
    g.ts32 = (float)g.ts; g.te32 = (float)g.te;
    sh.rng[tid] = g;
  }
  for (int i = tid; i < EDC_WAVES * FIT_MAX_RANGES * 6; i += EDC_THREADS) (&sh.part[0][0][0])[i] = 0.0;
  const float* src = x + ira::uniform(so);
  const long long nn = ira::uniform(n);
  const double norm = ira::uniform(nrm);
  const bool fast = norm > 1e-300 && norm < 1e300;
  __syncthreads();
  const double lnorm = fast ? ira::log2_table(norm, sh.ltab) : 0.0;
  const int j_hi = j_first - c * K;
  const int j_lo = j_hi - K + 1 > j_last ? j_hi - K + 1 : j_last;
  const int lane = tid & 63, wave = tid >> 6;
  int parity = 0;
  float xv[EDC_PER_THREAD];
  edc_tile_fetch(src, nn, j_hi, xv);
  double carry = sc[IRA_EDC_SCRATCH_DOUBLES / 2 + j_hi];
  for (int j = j_hi; j >= j_lo; --j) {
    float xn[EDC_PER_THREAD];
    double carry_next = 0.0;
    if (j > j_lo) {                                            // next tile's samples and carry: in flight during this tile's work
      edc_tile_fetch(src, nn, j - 1, xn);
      carry_next = sc[IRA_EDC_SCRATCH_DOUBLES / 2 + j - 1];
    }
    const long long hi = nn - (long long)j * EDC_TILE;
    const long long tstart = hi - EDC_TILE > 0 ? hi - EDC_TILE : 0;
    const int tlen = (int)(hi - tstart);
    {
      double sfx[EDC_PER_THREAD];
      tile_suffix_scan(xv, sh.scan, parity, sfx);
      const int i0 = EDC_PER_THREAD * tid;
#pragma unroll
      for (int r = 0; r < EDC_PER_THREAD; ++r)
        if (i0 + r < tlen) sh.db[i0 + r] = (float)np_max(edc_db64(sfx[r] + carry, eps, norm, lnorm, fast, sh.ltab), floor_db);
    }
    __syncthreads();
    parity ^= 1;
    for (int r = 0; r < P.nranges; ++r) {
      const EdcFitRange g = sh.rng[r];                       // uniform
      if (!g.ok || g.a1 < tstart || g.a0 >= tstart + tlen) continue;
      double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0, m4 = 0.0, m5 = 0.0;
      for (int i = tid; i < tlen; i += EDC_THREADS) {
        const long long gi = tstart + i;
        const float tf = ta.at(gi);
        if (gi >= g.a0 && gi <= g.a1 && tf >= g.ts32 && tf <= g.te32) {
          const double u = (double)tf - g.tmid, w = (double)sh.db[i] - g.ymid;
          m0 += 1.0; m1 += u; m2 += w; m3 += u * u; m4 += u * w; m5 += w * w;
        }
      }
      m0 = ira::wave_sum(m0); m1 = ira::wave_sum(m1); m2 = ira::wave_sum(m2);
      m3 = ira::wave_sum(m3); m4 = ira::wave_sum(m4); m5 = ira::wave_sum(m5);
      if (lane == 0) {
        double* pp = sh.part[wave][r];
        pp[0] += m0; pp[1] += m1; pp[2] += m2; pp[3] += m3; pp[4] += m4; pp[5] += m5;
      }
    }
    __syncthreads();                                         // sh.db is rewritten by the next tile
    if (j > j_lo) {
#pragma unroll
      for (int r = 0; r < EDC_PER_THREAD; ++r) xv[r] = xn[r];
      carry = carry_next;
    }
  }
  if (tid < FIT_MAX_RANGES * 6) {
    const int r = tid / 6, k = tid - 6 * r;
    double v = 0.0;
    for (int w = 0; w < EDC_WAVES; ++w) v += sh.part[w][r][k];
    sc[FITREC_PART + (c * FIT_MAX_RANGES + r) * 6 + k] = v;
  }
}

// One thread per (segment, range): the chunks' partial moments in chunk order, then the least-squares line.
__global__ __launch_bounds__(256) void edc_line_kernel(int nseg, FitParams P, const double* __restrict__ scratch,
                                                       double* __restrict__ fit_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nseg * P.nranges) return;
  const int seg = t / P.nranges, r = t - seg * P.nranges;
  const double qnan = __longlong_as_double(0x7ff8000000000000ll);
  const double* sc = scratch + (int64_t)seg * IRA_EDC_SCRATCH_DOUBLES;
  const int nchunks = (int)sc[3];
  if (nchunks < 0) return;                                   // empty segment: written by edc_fit_kernel
  const double* rec = sc + FITREC_HDR + FITREC_RANGE * r;
  const double ts = rec[0], te = rec[1], tmid = rec[2], ymid = rec[3];
  const bool ok = rec[6] != 0.0;
  double* o = fit_out + ((int64_t)seg * P.nranges + r) * IRA_FIT_DOUBLES;
  if (!ok) {
    o[0] = 0.0; o[1] = ts; o[2] = te; for (int k = 3; k < IRA_FIT_DOUBLES; ++k) o[k] = qnan;
    return;
  }
  double cnt = 0.0, su = 0.0, sw = 0.0, suu = 0.0, suw = 0.0, sww = 0.0;
  for (int c = 0; c < nchunks; ++c) {
    const double* pp = sc + FITREC_PART + (c * FIT_MAX_RANGES + r) * 6;
    cnt += pp[0]; su += pp[1]; sw += pp[2]; suu += pp[3]; suw += pp[4]; sww += pp[5];
  }
  const long long npts = (long long)cnt;
  if (npts < P.min_points) {
    o[0] = 0.0; o[1] = ts; o[2] = te; for (int k = 3; k < 7; ++k) o[k] = qnan; o[7] = (double)npts;
    return;
  }
  // centred moments from the shifted sums (shift = mid-range: the subtractions lose nothing that matters in f64)
  const double um = su / cnt, wm = sw / cnt;
  const double stt = suu - su * um, sty = suw - su * wm, syy = sww - sw * wm;
  const double slope = sty / stt;
  const double tm = tmid + um, ym = ymid + wm;
  const double icpt = ym - slope * tm;
  const double sres = syy - slope * sty;                     // residual sum of squares of the least-squares line
  const bool neg = slope < 0.0;                              // also false for NaN (stt == 0)
  o[0] = neg ? 1.0 : 0.0;
  o[1] = ts; o[2] = te; o[3] = slope; o[4] = icpt;
  o[5] = syy > 0.0 ? 1.0 - fmax(sres, 0.0) / syy : 0.0;
  o[6] = -60.0 / slope;
  o[7] = (double)npts;
}

}  // namespace

extern "C" int32_t ira_peak_index(const float* x_dev, const int64_t* off_dev, const int64_t* len_dev, int32_t nseg,
                                  int64_t max_len, int64_t* peak_dev, float* peak_abs_dev, void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(len_dev); IRA_CHECK_PTR(peak_dev);
  if (nseg <= 0) return nseg == 0 ? IRA_OK : IRA_E_SIZE;
  // indices travel in the low 32 bits of the atomicMax key; grid.y carries the segment
  if (nseg > 65535 || max_len < 0 || max_len > 0xFFFFFFFFll) return IRA_E_SIZE;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(peak_dev, 0, sizeof(int64_t) * (size_t)nseg, st);
  if (e != hipSuccess) return ira_hip_status(e);
  // the grid covers the LONGEST segment (the host knows the lengths); shorter segments' spare chunks exit at once
  const int64_t chunks = max_len > 0 ? (max_len + PEAK_CHUNK - 1) / PEAK_CHUNK : 1;
  peak_partial_kernel<<<dim3((unsigned)chunks, nseg), PEAK_THREADS, 0, st>>>(
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
  if (max_len <= 0 || max_len > (int64_t)EDC_MAX_TILES * EDC_TILE) return IRA_E_SIZE;
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = (int)((max_len + EDC_TILE - 1) / EDC_TILE);
  if (nseg > 65535) return IRA_E_SIZE;
  edc_sums_kernel<<<dim3(ntiles, nseg), EDC_THREADS, 0, st>>>(x_dev, off_dev, len_dev, scratch_dev, nullptr);
  edc_carry_kernel<<<nseg, IRA_WAVE, 0, st>>>(len_dev, nseg, eps, scratch_dev);
  edc_emit_kernel<<<dim3(ntiles, nseg), EDC_THREADS, 0, st>>>(x_dev, off_dev, len_dev, eps, floor_db, edc_db_dev,
                                                              edc_db64_dev, edc_off_dev, scratch_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_edc_fits(const float* x_dev, const int64_t* off_dev, const int64_t* len_dev, int32_t nseg,
                                int64_t max_len, double eps, double floor_db, float t_mul, float t_div,
                                const double* ranges_hi_lo, int32_t nranges, int32_t min_points,
                                const double* cross_targets, int32_t ncross, double* fit_out_dev,
                                double* cross_out_dev, float* edc_db_dev, const int64_t* edc_off_dev,
                                double* scratch_dev, const double* tile_part_dev, const int64_t* part_off_dev,
                                const int32_t* part_wgs_dev, const int32_t* part_tiles_dev, void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(len_dev); IRA_CHECK_PTR(scratch_dev);
  if (nranges < 0 || nranges > FIT_MAX_RANGES || ncross < 0 || ncross > FIT_MAX_CROSS) return IRA_E_SIZE;
  if (nranges > 0) { IRA_CHECK_PTR(ranges_hi_lo); IRA_CHECK_PTR(fit_out_dev); }
  if (ncross > 0) { IRA_CHECK_PTR(cross_targets); IRA_CHECK_PTR(cross_out_dev); }
  if (edc_db_dev != nullptr) IRA_CHECK_PTR(edc_off_dev);
  if (nseg <= 0) return nseg == 0 ? IRA_OK : IRA_E_SIZE;
  if (max_len <= 0 || max_len > (int64_t)EDC_MAX_TILES * EDC_TILE) return IRA_E_SIZE;
  if (nseg > 65535) return IRA_E_SIZE;
  FitParams P{};
  for (int r = 0; r < nranges; ++r) { P.hi[r] = ranges_hi_lo[2 * r]; P.lo[r] = ranges_hi_lo[2 * r + 1]; }
  for (int j = 0; j < ncross; ++j) P.cross[j] = cross_targets[j];
  P.nranges = nranges; P.ncross = ncross; P.min_points = min_points; P.rel_to_peak = 0;
  P.floor_db = floor_db; P.min_peak_above_floor = 0.0; P.t_mul = t_mul; P.t_div = t_div; P.t_axis = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = (int)((max_len + EDC_TILE - 1) / EDC_TILE);
  if (sizeof(EdcFitShared) > 64 * 1024) {   // > 64 KB of dynamic LDS needs the opt-in (idempotent, host-side only)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(edc_fit_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(EdcFitShared));
    if (e != hipSuccess) return ira_hip_status(e);
  }
  if (tile_part_dev != nullptr && (part_off_dev == nullptr || part_wgs_dev == nullptr || part_tiles_dev == nullptr))
    return IRA_E_NULL;
  edc_sums_kernel<<<dim3(ntiles, nseg), EDC_THREADS, 0, st>>>(x_dev, off_dev, len_dev, scratch_dev,
                                                              tile_part_dev ? part_off_dev : nullptr);
  if (tile_part_dev != nullptr && ntiles > 1)
    edc_tiles_from_parts_kernel<<<dim3((ntiles - 1 + 127) / 128, nseg), 128, 0, st>>>(len_dev, tile_part_dev, part_off_dev,
                                                                                      part_wgs_dev, part_tiles_dev, scratch_dev);
  edc_carry_kernel<<<nseg, IRA_WAVE, 0, st>>>(len_dev, nseg, eps, scratch_dev);
  if (nranges + ncross > 0)
    edc_fit_kernel<<<nseg, EDC_THREADS, sizeof(EdcFitShared), st>>>(x_dev, off_dev, len_dev, eps, floor_db, P,
                                                                     scratch_dev, fit_out_dev,
                                                                     ncross > 0 ? cross_out_dev : nullptr);
  if (nranges > 0) {
    int max_chunks = (ntiles + FIT_MIN_CHUNK_TILES - 1) / FIT_MIN_CHUNK_TILES;
    if (max_chunks > FIT_MAX_CHUNKS) max_chunks = FIT_MAX_CHUNKS;
    edc_moments_kernel<<<dim3(max_chunks, nseg), EDC_THREADS, 0, st>>>(x_dev, off_dev, len_dev, eps, floor_db, P,
                                                                       scratch_dev);
    edc_line_kernel<<<(nseg * nranges + 255) / 256, 256, 0, st>>>(nseg, P, scratch_dev, fit_out_dev);
  }
  if (edc_db_dev != nullptr)
    edc_emit_kernel<<<dim3(ntiles, nseg), EDC_THREADS, 0, st>>>(x_dev, off_dev, len_dev, eps, floor_db, edc_db_dev,
                                                                nullptr, edc_off_dev, scratch_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_edc_box_smooth(const double* edc_db64_dev, const int64_t* off_dev, const int64_t* len_dev,
                                      int32_t nseg, int64_t max_len, int32_t window, double floor_db, float* out_dev,
                                      void* stream) {
  IRA_CHECK_PTR(edc_db64_dev); IRA_CHECK_PTR(off_dev); IRA_CHECK_PTR(len_dev); IRA_CHECK_PTR(out_dev);
  if (nseg <= 0) return nseg == 0 ? IRA_OK : IRA_E_SIZE;
  if (window < 1 || max_len <= 0 || nseg > 65535) return IRA_E_SIZE;
  if ((int64_t)window > max_len) return IRA_E_UNSUPPORTED;      // numpy's "same" then returns `window` values, not len
  long long blocks = (max_len + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  edc_box_smooth_kernel<<<dim3((unsigned)blocks, nseg), 256, 0, (hipStream_t)stream>>>(edc_db64_dev, off_dev, len_dev, window,
                                                                                       floor_db, out_dev);
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
