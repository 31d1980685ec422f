// Direct float64 DFTs of SMOOTH lengths n = 2^a 3^b 5^c (e.g. 480 000 = 10 s at 48 kHz): a four-step n = N1 x N2
// transform with mixed-radix (5, 4, 3, 2) Stockham sub-FFTs in LDS -- two passes over n complex values instead of
// Bluestein's three passes over M = 2^ceil(log2(2n)) (ira_fftlong.hip), i.e. ~2.3x less memory traffic and ~4x fewer
// flops for the RT60 filter bank (reference analyse/rt60bands.py:170-175) and every other whole-file transform whose
// length happens to be smooth.  Same inputs, outputs and pairing conventions as the Bluestein entry points.
//
//   x[n1*N2 + n2]  --P1 (columns: N1-point FFT over n1, twiddle W_n^(k1 n2))-->  work, stored in TILES of C2 adjacent k1:
//                     work[(k1 / C2) * (N2 * C2) + n2 * C2 + k1 % C2]  (scattered 32-byte WRITES, which do not stall)
//   work           --P2 (N2-point FFT over n2 for the C2 values of k1 of one tile, read as ONE contiguous block)-->
//                     X[k1 + N1*k2]                                                                 (runs of C2 values)
// Inverse transforms run the same forward machinery on the conjugated input (conj(DFT(conj .)) / n).
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <type_traits>

#include "ira_bandmask.h"
#include "ira_fft_reg.h"

namespace {

using ira::cplx;
typedef cplx<double> cd;

constexpr int SM_THREADS = 256;
constexpr int SM_MAX_N = 1024;     // largest sub-transform (LDS: 2 buffers x C x N x 16 B)
constexpr int SM_MAX_RADICES = 12;

// Division of a small non-negative integer by a plan constant d: floor(x / d) = umulhi(x, magic), magic = floor(2^32 / d) + 1,
// exact for x < 2^32 / d (here x < 2^20, d <= 1024).  The tile index arithmetic of these kernels (i / C, i % C, k1 / C2,
// p / N2, the mixed-radix digits of dif_slot) was ~40 % of their VALU instructions as generic 32-bit divisions (~25
// instructions each), and the kernels are VALU-bound (~70 % busy).
struct FastDiv { unsigned d, magic; };
__host__ __device__ inline FastDiv fast_div_of(unsigned d) {
  return {d, d <= 1u ? 0u : (unsigned)(0x100000000ull / d) + 1u};
}
__device__ __forceinline__ unsigned fdiv(unsigned x, FastDiv f) { return f.d <= 1u ? x : __umulhi(x, f.magic); }

// The radix passes of one in-LDS transform.  Every field is read with COMPILE-TIME indices (pick(), dif_slot()): the plan is
// a kernel argument, and an entry indexed by the pass counter is a scalar load from the kernel-argument segment followed by
// s_waitcnt lgkmcnt(0) every time the source mentions it -- round 2's kernels issued ~41 of them per wave (SQ_INSTS_SMEM,
// profiles/r02_smooth_fft_counters.txt), three to four per ELEMENT in the output stages (dif_slot), each a serial ~200-cycle
// stall on the critical path of a tile.  With constant indices the loads are loop-invariant and leave the loops.
constexpr int SM_MAXP = 6;          // passes per transform: 729 = 3^6 is the longest chain for lengths <= 1024
struct PassTab {
  int nr;
  int r[SM_MAXP];                   // radix of pass i
  int len[SM_MAXP];                 // sub-transform length entering pass i: N / (r_0 ... r_{i-1})
  int scale[SM_MAXP];               // N / len[i]
  int span[SM_MAXP];                // N / (r_0 ... r_i): weight of digit i in the slot of X[k] after the passes
  unsigned per[SM_MAXP], per_magic[SM_MAXP];     // butterflies per transform in pass i (N / r_i) as a divisor
  unsigned m_magic[SM_MAXP];        // floor(2^32 / (len[i] / r_i)) + 1
  unsigned r_magic[SM_MAXP];        // floor(2^32 / r_i) + 1
};

struct SmoothPlan {
  int n, n1, n2;
  FastDiv dc1, dc2, dn1, dn2;       // the same numbers as divisors
  PassTab p1, p2;                   // radix passes of the N1- and N2-point transforms
  int c1, c2;                       // columns per workgroup in pass 1 / pass 2
  unsigned c2_one;                  // 1 if c2 == 1 (branch-free i / c2 in pass 2's tile load, see there)
  int ld1, ld2;                     // LDS column strides (= n1 / n2, see column_stride)
  const cd* t1;                     // exp(-2 pi i k / N1), k < N1
  const cd* t2;                     // exp(-2 pi i k / N2), k < N2
  const cd* tf;                     // exp(-2 pi i k / n),  k < N2
  double hstep_c, hstep_s;          // half-size inverse: cos / sin of pi (256 / c1) / n1, the step of the W_2n^(-i) twiddle
                                    // between a thread's consecutive pass-1 elements (valid when 256 % c1 == 0)
  int pairs;                        // forward transform of an interleaved real signal with the untangling FUSED into pass 2:
                                    // the intermediate is stored in MIRROR-PAIR tiles -- tile 0 = rows {0, n1/2}, tile p =
                                    // rows {p, n1 - p} (n1 even, c2 = 2) -- so that Z[k] and Z[n - k] meet in one workgroup
  double pstep_c, pstep_s;          // pairs: cos / sin of -pi (256 / 2) / n2, the step of W_2n^k between a thread's
                                    // consecutive pass-2 elements (k advances by 128 n1)
  int sparse_q;                     // narrow-band path: jobs whose band hull spans <= sparse_q * n2 bins skip pass 1 (0: off)
  int stamp;                        // diagnostics (IRA_SMOOTH_STAMP): per-phase cycle counts of one workgroup per kernel
  int ablate;                       // diagnostics (IRA_SMOOTH_ABLATE, timing only -- results are wrong): 1 pass-2 band output
                                    // written tile-major (contiguous per workgroup), 2 pass-1 output contiguous, 4 pass-1
                                    // spectrum gather contiguous, 8 / 16 no transform in pass 1 / 2, 32 no input arithmetic
                                    // in pass 1, 64 no output-twiddle loads in pass 1, 128 no digit-reversed slot lookups
                                    // (tools/experiments/smooth_ablate.sh, profiles/r03_smooth_ablation.txt); narrow-band kernel: 256 no
                                    // terms, 512 no W_n^(k1 n2) loads, 1024 no transform, 2048 no output stores
};

// ---- radix butterflies, forward sign (W = exp(-2 pi i / r)), natural order in and out, registers only -------------------
__device__ __forceinline__ cd mul_mi(cd z) { return {z.im, -z.re}; }      // -i z

template <int R>
__device__ __forceinline__ void bfly(cd (&a)[R]);

template <>
__device__ __forceinline__ void bfly<2>(cd (&a)[2]) {
  const cd t = a[1];
  a[1] = ira::csub(a[0], t);
  a[0] = ira::cadd(a[0], t);
}
template <>
__device__ __forceinline__ void bfly<3>(cd (&a)[3]) {
  const cd t1 = ira::cadd(a[1], a[2]);
  const cd t2 = {a[0].re - 0.5 * t1.re, a[0].im - 0.5 * t1.im};
  const cd d = ira::csub(a[1], a[2]);
  const cd r = mul_mi(cd{0.86602540378443864676 * d.re, 0.86602540378443864676 * d.im});
  a[0] = ira::cadd(a[0], t1);
  a[1] = ira::cadd(t2, r);
  a[2] = ira::csub(t2, r);
}
template <>
__device__ __forceinline__ void bfly<4>(cd (&a)[4]) {
  const cd p02 = ira::cadd(a[0], a[2]), m02 = ira::csub(a[0], a[2]);
  const cd p13 = ira::cadd(a[1], a[3]), m13 = mul_mi(ira::csub(a[1], a[3]));
  a[0] = ira::cadd(p02, p13);
  a[2] = ira::csub(p02, p13);
  a[1] = ira::cadd(m02, m13);
  a[3] = ira::csub(m02, m13);
}
template <>
__device__ __forceinline__ void bfly<5>(cd (&a)[5]) {
  constexpr double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
  constexpr double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
  const cd t1 = ira::cadd(a[1], a[4]), t2 = ira::cadd(a[2], a[3]);
  const cd t3 = ira::csub(a[1], a[4]), t4 = ira::csub(a[2], a[3]);
  const cd m1 = {a[0].re + c1 * t1.re + c2 * t2.re, a[0].im + c1 * t1.im + c2 * t2.im};
  const cd m2 = {a[0].re + c2 * t1.re + c1 * t2.re, a[0].im + c2 * t1.im + c1 * t2.im};
  const cd n1 = mul_mi(cd{s1 * t3.re + s2 * t4.re, s1 * t3.im + s2 * t4.im});
  const cd n2 = mul_mi(cd{s2 * t3.re - s1 * t4.re, s2 * t3.im - s1 * t4.im});
  a[0] = {a[0].re + t1.re + t2.re, a[0].im + t1.im + t2.im};
  a[1] = ira::cadd(m1, n1);
  a[4] = ira::csub(m1, n1);
  a[2] = ira::cadd(m2, n2);
  a[3] = ira::csub(m2, n2);
}
// 6 = 2 x 3 (Cooley-Tukey inside the registers): n = 2a + b, k = k1 + 3 k2
template <>
__device__ __forceinline__ void bfly<6>(cd (&a)[6]) {
  cd e[3] = {a[0], a[2], a[4]}, o[3] = {a[1], a[3], a[5]};
  bfly<3>(e);
  bfly<3>(o);
  o[1] = ira::cmul(o[1], cd{0.5, -0.86602540378443864676});      // W6^1
  o[2] = ira::cmul(o[2], cd{-0.5, -0.86602540378443864676});     // W6^2
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    a[k1] = ira::cadd(e[k1], o[k1]);
    a[k1 + 3] = ira::csub(e[k1], o[k1]);
  }
}
// 10 = 2 x 5: n = 2a + b, k = k1 + 5 k2
template <>
__device__ __forceinline__ void bfly<10>(cd (&a)[10]) {
  cd e[5] = {a[0], a[2], a[4], a[6], a[8]}, o[5] = {a[1], a[3], a[5], a[7], a[9]};
  bfly<5>(e);
  bfly<5>(o);
  o[1] = ira::cmul(o[1], cd{0.80901699437494742410, -0.58778525229247312917});      // W10^1
  o[2] = ira::cmul(o[2], cd{0.30901699437494742410, -0.95105651629515357212});      // W10^2
  o[3] = ira::cmul(o[3], cd{-0.30901699437494742410, -0.95105651629515357212});     // W10^3
  o[4] = ira::cmul(o[4], cd{-0.80901699437494742410, -0.58778525229247312917});     // W10^4
#pragma unroll
  for (int k1 = 0; k1 < 5; ++k1) {
    a[k1] = ira::cadd(e[k1], o[k1]);
    a[k1 + 5] = ira::csub(e[k1], o[k1]);
  }
}
// powers of two through the shared decimation-in-frequency kernel (it leaves X[k] in slot bitrev(k))
template <>
__device__ __forceinline__ void bfly<8>(cd (&a)[8]) {
  ira::dft_dif<double, 8>(a);
  cd t;
  t = a[1]; a[1] = a[4]; a[4] = t;
  t = a[3]; a[3] = a[6]; a[6] = t;
}
// Sub-FFT twiddles W_N^t, t < N <= 1024, from two 32+33-entry LDS tables: coarse[t >> 5] = W_N^(32 (t >> 5)),
// fine[t & 31] = W_N^(t & 31): two LDS reads and one complex multiply (a dependent GLOBAL load per butterfly was the
// critical path of every pass, an in-kernel sincospi costs ~100 instructions).
constexpr int SM_TW = 66;       // 33 coarse + 33 fine entries
// In two halves: the table LOAD is issued before the tile's loads, the LDS write comes after them (one round trip for both).
__device__ __forceinline__ cd twiddle_lds_fetch(const cd* __restrict__ tw, int N, int tid) {
  const int t = tid < 33 ? 32 * tid : tid - 33;
  return (tid < 66 && t < N) ? tw[t] : cd{1.0, 0.0};
}
__device__ __forceinline__ void twiddle_lds_put(cd* tab, cd v, int tid) {
  if (tid < 66) tab[tid] = v;
}

// ---- the transform IN PLACE (one LDS buffer) ------------------------------------------------------------------------------
// (Round 1's ping-pong Stockham variant, kept as an A/B until round 3, is gone: it needed twice the LDS per tile.)
// Decimation in frequency with every butterfly writing back to the R slots it read: no second buffer, no staging, one
// barrier per pass.  The price is the output order: X[k], k = k_0 + r_0 k_1 + r_0 r_1 k_2 ..., ends up at the
// digit-reversed slot k_0 (N / r_0) + k_1 (N / (r_0 r_1)) + ...; the consumers ask dif_slot() where a k lives.
// Half the LDS per tile means twice the columns per workgroup at the same number of resident workgroups: the passes are a
// closed queue of tiles cycling between a memory phase and an LDS phase, and what is in flight per CU is what LDS holds.
// (The butterflies of ALL nbat transforms are dealt to the threads in one sweep: a transform has only N / R = 64 ... 100 of
// them, and one transform after the other left three quarters of the 256 lanes idle in each of nbat dependent rounds.)
template <int R>
__device__ __forceinline__ void dif_pass(cd* x, int ld, int len, int scale, unsigned per, unsigned per_magic,
                                         unsigned m_magic, const cd* __restrict__ tw, int tid, int nbat) {
  const int m = len / R;                        // (R is a compile-time constant)
  {
    for (int g = tid; g < (int)per * nbat; g += SM_THREADS) {
      const int t = per == 1u ? g : (int)__umulhi((unsigned)g, per_magic);                // g / per
      const int bf = g - t * (int)per;
      cd* xt = x + t * ld;
      const int blk = m == 1 ? bf : (int)__umulhi((unsigned)bf, m_magic);
      const int p = bf - blk * m;
      cd* base = xt + blk * len + p;
      cd v[R];
#pragma unroll
      for (int j = 0; j < R; ++j) v[j] = base[m * j];
      bfly<R>(v);
      base[0] = v[0];
      if (p == 0) {
#pragma unroll
        for (int k = 1; k < R; ++k) base[m * k] = v[k];
      } else {
        const int tt = scale * p;                                          // W_len^(p k) = W_N^(scale p k), scale p < N / R
        const cd w1 = ira::cmul(tw[tt >> 5], tw[33 + (tt & 31)]);
        cd w = w1;
#pragma unroll
        for (int k = 1; k < R; ++k) {
          base[m * k] = ira::cmul(v[k], w);
          if (k + 1 < R) w = ira::cmul(w, w1);
        }
      }
    }
  }
  __syncthreads();
}

// entry `i` of a per-pass array by a chain of scalar selects over compile-time indices (see PassTab)
template <typename T>
__device__ __forceinline__ T pick(int i, const T (&a)[SM_MAXP]) {
  T v = a[0];
#pragma unroll
  for (int j = 1; j < SM_MAXP; ++j) v = (i == j) ? a[j] : v;
  return v;
}

__device__ __forceinline__ void lds_fft_dif_inplace(cd* a, int ld, const PassTab& T, const cd* __restrict__ tw, int tid,
                                                    int nbat) {
  for (int pass = 0; pass < T.nr; ++pass) {
    const int r = pick(pass, T.r), len = pick(pass, T.len), scale = pick(pass, T.scale);
    const unsigned per = pick(pass, T.per), pm = pick(pass, T.per_magic), mm = pick(pass, T.m_magic);
    switch (r) {
      case 10: dif_pass<10>(a, ld, len, scale, per, pm, mm, tw, tid, nbat); break;
      case 8: dif_pass<8>(a, ld, len, scale, per, pm, mm, tw, tid, nbat); break;
      case 6: dif_pass<6>(a, ld, len, scale, per, pm, mm, tw, tid, nbat); break;
      case 5: dif_pass<5>(a, ld, len, scale, per, pm, mm, tw, tid, nbat); break;
      case 4: dif_pass<4>(a, ld, len, scale, per, pm, mm, tw, tid, nbat); break;
      case 3: dif_pass<3>(a, ld, len, scale, per, pm, mm, tw, tid, nbat); break;
      default: dif_pass<2>(a, ld, len, scale, per, pm, mm, tw, tid, nbat); break;
    }
  }
}

// slot of X[k] after lds_fft_dif_inplace: digit i of k (mixed radix r_0, r_1, ...) times span_i = N / (r_0 ... r_i)
__device__ __forceinline__ int dif_slot(int k, const PassTab& T) {
  int slot = 0;
#pragma unroll
  for (int i = 0; i < SM_MAXP; ++i) {
    if (i < T.nr) {
      const int q = (int)__umulhi((unsigned)k, T.r_magic[i]);             // k / r_i   (r_i >= 2)
      slot += (k - q * T.r[i]) * T.span[i];
      k = q;
    }
  }
  return slot;
}

using BandMaskS = ira::BandMask;

struct SJobs {
  // forward: one or two real signals per job
  const float* x;
  const int64_t* xoff;
  const int64_t* x2off;          // null or -1: single
  int use_hann;
  // interleave = 1 (forward): every job is ONE real signal of even length 2 n whose even / odd samples are the real /
  // imaginary parts of an n-point complex sequence (x2off = xoff + 1, both read with stride 2, Hann index 2 i / 2 i + 1 on
  // the REAL signal's window); smooth_half_split_kernel then untangles the n-point transform into the 2n-point half
  // spectrum.  half_out = 1 (inverse): the mirror image -- one band of a 2 n-point real signal per n-point transform.
  // Both keep a channel's transforms to itself: no signal of another channel shares its complex transform, so a channel's
  // results do not depend on which channels happen to sit beside it in a batch (byte-identical
  // records for any sharding).
  int interleave, half_out;
  const int32_t* data_len; const int32_t* win_len; const int32_t* data_len2; const int32_t* win_len2;   // optional
  cd* spec_out;
  const int64_t* spec_off; const int64_t* spec_off2;
  cd* zpair; const int64_t* zpair_off;
  // inverse: masked spectra -> band signals
  const cd* spec;
  const int64_t* sp_off; const int64_t* sp_off2;
  const BandMaskS* bands;
  const double* freq_val;
  float* y;
  const int64_t* y1_off; const int64_t* y2_off;
  // inverse, optional (null = off): per-job record of the NARROW-band path (SparseInfo below), written by
  // band_compact_kernel and read by the three kernels that follow it
  int32_t* info;
  // inverse, optional (null = off; round 5): energies of the Schroeder-EDC tiles of every band signal, one partial per
  // (signal, tile, workgroup of the second pass) -- see band_tile_partials
  double* tile_part;
};

// ---- narrow bands: the first pass is a handful of terms per point, not a transform ---------------------------------------
// A third-octave band around 100 Hz occupies ~500 of the 240 001 bins of a 10 s spectrum.  In the four-step inverse the
// first pass transforms, for every column n2, the N1 bins n2, n2 + N2, n2 + 2 N2, ... -- of which a band of W bins (and its
// Hermitian mirror) touches ceil(W / N2) each: for W <= q N2 the column "transform"
//     Y[k1][n2] = W_n^(k1 n2) * sum over the <= 2q non-zero n1 of  x[n1 N2 + n2] W_N1^(n1 k1)
// is cheaper evaluated as written than as an N1-point FFT, and it can be evaluated where it is consumed: in the input stage
// of the second pass.  Such a job never touches its n-point work array (7.7 MB written and read back per 10 s pair) and
// never runs pass 1 at all; the masked, Hermitian-extended values x[i] of the 2W bins are computed once per job
// (band_compact_kernel, into the head of the job's otherwise unused work array) instead of once per column.
// Which jobs are narrow is decided on the device from the same float32 mask cuts the regular path uses (no host copy of
// that logic): SparseInfo = {narrow, lo, w, q} with [lo, lo + w) the hull of the job's band supports on the positive
// side and q = ceil(w / N2) the terms per cluster.  Results are the regular path's up to the rounding of a different
// (shorter) summation order.
constexpr int SM_SPARSE_INFO = 4;       // int32 per job
constexpr int SM_SPARSE_R = 16;         // row-twiddle table entries per row: q + 2 <= 16
constexpr int SM_SPARSE_QMAX = SM_SPARSE_R - 2;

enum { SM_SIGNAL = 0, SM_SPECTRUM = 1 };
enum { SM_OUT_SPEC = 0, SM_OUT_BANDS = 1, SM_OUT_SPEC_PAIRS = 2 };

__device__ __forceinline__ double hann_s(long long i, long long L) {
  if (L <= 1) return 1.0;
  return 0.5 + 0.5 * cospi((double)(2 * i + 1 - L) / (double)(L - 1));
}

// The job's own values (offsets, lengths, band records), read ONCE per workgroup into scalar registers (ira::uniform): the
// tile loops below then contain no loads but the tile's own.  (As plain J.xoff[e] / J.bands[2 e] expressions inside the
// loops every one of them was a vector load followed by s_waitcnt vmcnt(0), i.e. one memory round trip per ELEMENT: the
// pass-1 input phase of the band inverses took 47 k cycles of a 69 k-cycle tile)
struct SCtx {
  long long o1, o2;                 // SM_SIGNAL: sample offsets (o2 < 0: one signal); SM_SPECTRUM: spectrum offsets
  // What the SM_SPECTRUM tile loops index with is 32-bit (n <= 2^20) on wave-uniform 64-bit BASES: ~6 % fewer instructions in
  // the input stage of the inverse pass 1 (VALU 90 % busy in the third-octave bank, profiles/r04_block_counters.txt); the A/B
  // on one box is within its noise (1.33 vs 1.34 ms per 256 two-band jobs).
  int st;                           // SM_SIGNAL: sample stride (2 = even / odd samples of one real signal)
  int nd1, nd2, lw1, lw2;           // SM_SIGNAL: samples actually read / Hann window lengths of the two signals
  const cd* sp1; const cd* sp2;     // SM_SPECTRUM: the two spectra
  bool two;                         // SM_SPECTRUM: the two bands come from two different spectra
  BandMaskS b1, b2;
  ira::MaskCuts k1, k2;             // first bins past each mask edge (ira_bandmask.h)
  int s1_lo, s1_hi, s2_lo, s2_hi;   // bins where band 1 / band 2 can be non-zero
  double fv;
};

__device__ __forceinline__ BandMaskS uniform_band(const BandMaskS& p) {
  BandMaskS b{};
  b.kind = ira::uniform(p.kind);
  b.hp_x0 = ira::uniform(p.hp_x0); b.hp_x1 = ira::uniform(p.hp_x1);
  b.lp_x0 = ira::uniform(p.lp_x0); b.lp_x1 = ira::uniform(p.lp_x1);
  return b;
}

// (all loads first, then the moves to scalar registers: ONE round trip)
template <int MODE>
__device__ __forceinline__ SCtx smooth_ctx(const SmoothPlan& P, const SJobs& J, int e) {
  SCtx c{};
  const int n = P.n;
  if (MODE == SM_SIGNAL) {
    const long long o1 = J.xoff[e];
    const long long o2 = J.x2off ? (long long)J.x2off[e] : -1ll;
    const int nd1 = J.data_len ? J.data_len[e] : (int)n, nd2 = J.data_len2 ? J.data_len2[e] : -1;
    const int lw1 = J.win_len ? J.win_len[e] : (int)n, lw2 = J.win_len2 ? J.win_len2[e] : -1;
    c.o1 = ira::uniform(o1);
    c.o2 = ira::uniform(o2);
    c.st = 1;
    c.nd1 = ira::uniform(nd1);
    c.nd2 = J.data_len2 ? ira::uniform(nd2) : c.nd1;
    c.lw1 = ira::uniform(lw1);
    c.lw2 = J.win_len2 ? ira::uniform(lw2) : c.lw1;
    if (J.interleave) {
      // one real signal of 2 n samples (data_len / win_len, when given, count REAL samples): element i of the transform
      // is x[2 i] + i x[2 i + 1]
      const int real_nd = J.data_len ? c.nd1 : 2 * n, real_lw = J.win_len ? c.lw1 : 2 * n;
      c.st = 2;
      c.o2 = c.o1 + 1;
      c.nd1 = (real_nd + 1) / 2;          // even samples available
      c.nd2 = real_nd / 2;                // odd samples available
      c.lw1 = c.lw2 = real_lw;
    }
    c.nd1 = c.nd1 > 0 ? c.nd1 : 0;
    c.nd2 = (c.o2 >= 0 && c.nd2 > 0) ? c.nd2 : 0;          // no second signal: nothing of it is ever read
  } else {
    const long long o1 = J.sp_off[e];
    const long long o2 = J.sp_off2 ? (long long)J.sp_off2[e] : -1ll;
    const BandMaskS b1 = J.bands[2 * e], b2 = J.bands[2 * e + 1];
    const double fv = J.freq_val[e];
    c.o1 = ira::uniform(o1);
    c.o2 = J.sp_off2 ? ira::uniform(o2) : c.o1;
    c.two = c.o2 != c.o1;
    c.b1 = uniform_band(b1);
    c.b2 = uniform_band(b2);
    c.fv = ira::uniform(fv);
    ira::band_cuts(c.b1, c.b2, c.fv, J.half_out ? (int)n : (int)(n / 2), c.k1, c.k2);   // highest bin of the half spectrum
    ira::band_support(c.b1, c.k1, c.s1_lo, c.s1_hi);
    ira::band_support(c.b2, c.k2, c.s2_lo, c.s2_hi);
    c.sp1 = J.spec + c.o1;
    c.sp2 = J.spec + c.o2;
  }
  return c;
}

// Input generation in two phases so that a thread's memory reads are all in flight before the first one is used:
// smooth_fetch does nothing but the loads, smooth_value the arithmetic (window, masks, Hermitian extension).
// (The loads are UNCONDITIONAL -- a lane beyond the data reads sample 0 of the batch buffer and drops it in smooth_value --
// and nothing is computed from them here: a load inside a per-lane branch, or one converted on the spot, is waited for at
// once, which made the forward pass-1 input phase ten serial round trips.)
struct RawIn { double a, b, c, d; float fa, fb; };
// (half-size inverse) bins ka and kb of ONE spectrum; a bin outside the band's support reads bin 0 (multiplied by 0 later)
__device__ __forceinline__ void r_load2(const SCtx& c, int ka, int kb, RawIn& r) {
  const cd xa = c.sp1[(ka >= c.s1_lo && ka < c.s1_hi) ? ka : 0];
  const cd xb = c.sp1[(kb >= c.s1_lo && kb < c.s1_hi) ? kb : 0];
  r.a = xa.re; r.b = xa.im; r.c = xb.re; r.d = xb.im;
}

template <int MODE, bool HALF>
__device__ __forceinline__ RawIn smooth_fetch(const SmoothPlan& P, const SJobs& J, const SCtx& c, int i) {
  RawIn r{0.0, 0.0, 0.0, 0.0, 0.0f, 0.0f};
  const int n = P.n;
  if (MODE == SM_SIGNAL) {
    // (64-bit element indices in SM_SIGNAL mode, index_of() included: with 32-bit ones hipcc's allocation of the forward
    // kernels tips over their 80-register bound and spills 28 bytes per lane)
    const long long li = i;
    r.fa = *(li < c.nd1 ? J.x + c.o1 + c.st * li : J.x);
    r.fb = *((c.o2 >= 0 && li < c.nd2) ? J.x + c.o2 + c.st * li : J.x);       // no branch around it either (same reason)
  } else if (HALF) {
    // half-size inverse of ONE band: element i needs the masked bins i and n - i of the (n + 1)-bin half spectrum
    r_load2(c, i, n - i, r);
  } else {
    const int k = i > n / 2 ? n - i : i;
    // A bin whose mask is zero for certain reads bin 0 instead (its value is multiplied by 0 either way): the low and mid
    // bands are empty above 2.2 kHz, i.e. 98 % / 91 % of their pass-1 reads hit one cached line instead of HBM.
    // (Unconditional loads on purpose: see RawIn.)
    if (!c.two) {
      const bool need = (k >= c.s1_lo && k < c.s1_hi) || (k >= c.s2_lo && k < c.s2_hi);
      const cd x1 = c.sp1[need ? k : 0];
      r.a = x1.re; r.b = x1.im;
    } else {
      const cd x1 = c.sp1[(k >= c.s1_lo && k < c.s1_hi) ? k : 0];
      r.a = x1.re; r.b = x1.im;
      const cd x2 = c.sp2[(k >= c.s2_lo && k < c.s2_hi) ? k : 0];
      r.c = x2.re; r.d = x2.im;
    }
  }
  return r;
}

// (HALF: cs + i sn = W_2n^(-i) = exp(+i pi i / n), from the caller -- one sincospi per thread and a rotation per element
// instead of a sincospi per element, which was most of the half-size pass-1 input stage: 256 half-size jobs took as long
// as 256 full-size ones)
template <int MODE, bool HALF>
__device__ __forceinline__ cd smooth_value(const SmoothPlan& P, const SJobs& J, const SCtx& c, int i, const RawIn& r,
                                           double cs, double sn) {
  const int n = P.n;
  if (MODE == SM_SIGNAL) {
    const long long li = i;
    double v = li < c.nd1 ? (double)r.fa : 0.0, v2 = (c.o2 >= 0 && li < c.nd2) ? (double)r.fb : 0.0;
    if (J.use_hann) {
      v *= hann_s(c.st * li, c.lw1);
      if (c.o2 >= 0) v2 *= hann_s(c.st * li + (c.st - 1), c.lw2);
    }
    return {v, v2};
  } else if (HALF) {
    // One band y of 2 n real samples from an n-point transform: with Xm = X * mask (n + 1 bins),
    //   E[i] = (Xm[i] + conj Xm[n-i]) / 2,  O[i] = W_2n^(-i) (Xm[i] - conj Xm[n-i]) / 2,  Z = E + i O,
    //   y[2 m] + i y[2 m + 1] = IDFT_n(Z)[m]        (inverse = conj(DFT(conj .)) / n: feed conj(Z))
    const double ma = (double)ira::mask_cut(c.b1, c.k1, i, c.fv);
    const double mb = (double)ira::mask_cut(c.b1, c.k1, n - i, c.fv);
    const cd xa = {r.a * ma, r.b * ma}, xb = {r.c * mb, -r.d * mb};            // Xm[i], conj Xm[n - i]
    const cd e = {0.5 * (xa.re + xb.re), 0.5 * (xa.im + xb.im)};
    const cd d = {0.5 * (xa.re - xb.re), 0.5 * (xa.im - xb.im)};
    const cd o = {cs * d.re - sn * d.im, cs * d.im + sn * d.re};                // W_2n^(-i) d
    const cd z = {e.re - o.im, e.im + o.re};                                     // E + i O
    return {z.re, -z.im};
  } else {
    // conj of the Hermitian extension of X1 m1 + i X2 m2  (inverse = conj(DFT(conj .)) / n)
    const bool upper = i > n / 2;
    const int k = upper ? n - i : i;
    cd x1 = {r.a, upper ? -r.b : r.b};
    const double m1 = (double)ira::mask_cut(c.b1, c.k1, k, c.fv);
    const double m2 = (double)ira::mask_cut(c.b2, c.k2, k, c.fv);
    cd w;
    if (!c.two) {
      w = ira::cmul(x1, cd{m1, m2});
    } else {
      const cd x2 = {r.c, upper ? -r.d : r.d};
      w = {x1.re * m1 - x2.im * m2, x1.im * m1 + x2.re * m2};
    }
    return {w.re, -w.im};
  }
}

// SM_SPECTRUM: can element i of the transform be non-zero at all (is one of the bins it reads inside a band's support)?
template <bool HALF>
__device__ __forceinline__ bool smooth_nonzero(const SmoothPlan& P, const SCtx& c, int i) {
  const int n = P.n;
  if (HALF) {
    const int k2 = n - i;
    return (i >= c.s1_lo && i < c.s1_hi) || (k2 >= c.s1_lo && k2 < c.s1_hi);
  }
  const int k = i > n / 2 ? n - i : i;
  return (k >= c.s1_lo && k < c.s1_hi) || (k >= c.s2_lo && k < c.s2_hi);
}

// ---- narrow-band path, step 1: classify the job and compact its non-zero input.  grid (ceil(sparse_q n2 / 256), jobs) -----
// work[e n + t] = x[lo + t], work[e n + wp + t] = x[n - (lo + t)] (0 where lo + t = 0: index n does not exist), t < w, with
// x = the conjugated, masked, Hermitian-extended sequence the regular pass 1 builds on the fly (smooth_value); both
// clusters are followed by n2 zeros (wp = w + n2), so that the consumer can step through them without range tests.
template <bool HALF>
__global__ __launch_bounds__(256) void band_compact_kernel(SmoothPlan P, SJobs J, cd* __restrict__ work) {
  const int e = blockIdx.y, tid = threadIdx.x;
  const SCtx ctx = smooth_ctx<SM_SPECTRUM>(P, J, e);
  const int n = P.n;
  const int nbins = HALF ? n + 1 : n / 2 + 1;                      // bins of the half spectrum the masks are defined on
  int lo = 0x7fffffff, hi = 0;
  if (ctx.s1_hi > ctx.s1_lo) { lo = min(lo, ctx.s1_lo); hi = max(hi, min(ctx.s1_hi, nbins)); }
  if (!HALF && ctx.s2_hi > ctx.s2_lo) { lo = min(lo, ctx.s2_lo); hi = max(hi, min(ctx.s2_hi, nbins)); }
  if (hi <= lo) { lo = 0; hi = 0; }                                 // nothing passes: an all-zero signal, zero terms
  const int w = hi - lo;
  // the two clusters [lo, hi) and (n - hi, n - lo] must not meet, and must fit the head of the work array
  const int wp = w + P.n2;
  const bool narrow = w <= P.sparse_q * P.n2 && 2 * hi <= n && 2 * wp <= n;
  if (blockIdx.x == 0 && tid == 0) {
    int32_t* r = J.info + (size_t)SM_SPARSE_INFO * e;
    r[0] = narrow ? 1 : 0; r[1] = lo; r[2] = w; r[3] = (w + P.n2 - 1) / P.n2;
  }
  const int t = (int)blockIdx.x * 256 + tid;
  if (!narrow || t >= wp) return;
  cd* z = work + (long long)e * n;
  if (t >= w) {
    z[t] = cd{0.0, 0.0};
    z[wp + t] = cd{0.0, 0.0};
    return;
  }
  const int i = lo + t, i2 = n - i;
  const bool has2 = i > 0;
  const RawIn r1 = smooth_fetch<SM_SPECTRUM, HALF>(P, J, ctx, i);
  const RawIn r2 = smooth_fetch<SM_SPECTRUM, HALF>(P, J, ctx, has2 ? i2 : i);
  double c1 = 1.0, s1 = 0.0, c2 = 1.0, s2 = 0.0;
  if (HALF) {
    sincospi((double)i / (double)n, &s1, &c1);
    sincospi((double)(has2 ? i2 : i) / (double)n, &s2, &c2);
  }
  z[t] = smooth_value<SM_SPECTRUM, HALF>(P, J, ctx, i, r1, c1, s1);
  z[wp + t] = has2 ? smooth_value<SM_SPECTRUM, HALF>(P, J, ctx, i2, r2, c2, s2) : cd{0.0, 0.0};
}

constexpr int SM_U = 8;     // independent loads in flight per thread (pass 2)
constexpr int SM_UC = 5;    // (pass 1: a fetch is up to four doubles; 640 x 2 columns = 5 x 256: ONE round of loads)

// XCD-aware remap (speed only): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Neighbouring
// column tiles read/write different 32-byte pieces of the SAME 128-byte lines; give each XCD a contiguous range of
// (job, tile) pairs so that those pieces meet in one L2.
__device__ __forceinline__ void smooth_remap(unsigned& bx, unsigned& by) {
  const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
  const unsigned orig = blockIdx.y * gx + blockIdx.x;
  const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
  const unsigned wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
  bx = wg % gx; by = wg / gx;
}

#define SM_STAMP(var) do { if (IRA_ABL(P.stamp)) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)

// ---- pass 1: columns.  grid (N2 / C, jobs) ------------------------------------------------------------------------------
template <int MODE, bool HALF = false>
__global__ __launch_bounds__(SM_THREADS, ((MODE == SM_SIGNAL || !HALF) ? 6 : 5)) void smooth_cols_kernel(SmoothPlan P, SJobs J, cd* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cd* a = reinterpret_cast<cd*>(smem);
  const int C = P.c1, N1 = P.n1, N2 = P.n2;
  const int LD = P.ld1;                                    // column stride in LDS
  cd* twl = a + (size_t)C * LD;                            // SM_TW entries
  unsigned bx, by;
  smooth_remap(bx, by);
  const int e = (int)by, tid = threadIdx.x;
  const int n2_0 = (int)bx * C;
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  SM_STAMP(s0);
  if (MODE == SM_SPECTRUM && J.info != nullptr && ira::uniform(J.info[SM_SPARSE_INFO * e]) != 0) return;   // narrow job: no pass 1
  const SCtx ctx = smooth_ctx<MODE>(P, J, e);
  const cd twv = twiddle_lds_fetch(P.t1, N1, tid);        // issued first: shares the round trip of the tile loads below
  const int total1 = N1 * C;
  for (int base = 0; base < total1; base += SM_THREADS * SM_UC) {
    RawIn raw[SM_UC];
    using idx_t = typename std::conditional<MODE == SM_SIGNAL, long long, int>::type;
    auto index_of = [&](int u) -> idx_t {                   // recomputed, not kept: registers are what limits occupancy
      int i = base + tid + SM_THREADS * u;
      i = i < total1 ? i : total1 - 1;                      // clamp: every fetch is unconditional (stays in registers)
      if (IRA_ABL(P.ablate & 4)) return (int)(((long long)bx * total1 + i) % P.n);
      const int row = (int)fdiv((unsigned)i, P.dc1);
      return (idx_t)row * N2 + n2_0 + (i - row * C);
    };
#pragma unroll
    for (int u = 0; u < SM_UC; ++u) raw[u] = smooth_fetch<MODE, HALF>(P, J, ctx, index_of(u));
    // half-size inverse: the twiddle exp(i pi idx / n) of the thread's first element of this batch (issued under the loads);
    // with 256 % C == 0 a thread's elements are (256 / C) rows apart and the next ones follow by rotation
    double cs = 1.0, sn = 0.0;
    const bool rotate = HALF && (SM_THREADS % C) == 0;
    if (HALF && rotate) sincospi((double)index_of(0) / (double)P.n, &sn, &cs);
#pragma unroll
    for (int u = 0; u < SM_UC; ++u) {
      const int i = base + tid + SM_THREADS * u;
      if (HALF && !rotate) sincospi((double)index_of(u) / (double)P.n, &sn, &cs);
      // A third-octave band occupies a few per cent of the spectrum: when no lane of the wave holds a bin inside a band's
      // support the masks, the Hermitian extension and the products (~80 of the ~280 instructions per element of this
      // kernel) are skipped and the element is zero -- what the multiplication by the zero masks gives.
      cd v = {0.0, 0.0};
      if (IRA_ABL(P.ablate & 32)) v = {raw[u].a, raw[u].b}; else
      if (MODE != SM_SPECTRUM || __builtin_amdgcn_ballot_w64(smooth_nonzero<HALF>(P, ctx, index_of(u))) != 0ull)
        v = smooth_value<MODE, HALF>(P, J, ctx, index_of(u), raw[u], cs, sn);
      const int row = (int)fdiv((unsigned)i, P.dc1);
      if (i < total1) a[(i - row * C) * LD + row] = v;
      if (HALF && rotate) {
        const double nc = cs * P.hstep_c - sn * P.hstep_s;
        sn = sn * P.hstep_c + cs * P.hstep_s;
        cs = nc;
      }
    }
  }
  twiddle_lds_put(twl, twv, tid);
  __syncthreads();
  SM_STAMP(s1);
  const cd* r = a;
  if (!(IRA_ABL(P.ablate & 8))) lds_fft_dif_inplace(a, LD, P.p1, twl, tid, C);
  SM_STAMP(s2);
  cd* w = work + (long long)e * P.n;
  const int C2 = P.c2;
  // SM_SIGNAL with HALF = the MIRROR-PAIR layout of the intermediate (SmoothPlan::pairs): position j of a column stands for
  // row k1 = j/2 (even j; tile j/2, member 0) or its mirror n1 - j/2 (odd j; member 1) -- j = 0 / 1 are the two rows that
  // mirror themselves, 0 and n1/2 -- so that neighbouring lanes write the two members of one tile: 32 contiguous bytes.
  constexpr bool PAIRS = MODE == SM_SIGNAL && HALF;
  auto row_of = [&](int j) -> int {
    if (!PAIRS) return j;
    const int q = j >> 1;
    return (j & 1) ? (q == 0 ? (N1 >> 1) : N1 - q) : q;
  };
  for (int base = 0; base < total1; base += SM_THREADS * SM_UC) {
    cd th[SM_UC], tl[SM_UC];                               // the twiddle factors W_n^(k1 n2) of the batch: loads first
#pragma unroll
    for (int u = 0; u < SM_UC; ++u) {
      int i = base + tid + SM_THREADS * u;
      i = i < total1 ? i : total1 - 1;
      const int c = (int)fdiv((unsigned)i, P.dn1), k1 = row_of(i - c * N1);
      const unsigned p = (unsigned)k1 * (unsigned)(n2_0 + c);
      const unsigned hi = fdiv(p, P.dn2);
      if (IRA_ABL(P.ablate & 64)) { th[u] = {1.0, 0.0}; tl[u] = {0.0, 1.0}; continue; }
      th[u] = P.t1[hi];
      tl[u] = P.tf[p - hi * (unsigned)N2];
    }
#pragma unroll
    for (int u = 0; u < SM_UC; ++u) {
      const int i = base + tid + SM_THREADS * u;
      if (i >= total1) continue;
      const int c = (int)fdiv((unsigned)i, P.dn1), j = i - c * N1, k1 = row_of(j);
      const int n2 = n2_0 + c;
      const int kt = PAIRS ? (j >> 1) : (int)fdiv((unsigned)k1, P.dc2);                 // tile of k1 in pass 2
      const int km = PAIRS ? (j & 1) : k1 - kt * C2;                                     // and its place in the tile
      w[(IRA_ABL(P.ablate & 2)) ? (long long)bx * total1 + i : (long long)kt * ((long long)N2 * C2) + (long long)n2 * C2 + km] =
          ira::cmul(r[c * LD + ((IRA_ABL(P.ablate & 128)) ? k1 : dif_slot(k1, P.p1))], ira::cmul(th[u], tl[u]));
    }
  }
  if (IRA_ABL(P.stamp)) {
    SM_STAMP(s3);
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2)
      printf("SMOOTH cols<%d> N1 %d C %d: input %llu  fft %llu  twiddle+store %llu cycles\n", MODE, N1, C, s1 - s0, s2 - s1, s3 - s2);
  }
}

// ---- pass 2: N2-point transforms for C adjacent k1.  grid (N1 / C, jobs) --------------------------------------------------
// ---- tile energies of the band signals, from the second pass's own output stage (round 5) ---------------------------------
// The Schroeder EDC of a band signal (ira_edc.hip) starts from the energies of its 4096-sample tiles, counted from the END
// of the signal; edc_sums_kernel got them by reading every band signal again right after this pass had written it (config
// 3: 12.8 GB per step).  The values are all here: after the in-LDS transform the tile `r` holds the workgroup's N2 x C
// outputs k = k1 + N1 k2 (a comb through the whole signal), and what was stored is float32(re / n), float32(-im / n).
// Thread j sums the squares of the workgroup's samples that fall into EDC tile j, in a fixed order (columns, then k2
// ascending), in float64 (the square of a float32 value is exact) -> part[((job * 2 + signal) * wgs + bx) * ntile + j];
// edc_tiles_from_parts_kernel adds the wgs partials of a tile in a fixed order.  No atomics: the totals do not depend on
// scheduling.  (A half_out job writes ONE signal of 2 n samples, 2 k and 2 k + 1 from re and im: both go to signal 0.)
constexpr int SM_EDC_TILE = 4096;                           // = EDC_TILE of ira_edc.hip (checked by ira_edc_fits's caller contract)
__device__ __forceinline__ void band_tile_partials(const SmoothPlan& P, const SJobs& J, const cd* r, int LD, int k1_0, int e,
                                                   unsigned bx, bool second, int tid) {
  const long long n = P.n;
  const int C = P.c2, N1 = P.n1, N2 = P.n2;
  const bool half = J.half_out != 0;
  const long long len = half ? 2 * n : n;
  const int ntile = (int)((len + SM_EDC_TILE - 1) / SM_EDC_TILE), wgs = N1 / C;
  const double sc = 1.0 / (double)n;
  // layout [job][signal][workgroup][tile]: a workgroup's partials of one signal are contiguous (first version: [tile][workgroup],
  // 2 x 118 scattered 8-byte stores per workgroup -- the pass lost more than ira_edc_fits gained)
  double* base = J.tile_part + ((long long)e * 2 * wgs + bx) * ntile;
  // two neighbouring lanes share a tile: lane parity = the first column each of them takes (the columns of a tile are dealt to
  // the two, then added in lane order -- a fixed order); all 256 lanes work when the signal has >= 128 tiles
  const int CS = C >= 2 ? 2 : 1;
  for (int idx = tid; idx < ((ntile * CS + 1) & ~1); idx += SM_THREADS) {
    const int j = idx / CS, cs = idx - j * CS;
    double s1 = 0.0, s2 = 0.0;
    if (j < ntile) {
      const long long hi = len - (long long)j * SM_EDC_TILE, lo = hi > SM_EDC_TILE ? hi - SM_EDC_TILE : 0;   // samples [lo, hi)
      const long long klo = half ? lo / 2 : lo, khi = half ? hi / 2 : hi;      // (len and the tile length are even: so are lo, hi)
      for (int c = cs; c < C; c += CS) {
        const long long k1 = k1_0 + c, a = klo - k1, b = khi - k1;
        const int k2a = a <= 0 ? 0 : (int)fdiv((unsigned)(a + N1 - 1), P.dn1);
        int k2b = b <= 0 ? 0 : (int)fdiv((unsigned)(b + N1 - 1), P.dn1);
        k2b = k2b < N2 ? k2b : N2;
        for (int k2 = k2a; k2 < k2b; ++k2) {
          const cd v = r[c * LD + dif_slot(k2, P.p2)];
          const double y1 = (double)(float)(v.re * sc), y2 = (double)(float)(-v.im * sc);
          s1 = fma(y1, y1, s1);
          s2 = fma(y2, y2, s2);
        }
      }
    }
    if (CS == 2) {                                             // even lane: its columns + the odd neighbour's
      s1 += __shfl_down(s1, 1, 64);
      s2 += __shfl_down(s2, 1, 64);
    }
    if (cs == 0 && j < ntile) {
      if (half) {
        base[j] = s1 + s2;
      } else {
        base[j] = s1;
        if (second) base[(long long)wgs * ntile + j] = s2;
      }
    }
  }
}

template <int OUT>
__global__ __launch_bounds__(SM_THREADS) void smooth_rows_kernel(SmoothPlan P, SJobs J, const cd* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cd* a = reinterpret_cast<cd*>(smem);
  const int C = P.c2, N1 = P.n1, N2 = P.n2;
  const int LD = P.ld2;                                    // column stride in LDS
  cd* twl = a + (size_t)C * LD;
  unsigned bx, by;
  smooth_remap(bx, by);
  const int e = (int)by, tid = threadIdx.x;
  const int k1_0 = (int)bx * C;
  const cd* w = work + (long long)e * P.n;
  if (OUT == SM_OUT_BANDS && J.info != nullptr && ira::uniform(J.info[SM_SPARSE_INFO * e]) != 0) return;   // narrow job: see below
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  SM_STAMP(s0);
  // the job's output offsets, once (scalar registers; see SCtx)
  bool paired = false;
  long long out1 = 0, out2 = -1;
  if (OUT == SM_OUT_SPEC_PAIRS) {
    const long long so = J.spec_off[e];
    out1 = ira::uniform(so);
  } else if (OUT == SM_OUT_SPEC) {
    const long long x2 = J.x2off ? (long long)J.x2off[e] : -1ll, zo = J.x2off ? (long long)J.zpair_off[e] : 0ll;
    const long long so = J.spec_off[e];
    paired = ira::uniform(x2) >= 0;
    out1 = paired ? ira::uniform(zo) : ira::uniform(so);
  } else {
    const long long y1 = J.y1_off[e], y2 = J.y2_off[e];
    out1 = ira::uniform(y1);
    out2 = ira::uniform(y2);
  }
  const cd twv = twiddle_lds_fetch(P.t2, N2, tid);        // issued first: shares the round trip of the tile load below
  const int total2 = N2 * C;
  for (int base = 0; base < total2; base += SM_THREADS * SM_U) {
    // Clamped index for the load AND for the LDS store (a lane past the end rewrites the last element with its own value):
    // behind an `if (i < total2)` the compiler sinks each load into its store's branch -- eight serial round trips.
    cd raw[SM_U];
#pragma unroll
    for (int u = 0; u < SM_U; ++u) {
      int i = base + tid + SM_THREADS * u;
      i = i < total2 ? i : total2 - 1;
      raw[u] = w[(long long)bx * total2 + i];               // this tile: n2 * C + c, contiguous
    }
    // (nothing may be scheduled across this line: left alone, hipcc 7.2 sinks every load to its LDS store to save
    // registers -- load, s_waitcnt vmcnt(0), ds_write, eight times in a row: eight serial memory round trips per tile,
    // tools/isa_serial_loads.py)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < SM_U; ++u) {
      int i = base + tid + SM_THREADS * u;
      i = i < total2 ? i : total2 - 1;
      // i / C BRANCH-FREE (C = 1: magic 0, c2_one 1).  fdiv()'s `d <= 1 ? x : umulhi(x, magic)` is a uniform BRANCH here, and
      // hipcc sinks each tile load into the block of its LDS store across it: load, s_waitcnt vmcnt(0), ds_write, eight times
      // in a row -- eight serial memory round trips per tile (ISA dump, round 3; the sched_barrier above cannot help, the
      // sinking happens before scheduling).  Same family as the `if (i < total)` guards of round 2, DESIGN.md section 4.3.
      // Worth 1.6 % here (0.976 -> 0.960 ms per 64 channels): the tile's loads are not what its workgroup waits for.  (The
      // same rewrite of fdiv() itself costs the pass-1 kernels 10-17 more spilled registers and loses 3 %.)
      const int row = (int)(__umulhi((unsigned)i, P.dc2.magic) + (unsigned)i * P.c2_one);
      a[(i - row * C) * LD + row] = raw[u];
    }
  }
  twiddle_lds_put(twl, twv, tid);
  __syncthreads();
  SM_STAMP(s1);
  const cd* r = a;
  if (!(IRA_ABL(P.ablate & 16))) lds_fft_dif_inplace(a, LD, P.p2, twl, tid, C);
  SM_STAMP(s2);
  const long long n = P.n;
  if (OUT == SM_OUT_SPEC_PAIRS) {
    // The tile holds Z = DFT_n(x[2m] + i x[2m+1]) on rows a and b = the mirror of a (tile 0: rows 0 and n1/2, each its own
    // mirror): Z[k] and Z[n - k] are both here, and the real signal's spectrum of length 2n leaves this kernel directly,
    //   X[k] = E[k] + W_2n^k O[k],  E = (Z[k] + conj Z[n-k]) / 2,  O = (Z[k] - conj Z[n-k]) / (2i),   k = 0 .. n
    // (round 3 wrote Z out and read it back twice in a separate pass, smooth_half_split_kernel: 11.5 MB per 10 s channel).
    // W_2n^k = exp(-i pi k / n) along a thread's elements (k2 advances by 128: k by 128 n1) by rotation from an exactly
    // reduced start value; a handful of steps.
    const int p = (int)bx;
    const int ra = p, rb = p == 0 ? (N1 >> 1) : N1 - p;
    double cs = 1.0, sn = 0.0;
    {
      const int k2 = tid >> 1;
      const long long k0 = (long long)((tid & 1) ? rb : ra) + (long long)N1 * k2;
      sincospi(-(double)k0 / (double)n, &sn, &cs);
    }
    for (int i = tid; i < N2 * 2; i += SM_THREADS) {
      const int k2 = i >> 1, c = i & 1;
      const long long k = (long long)(c ? rb : ra) + (long long)N1 * k2;
      // the mirror n - k: row 0 pairs k2 with (n2 - k2) mod n2 in the same row; every other row pairs k2 with n2 - 1 - k2 of
      // its mirror row (row n1/2 is its own mirror row)
      const bool self0 = p == 0 && c == 0;
      const int k2m = self0 ? (k2 == 0 ? 0 : N2 - k2) : N2 - 1 - k2;
      const int cm = p == 0 ? c : 1 - c;
      const cd zk = r[c * LD + dif_slot(k2, P.p2)], zl = r[cm * LD + dif_slot(k2m, P.p2)];
      const cd ev = {0.5 * (zk.re + zl.re), 0.5 * (zk.im - zl.im)};
      const cd od = {0.5 * (zk.im + zl.im), 0.5 * (zl.re - zk.re)};
      cd x = {ev.re + (cs * od.re - sn * od.im), ev.im + (cs * od.im + sn * od.re)};
      if (k == 0) x.im = 0.0;                                              // DC of a real signal
      J.spec_out[out1 + k] = x;
      if (k == 0) J.spec_out[out1 + n] = cd{zk.re - zk.im, 0.0};           // Nyquist bin k = n: E[0] - O[0]
      const double nc = cs * P.pstep_c - sn * P.pstep_s;
      sn = sn * P.pstep_c + cs * P.pstep_s;
      cs = nc;
    }
  } else
  for (int i = tid; i < N2 * C; i += SM_THREADS) {
    const int k2 = (int)fdiv((unsigned)i, P.dc2), c = i - k2 * C;
    const long long k = (long long)(k1_0 + c) + (long long)N1 * k2;        // natural output index
    cd v = r[c * LD + ((IRA_ABL(P.ablate & 128)) ? k2 : dif_slot(k2, P.p2))];
    if (OUT == SM_OUT_SPEC || OUT == SM_OUT_SPEC_PAIRS) {
      if (paired) {
        J.zpair[out1 + k] = v;
      } else if (k <= n / 2) {
        if (k == 0 || 2 * k == n) v.im = 0.0;
        J.spec_out[out1 + k] = v;
      }
    } else if (J.half_out) {
      const double sc = 1.0 / (double)n;                    // samples 2 k and 2 k + 1 of the band signal
      J.y[out1 + 2 * k] = (float)(v.re * sc);
      J.y[out1 + 2 * k + 1] = (float)(-v.im * sc);
    } else {
      const double sc = 1.0 / (double)n;
      const long long ko = (IRA_ABL(P.ablate & 1)) ? (long long)bx * total2 + i : k;
      J.y[out1 + ko] = (float)(v.re * sc);
      if (out2 >= 0) J.y[out2 + ko] = (float)(-v.im * sc);
    }
  }
  if (OUT == SM_OUT_BANDS && J.tile_part != nullptr) band_tile_partials(P, J, r, LD, k1_0, e, bx, out2 >= 0, tid);
  if (IRA_ABL(P.stamp)) {
    SM_STAMP(s3);
    if (tid == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2)
      printf("SMOOTH rows<%d> N2 %d C %d: load %llu  fft %llu  output %llu cycles\n", OUT, N2, C, s1 - s0, s2 - s1, s3 - s2);
  }
}


// ---- narrow-band path, step 2: pass 2 with the pruned pass 1 as its input stage.  grid (N1 / C, jobs) ----------------------
// Element (c, n2) of the tile is  W_n^(k1 n2) * [ sum_m xp[n2 + N2 m] R[m] + sum_m' xn[N2 m' - n2] conj(R[m']) ]  with
// R[m] = W_N1^(k1 m), xp / xn the two compacted clusters (i = n2 + N2 m and i = N2 m' - n2 run over the bins of [lo, hi);
// the mirror element n - i sits at n1 = N1 - m').  R for the ~q + 2 values of m a job can meet lives in LDS per row, the
// cluster values come from L2 (2 w values per job, read by every tile of the job), W_n^(k1 n2) from the same two global
// tables pass 1 uses.  Everything after the input stage is smooth_rows_kernel<SM_OUT_BANDS>.
__global__ __launch_bounds__(SM_THREADS, 5) void smooth_rows_sparse_kernel(SmoothPlan P, SJobs J, const cd* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cd* a = reinterpret_cast<cd*>(smem);
  const int C = P.c2, N1 = P.n1, N2 = P.n2;
  const int LD = P.ld2;
  cd* twl = a + (size_t)C * LD;
  cd* rt = twl + SM_TW;                                    // per-row tables: C rows of SM_SPARSE_R + SM_TW entries
  unsigned bx, by;
  smooth_remap(bx, by);
  const int e = (int)by, tid = threadIdx.x;
  const int32_t* inf = J.info + (size_t)SM_SPARSE_INFO * e;
  const int i_narrow = inf[0], i_lo = inf[1], i_w = inf[2], i_q = inf[3];
  const long long y1 = J.y1_off[e], y2 = J.y2_off[e];
  if (ira::uniform(i_narrow) == 0) return;
  const int lo = ira::uniform(i_lo), W = ira::uniform(i_w), Q = ira::uniform(i_q);
  const int Wp = W + N2;                                   // a cluster and its zero padding
  const long long out1 = ira::uniform(y1), out2 = ira::uniform(y2);
  const int k1_0 = (int)bx * C;
  const cd* z = work + (long long)e * P.n;                 // xp = z[0 .. Wp), xn = z[Wp .. 2 Wp)
  const cd twv = twiddle_lds_fetch(P.t2, N2, tid);
  const int mb = (int)fdiv((unsigned)lo, P.dn2);           // smallest m any element can need
  // per-row tables: R[m], m = mb .. mb + 15, and W_n^(k1 n2) = T[n2 >> 5] F[n2 & 31] (33 + 33 entries, the layout of the
  // sub-transform's own table) -- all from the two global tables pass 1 uses: W_n^p = W_N1^(p / N2) W_n^(p mod N2)
  for (int j = tid; j < C * (SM_SPARSE_R + SM_TW); j += SM_THREADS) {
    const int c = j / (SM_SPARSE_R + SM_TW), q = j - c * (SM_SPARSE_R + SM_TW);
    const unsigned k1 = (unsigned)(k1_0 + c);
    cd v;
    if (q < SM_SPARSE_R) {
      const unsigned p = k1 * (unsigned)(mb + q);                      // < 2^21
      v = P.t1[p - fdiv(p, P.dn1) * (unsigned)N1];
    } else {
      const int t = q - SM_SPARSE_R;
      const unsigned p = k1 * (unsigned)(t < 33 ? 32 * t : t - 33);    // < 2^21
      const unsigned ph = fdiv(p, P.dn2), pl = p - ph * (unsigned)N2;
      v = ira::cmul(P.t1[ph - fdiv(ph, P.dn1) * (unsigned)N1], P.tf[pl]);
    }
    rt[j] = v;
  }
  twiddle_lds_put(twl, twv, tid);
  __syncthreads();
  // A thread owns COLUMNS: n2 = tid, tid + 256, ... for a pair of rows at a time -- the cluster values are the same for every
  // row, so one load feeds both rows' sums, and the whole tile is one batch (Q dependent round trips to L2 per tile instead
  // of 2 Q: what this stage costs is their latency, tools/experiments/r4_sparse_ablate.sh).
  constexpr int NU = 3;                                    // columns per thread and batch (750 = 3 x 256 - 18)
  constexpr int RS = SM_SPARSE_R + SM_TW;                  // table entries per row
  for (int cg = 0; cg < C; cg += 2) {
    const bool two = cg + 1 < C;
    const cd* ra = rt + cg * RS;
    const cd* rb = rt + (two ? cg + 1 : cg) * RS;
    for (int base = 0; base < N2; base += SM_THREADS * NU) {
      cd acc0[NU], acc1[NU];
      int zp[NU], zn[NU], rp[NU], rn[NU];                  // first value in each cluster (then every N2-th), first R entries
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int n2 = base + tid + SM_THREADS * u;
        n2 = n2 < N2 ? n2 : N2 - 1;
        const int d = lo - n2;
        const int m0 = d > 0 ? (int)fdiv((unsigned)(d + N2 - 1), P.dn2) : 0;         // first m with n2 + N2 m >= lo
        int m1 = (int)fdiv((unsigned)(lo + n2 + N2 - 1), P.dn2);                     // first m' with N2 m' - n2 >= lo ...
        m1 = m1 < 1 ? 1 : m1;                                                        // ... and > 0
        zp[u] = n2 + N2 * m0 - lo;
        zn[u] = Wp + (N2 * m1 - n2 - lo);
        rp[u] = m0 - mb;
        rn[u] = m1 - mb;
        acc0[u] = {0.0, 0.0};
        acc1[u] = {0.0, 0.0};
      }
      // (no range tests: the clusters are zero-padded by N2 values, band_compact_kernel)
      for (int m = 0; m < ((IRA_ABL(P.ablate & 256)) ? 0 : Q); ++m) {
        cd xp[NU], xn[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          xp[u] = z[zp[u] + N2 * m];
          xn[u] = z[zn[u] + N2 * m];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const cd p0 = ra[rp[u] + m], q0 = ra[rn[u] + m], p1 = rb[rp[u] + m], q1 = rb[rn[u] + m];
          acc0[u].re = fma(xp[u].re, p0.re, acc0[u].re); acc0[u].re = fma(-xp[u].im, p0.im, acc0[u].re);
          acc0[u].im = fma(xp[u].re, p0.im, acc0[u].im); acc0[u].im = fma(xp[u].im, p0.re, acc0[u].im);
          acc0[u].re = fma(xn[u].re, q0.re, acc0[u].re); acc0[u].re = fma(xn[u].im, q0.im, acc0[u].re);    // conj(R)
          acc0[u].im = fma(xn[u].im, q0.re, acc0[u].im); acc0[u].im = fma(-xn[u].re, q0.im, acc0[u].im);
          acc1[u].re = fma(xp[u].re, p1.re, acc1[u].re); acc1[u].re = fma(-xp[u].im, p1.im, acc1[u].re);
          acc1[u].im = fma(xp[u].re, p1.im, acc1[u].im); acc1[u].im = fma(xp[u].im, p1.re, acc1[u].im);
          acc1[u].re = fma(xn[u].re, q1.re, acc1[u].re); acc1[u].re = fma(xn[u].im, q1.im, acc1[u].re);
          acc1[u].im = fma(xn[u].im, q1.re, acc1[u].im); acc1[u].im = fma(-xn[u].re, q1.im, acc1[u].im);
        }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int n2 = base + tid + SM_THREADS * u;
        if (n2 >= N2) continue;
        const int hi5 = SM_SPARSE_R + (n2 >> 5), lo5 = SM_SPARSE_R + 33 + (n2 & 31);
        cd w0 = {1.0, 0.0}, w1 = {1.0, 0.0};
        if (!(IRA_ABL(P.ablate & 512))) {
          w0 = ira::cmul(ra[hi5], ra[lo5]);
          w1 = ira::cmul(rb[hi5], rb[lo5]);
        }
        a[cg * LD + n2] = ira::cmul(acc0[u], w0);
        if (two) a[(cg + 1) * LD + n2] = ira::cmul(acc1[u], w1);
      }
    }
  }
  __syncthreads();
  const cd* r = a;
  if (!(IRA_ABL(P.ablate & 1024))) lds_fft_dif_inplace(a, LD, P.p2, twl, tid, C);
  const long long n = P.n;
  const double sc = 1.0 / (double)n;
  for (int i = tid; i < N2 * C; i += SM_THREADS) {
    const int k2 = (int)fdiv((unsigned)i, P.dc2), c = i - k2 * C;
    const long long k = (long long)(k1_0 + c) + (long long)N1 * k2;
    const cd v = r[c * LD + dif_slot(k2, P.p2)];
    if ((IRA_ABL(P.ablate & 2048)) && v.re != 12345.678) continue;
    if (J.half_out) {
      J.y[out1 + 2 * k] = (float)(v.re * sc);
      J.y[out1 + 2 * k + 1] = (float)(-v.im * sc);
    } else {
      J.y[out1 + k] = (float)(v.re * sc);
      if (out2 >= 0) J.y[out2 + k] = (float)(-v.im * sc);
    }
  }
  if (J.tile_part != nullptr) band_tile_partials(P, J, r, LD, k1_0, e, bx, out2 >= 0, tid);
}

// split of Z = DFT(x1 + i x2) into the two half spectra (same convention as pair_split_kernel in ira_fftlong.hip)
__global__ __launch_bounds__(256) void smooth_pair_split_kernel(SmoothPlan P, SJobs J) {
  const int e = blockIdx.y;
  if (J.x2off == nullptr || J.x2off[e] < 0) return;
  const long long L = P.n;
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > L / 2) return;
  const cd* z = J.zpair + J.zpair_off[e];
  const long long o1 = J.spec_off[e], o2 = J.spec_off2[e];      // before the stores (see SCtx)
  const cd zk = z[k], zl = z[k == 0 ? 0 : L - k];
  J.spec_out[o1 + k] = {0.5 * (zk.re + zl.re), 0.5 * (zk.im - zl.im)};
  J.spec_out[o2 + k] = {0.5 * (zk.im + zl.im), 0.5 * (zl.re - zk.re)};
}

// Interleaved jobs: z[m] = x[2m] + i x[2m+1], Z = DFT_L(z); the real signal's spectrum of length 2L is
//   X[k] = E[k] + W_2L^k O[k],  E = (Z[k] + conj Z[L-k]) / 2,  O = (Z[k] - conj Z[L-k]) / (2i),  k = 0 .. L  (Z index mod L)
// (same convention as half_split_kernel in ira_fftlong.hip)
__global__ __launch_bounds__(256) void smooth_half_split_kernel(SmoothPlan P, SJobs J) {
  const int e = blockIdx.y;
  const long long L = P.n;
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > L) return;
  const cd* z = J.zpair + J.zpair_off[e];
  const long long o1 = J.spec_off[e];                           // before the store (see SCtx)
  const cd zk = z[k == L ? 0 : k], zl = z[(k == 0 || k == L) ? 0 : L - k];
  const cd ev = {0.5 * (zk.re + zl.re), 0.5 * (zk.im - zl.im)};
  const cd od = {0.5 * (zk.im + zl.im), 0.5 * (zl.re - zk.re)};
  double sn, cs;
  sincospi(-(double)k / (double)L, &sn, &cs);                 // W_2L^k = exp(-i pi k / L)
  cd x = {ev.re + (cs * od.re - sn * od.im), ev.im + (cs * od.im + sn * od.re)};
  if (k == 0 || k == L) x.im = 0.0;                            // DC / Nyquist of a real signal
  J.spec_out[o1 + k] = x;
}

// ---- planning ------------------------------------------------------------------------------------------------------------
int factor_radices(int n, int* out) {
  // Few passes of moderate width: 8s, then one of 4 / 10 / 6 / 2 for the remaining twos, 5s, 3s.
  // Returns the count or -1 if n has another prime factor.
  int cnt = 0, twos = 0, threes = 0, fives = 0;
  while (n % 2 == 0) { n /= 2; ++twos; }
  while (n % 3 == 0) { n /= 3; ++threes; }
  while (n % 5 == 0) { n /= 5; ++fives; }
  if (n != 1) return -1;
  // No radix-16 passes: a 16-point butterfly holds 64 data registers and pins the kernels at ~120 VGPRs (4 waves per
  // SIMD); with 8 / 10 / 6 as the widest radices they need ~80 and five to six workgroups fit a CU.  A lone 2 joins a 5
  // (radix 10) or a 3 (radix 6).
  for (; twos >= 3; twos -= 3) out[cnt++] = 8;
  if (twos == 2) { out[cnt++] = 4; twos = 0; }
  if (twos == 1) {
    if (fives > 0) { out[cnt++] = 10; --fives; }
    else if (threes > 0) { out[cnt++] = 6; --threes; }
    else out[cnt++] = 2;
    twos = 0;
  }
  for (; fives > 0; --fives) { if (cnt >= SM_MAX_RADICES) return -1; out[cnt++] = 5; }
  for (; threes > 0; --threes) { if (cnt >= SM_MAX_RADICES) return -1; out[cnt++] = 3; }
  return cnt;
}

// columns per workgroup for a sub-transform of `len` points (other dimension `other`), dividing the other dimension.
// As many as keep the tile within 26 KB -- the registers allow four workgroups per CU and LDS must not be what stops the fourth; measured for
// 480000 = 640 x 750 (band inverses / forward, ms): columns 2/2 1.48 / 0.61, 3/4 1.50 / 0.63, 5/4 1.61 / 0.67,
// 6/5 2.03 / 0.82 (ping-pong 2/2: 1.62 / 0.64).
int pick_columns(int len, int other) {
  int best = 1;
  for (int c = 1; c <= 8; ++c)
    if (other % c == 0 && ((size_t)c * len + SM_TW) * sizeof(cd) <= 26 * 1024) best = c;
  return best;
}

// LDS column stride of the in-place plan: the column length itself.  Two layouts aimed at the bank conflicts of these
// kernels were built, measured with counters and REMOVED (profiles/r02_smooth_fft_counters.txt):
//   * a pad on the column stride (round 1, commit 33d5c94): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE stayed at 53 % (pass-1
//     kernel) and 37-39 % (pass-2 kernel), times unchanged -- the conflicts are inside a column, not between columns;
//   * a skew inside the column (slot i + (i >> 3), round 2): conflicts of the 640-point transform (radices 8, 8, 10) fell
//     to 36 % and its LDS-active cycles by 27 %, the 750-point transform (10, 5, 5, 3: no power-of-two strides to break)
//     rose to 49 %, and the band inverses went 1.31 -> 1.37 ms: the kernels are not bound by LDS cycles (LDS instructions
//     issue in 1.5 % of the wave cycles, 53 % of them wait on memory and barriers), and the 12 % of extra LDS per
//     workgroup cost one resident workgroup per CU in pass 2.
int column_stride(int len, int c) {
  (void)c;
  return len;
}

bool smooth_split(long long n, int* n1_out, int* n2_out) {
  if (n < 64 || n > (long long)SM_MAX_N * SM_MAX_N) return false;
  long long m = n;
  for (int p : {2, 3, 5}) while (m % p == 0) m /= p;
  if (m != 1) return false;
  // Among the pairs n1 * n2 = n with both <= SM_MAX_N take the most balanced one (measured on MI355X for n = 480000:
  // 640 x 750 beats 960 x 500 and 480 x 1000 although the latter allow more columns per workgroup: the smaller LDS
  // footprint, i.e. more resident workgroups, matters more).  IRA_SMOOTH_N1 forces n1 (tuning).
  const int forced = ira_tune_int("IRA_SMOOTH_N1", 0);
  long long best_score = -1;
  int best = 0;
  for (int d = 2; d <= SM_MAX_N; ++d) {
    if (n % d != 0 || n / d > SM_MAX_N || n / d < 2) continue;
    const int n1 = d, n2 = (int)(n / d);
    if (forced > 0 && n1 != forced) continue;
    const long long score = 2 * (SM_MAX_N - (n1 > n2 ? n1 - n2 : n2 - n1)) + (n1 <= n2 ? 1 : 0);
    if (score > best_score) { best_score = score; best = d; }
  }
  if (best < 2) return false;
  *n1_out = best;
  *n2_out = (int)(n / best);
  return true;
}

// radix passes of an N-point transform (radices from factor_radices or a tuning override); false if they do not fit PassTab
bool fill_passes(int N, const int* radices, int count, PassTab* T) {
  if (count < 1 || count > SM_MAXP) return false;
  *T = PassTab{};
  T->nr = count;
  int len = N;
  for (int i = 0; i < SM_MAXP; ++i) {
    const int r = i < count ? radices[i] : 2;                              // unused entries: harmless constants
    const int m = i < count ? len / r : 1;
    T->r[i] = r;
    T->len[i] = i < count ? len : 2;
    T->scale[i] = i < count ? N / len : 1;
    T->span[i] = i < count ? len / r : 0;
    T->per[i] = (unsigned)(N / r);
    T->per_magic[i] = fast_div_of((unsigned)(N / r)).magic;
    T->m_magic[i] = fast_div_of((unsigned)m).magic;
    T->r_magic[i] = fast_div_of((unsigned)r).magic;
    if (i < count) len /= r;
  }
  return len == 1;
}

int32_t make_smooth_plan(int32_t n, const void* t1, const void* t2, const void* tf, SmoothPlan* P, bool pairs = false) {
  int n1, n2;
  if (!smooth_split(n, &n1, &n2)) return IRA_E_UNSUPPORTED;
  if (pairs && (n1 % 2 != 0 || n1 < 4)) return IRA_E_UNSUPPORTED;
  P->n = n; P->n1 = n1; P->n2 = n2;
  int r1[SM_MAX_RADICES], r2[SM_MAX_RADICES];
  int nr1 = factor_radices(n1, r1), nr2 = factor_radices(n2, r2);
  if (nr1 < 0 || nr2 < 0) return IRA_E_UNSUPPORTED;
  // tuning: IRA_SMOOTH_R1 / IRA_SMOOTH_R2 = comma-separated radix order for the n1- / n2-point transforms (product checked)
  auto override_radices = [](const char* name, int len, int* out, int* cnt) {
    const char* ev = ira_tune_str(name);
    if (!ev) return;
    int tmp[SM_MAX_RADICES], k = 0;
    long long prod = 1;
    for (const char* c = ev; *c && k < SM_MAX_RADICES;) {
      const int r = std::atoi(c);
      if (!(r == 2 || r == 3 || r == 4 || r == 5 || r == 6 || r == 8 || r == 10)) return;
      tmp[k++] = r; prod *= r;
      while (*c && *c != ',') ++c;
      if (*c == ',') ++c;
    }
    if (prod != len) return;
    for (int i = 0; i < k; ++i) out[i] = tmp[i];
    *cnt = k;
  };
  override_radices("IRA_SMOOTH_R1", n1, r1, &nr1);
  override_radices("IRA_SMOOTH_R2", n2, r2, &nr2);
  if (!fill_passes(n1, r1, nr1, &P->p1) || !fill_passes(n2, r2, nr2, &P->p2)) return IRA_E_UNSUPPORTED;
  P->c1 = pick_columns(n1, n2);
  P->c2 = pick_columns(n2, n1);
  { const int v = ira_tune_int("IRA_SMOOTH_C1", 0); if (v >= 1 && n2 % v == 0) P->c1 = v; }
  { const int v = ira_tune_int("IRA_SMOOTH_C2", 0); if (v >= 1 && n1 % v == 0) P->c2 = v; }
  P->pairs = pairs ? 1 : 0;
  if (pairs) P->c2 = 2;                                  // a tile = a row and its mirror row
  {
    const double step = -3.14159265358979323846 * (double)(SM_THREADS / 2) / (double)n2;
    P->pstep_c = std::cos(step);
    P->pstep_s = std::sin(step);
  }
  P->ld1 = column_stride(n1, P->c1);
  P->ld2 = column_stride(n2, P->c2);
  P->dc1 = fast_div_of((unsigned)P->c1); P->dc2 = fast_div_of((unsigned)P->c2);
  P->c2_one = P->c2 <= 1 ? 1u : 0u;
  P->dn1 = fast_div_of((unsigned)n1); P->dn2 = fast_div_of((unsigned)n2);
  P->t1 = static_cast<const cd*>(t1); P->t2 = static_cast<const cd*>(t2); P->tf = static_cast<const cd*>(tf);
  {
    // narrow-band path: up to 9 terms per cluster and point (measured, config 3: see DESIGN.md section 4.6)
    int q = ira_tune_int("IRA_SPARSE_Q", 9);
    P->sparse_q = q < 0 ? 0 : (q > SM_SPARSE_QMAX ? SM_SPARSE_QMAX : q);
  }
  P->stamp = ira_tune_flag("IRA_SMOOTH_STAMP");
  P->ablate = ira_tune_int("IRA_SMOOTH_ABLATE", 0);
  {
    const double step = 3.14159265358979323846 * (double)(SM_THREADS / P->c1) / (double)n1;
    P->hstep_c = std::cos(step);
    P->hstep_s = std::sin(step);
  }
  return IRA_OK;
}

template <typename K>
hipError_t allow(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#define SM_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return ira_hip_status(_e); } while (0)

}  // namespace

// n = n1 * n2 with both factors <= 1024 and n = 2^a 3^b 5^c: IRA_OK and the split; otherwise IRA_E_UNSUPPORTED (use the
// Bluestein entry points).  (The A/B switch is the caller's: Engine.smooth_ffts.)
extern "C" int32_t ira_fft_smooth_split(int32_t n, int32_t* n1, int32_t* n2) {
  IRA_CHECK_PTR(n1); IRA_CHECK_PTR(n2);
  int a, b;
  if (!smooth_split(n, &a, &b)) return IRA_E_UNSUPPORTED;
  int r[SM_MAX_RADICES];
  const int ca = factor_radices(a, r), cb = factor_radices(b, r);
  if (ca < 1 || cb < 1 || ca > SM_MAXP || cb > SM_MAXP) return IRA_E_UNSUPPORTED;
  *n1 = a; *n2 = b;
  return IRA_OK;
}

extern "C" int32_t ira_rfft_smooth(const float* x_dev, const int64_t* xoff_dev, int32_t n, int32_t nb, int32_t use_hann,
                                   const void* t1_dev, const void* t2_dev, const void* tf_dev, double* work_dev,
                                   double* spec_out_dev, const int64_t* spec_off_dev, const int64_t* x2off_dev,
                                   const int64_t* spec_off2_dev, double* zpair_dev, const int64_t* zpair_off_dev,
                                   const int32_t* data_len_dev, const int32_t* win_len_dev,
                                   const int32_t* data_len2_dev, const int32_t* win_len2_dev, int32_t interleave,
                                   void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(xoff_dev); IRA_CHECK_PTR(t1_dev); IRA_CHECK_PTR(t2_dev); IRA_CHECK_PTR(tf_dev);
  IRA_CHECK_PTR(work_dev); IRA_CHECK_PTR(spec_out_dev); IRA_CHECK_PTR(spec_off_dev);
  if (nb <= 0) return nb == 0 ? IRA_OK : IRA_E_SIZE;
  if (nb > 65535) return IRA_E_SIZE;
  if (interleave && (data_len2_dev != nullptr || win_len2_dev != nullptr)) return IRA_E_NULL;
  // interleave without zpair scratch: the untangling is fused into pass 2 (mirror-pair tiles; needs an even n1 -- callers
  // ask ira_fft_smooth_split); with zpair: round 3's separate split pass (any n1)
  const bool pairs = interleave && zpair_dev == nullptr;
  if (interleave && !pairs && x2off_dev == nullptr) return IRA_E_NULL;
  SmoothPlan P;
  const int32_t rc = make_smooth_plan(n, t1_dev, t2_dev, tf_dev, &P, pairs);
  if (rc != IRA_OK) return rc;
  SJobs J{};
  J.x = x_dev; J.xoff = xoff_dev; J.use_hann = use_hann; J.interleave = interleave ? 1 : 0;
  J.data_len = data_len_dev; J.win_len = win_len_dev; J.data_len2 = data_len2_dev; J.win_len2 = win_len2_dev;
  J.spec_out = reinterpret_cast<cd*>(spec_out_dev); J.spec_off = spec_off_dev;
  if (x2off_dev != nullptr && !pairs) {
    if (spec_off2_dev == nullptr || zpair_dev == nullptr || zpair_off_dev == nullptr) return IRA_E_NULL;
    J.x2off = x2off_dev; J.spec_off2 = spec_off2_dev;
    J.zpair = reinterpret_cast<cd*>(zpair_dev); J.zpair_off = zpair_off_dev;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t l1 = ((size_t)P.c1 * P.ld1 + SM_TW) * sizeof(cd), l2 = ((size_t)P.c2 * P.ld2 + SM_TW) * sizeof(cd);
  cd* work = reinterpret_cast<cd*>(work_dev);
  if (pairs) {
    SM_TRY(allow(smooth_cols_kernel<SM_SIGNAL, true>, l1));
    SM_TRY(allow(smooth_rows_kernel<SM_OUT_SPEC_PAIRS>, l2));
    smooth_cols_kernel<SM_SIGNAL, true><<<dim3(P.n2 / P.c1, nb), SM_THREADS, l1, st>>>(P, J, work);
    smooth_rows_kernel<SM_OUT_SPEC_PAIRS><<<dim3(P.n1 / 2, nb), SM_THREADS, l2, st>>>(P, J, work);
    IRA_RETURN_LAUNCH();
  }
  SM_TRY(allow(smooth_cols_kernel<SM_SIGNAL>, l1));
  SM_TRY(allow(smooth_rows_kernel<SM_OUT_SPEC>, l2));
  smooth_cols_kernel<SM_SIGNAL><<<dim3(P.n2 / P.c1, nb), SM_THREADS, l1, st>>>(P, J, work);
  smooth_rows_kernel<SM_OUT_SPEC><<<dim3(P.n1 / P.c2, nb), SM_THREADS, l2, st>>>(P, J, work);
  if (interleave)
    smooth_half_split_kernel<<<dim3((n + 1 + 255) / 256, nb), 256, 0, st>>>(P, J);
  else if (x2off_dev != nullptr)
    smooth_pair_split_kernel<<<dim3((n / 2 + 1 + 255) / 256, nb), 256, 0, st>>>(P, J);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_band_tile_layout(int32_t n, int32_t half_out, int32_t* tiles, int32_t* workgroups) {
  IRA_CHECK_PTR(tiles); IRA_CHECK_PTR(workgroups);
  SmoothPlan P;
  const double dummy[2] = {0.0, 0.0};                            // make_smooth_plan only stores the table pointers
  const int32_t rc = make_smooth_plan(n, dummy, dummy, dummy, &P);
  if (rc != IRA_OK) return rc;
  const long long len = half_out ? 2ll * n : (long long)n;
  *tiles = (int32_t)((len + SM_EDC_TILE - 1) / SM_EDC_TILE);
  *workgroups = P.n1 / P.c2;
  return IRA_OK;
}

extern "C" int32_t ira_band_irfft_smooth(const double* spec_dev, const int64_t* spec_off_dev, int32_t n, int32_t nb,
                                         const double* band_params_dev, const double* freq_val_dev,
                                         const void* t1_dev, const void* t2_dev, const void* tf_dev, double* work_dev,
                                         float* y_dev, const int64_t* y1_off_dev, const int64_t* y2_off_dev,
                                         const int64_t* spec_off2_dev, int32_t half_out, int32_t* job_info_dev,
                                         double* tile_part_dev, void* stream) {
  IRA_CHECK_PTR(spec_dev); IRA_CHECK_PTR(spec_off_dev); IRA_CHECK_PTR(band_params_dev); IRA_CHECK_PTR(freq_val_dev);
  IRA_CHECK_PTR(t1_dev); IRA_CHECK_PTR(t2_dev); IRA_CHECK_PTR(tf_dev); IRA_CHECK_PTR(work_dev); IRA_CHECK_PTR(y_dev);
  IRA_CHECK_PTR(y1_off_dev); IRA_CHECK_PTR(y2_off_dev);
  if (nb <= 0) return nb == 0 ? IRA_OK : IRA_E_SIZE;
  if (nb > 65535) return IRA_E_SIZE;
  SmoothPlan P;
  const int32_t rc = make_smooth_plan(n, t1_dev, t2_dev, tf_dev, &P);
  if (rc != IRA_OK) return rc;
  static_assert(sizeof(BandMaskS) == 8 * sizeof(double), "band parameter record is 8 doubles");
  SJobs J{};
  J.spec = reinterpret_cast<const cd*>(spec_dev); J.sp_off = spec_off_dev; J.sp_off2 = spec_off2_dev;
  J.bands = reinterpret_cast<const BandMaskS*>(band_params_dev); J.freq_val = freq_val_dev;
  J.y = y_dev; J.y1_off = y1_off_dev; J.y2_off = y2_off_dev;
  J.half_out = half_out ? 1 : 0;
  J.tile_part = tile_part_dev;
  if (half_out && spec_off2_dev != nullptr) return IRA_E_UNSUPPORTED;       // one band of one spectrum per job
  hipStream_t st = (hipStream_t)stream;
  const size_t l1 = ((size_t)P.c1 * P.ld1 + SM_TW) * sizeof(cd), l2 = ((size_t)P.c2 * P.ld2 + SM_TW) * sizeof(cd);
  SM_TRY(allow(smooth_cols_kernel<SM_SPECTRUM, false>, l1));
  SM_TRY(allow(smooth_cols_kernel<SM_SPECTRUM, true>, l1));
  SM_TRY(allow(smooth_rows_kernel<SM_OUT_BANDS>, l2));
  cd* work = reinterpret_cast<cd*>(work_dev);
  if (job_info_dev != nullptr && P.sparse_q > 0) {
    // narrow jobs: compact + one fused pass; the two regular passes below return at once for them (and these for the rest)
    J.info = job_info_dev;
    const size_t ls = l2 + (size_t)P.c2 * (SM_SPARSE_R + SM_TW) * sizeof(cd);
    SM_TRY(allow(smooth_rows_sparse_kernel, ls));
    const unsigned cb = (unsigned)(((long long)(P.sparse_q + 1) * P.n2 + 255) / 256);     // a cluster + its padding
    if (half_out) band_compact_kernel<true><<<dim3(cb, nb), 256, 0, st>>>(P, J, work);
    else band_compact_kernel<false><<<dim3(cb, nb), 256, 0, st>>>(P, J, work);
    smooth_rows_sparse_kernel<<<dim3(P.n1 / P.c2, nb), SM_THREADS, ls, st>>>(P, J, work);
  }
  if (half_out) smooth_cols_kernel<SM_SPECTRUM, true><<<dim3(P.n2 / P.c1, nb), SM_THREADS, l1, st>>>(P, J, work);
  else smooth_cols_kernel<SM_SPECTRUM, false><<<dim3(P.n2 / P.c1, nb), SM_THREADS, l1, st>>>(P, J, work);
  smooth_rows_kernel<SM_OUT_BANDS><<<dim3(P.n1 / P.c2, nb), SM_THREADS, l2, st>>>(P, J, work);
  IRA_RETURN_LAUNCH();
}


// ---- a8 on its own: the band mask values the inverse transforms apply (float32, on the float32 frequency axis) -----------
namespace {
__global__ void band_mask_values_kernel(ira::BandMask band, double freq_val, long long nbins, float* __restrict__ out) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nbins) out[k] = ira::mask_at(band, (float)((double)k * freq_val));
}
}  // namespace

extern "C" int32_t ira_band_mask_values(const double* band_params8, double freq_val, int64_t nbins, float* mask_dev,
                                        void* stream) {
  IRA_CHECK_PTR(band_params8); IRA_CHECK_PTR(mask_dev);
  if (nbins < 0) return IRA_E_SIZE;
  if (nbins == 0) return IRA_OK;
  ira::BandMask b;
  static_assert(sizeof(ira::BandMask) == 8 * sizeof(double), "band parameter record is 8 doubles");
  std::memcpy(&b, band_params8, sizeof(b));
  band_mask_values_kernel<<<(unsigned)((nbins + 255) / 256), 256, 0, (hipStream_t)stream>>>(b, freq_val, nbins, mask_dev);
  IRA_RETURN_LAUNCH();
}
