// Arbitrary-length float64 DFTs of whole impulse responses (lengths are data dependent and generally not
// powers of two: N - argmax|x|), used by
//   * frequency response / filter response: rfft(x * hanning(L))        (reference frequency_response.py:204-213,
//                                                                         filterplot.py:145-152)
//   * RT60 bands: rfft(x) once, then irfft(spectrum * mask, n) per band  (reference rt60bands.py:170-175)
//
// Method: Bluestein (chirp-z).  With w[n] = exp(-i*pi*n^2/L):
//     X[k] = w[k] * sum_n (x[n] w[n]) * conj(w[k-n])
// i.e. one circular convolution of length M = 2^m >= 2L-1, done with a four-step FFT M = N1 x N2 whose
// sub-transforms run in LDS (ira_fft_lds.h):
//   K1 cols_fwd : generate the input on the fly (samples*window*chirp | chirp filter | masked spectrum*chirp),
//                 N1-point DIF down each column (stride N2), twiddle W_M^(n2*k1), store in place
//   K2 rows     : N2-point DIF along each row, multiply by the filter spectrum B (same permuted layout), N2-point
//                 inverse DIT, inverse twiddle -- the row never leaves LDS in between
//   K3 cols_inv : N1-point inverse DIT down each column, fused epilogue (spectrum bins | two band signals)
// The DIF leaves results bit-reversed and the DIT consumes bit-reversed input, so no transposes or reorder
// passes exist; K1->K2->K3 each read and write the M*16-byte work array once.
// All arithmetic is float64: the reference computes these FFTs in float64 (numpy pocketfft) and the unwrapped
// phase / band RT60 values are discrete functions of them.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "ira_bandmask.h"
#include "ira_fft_lds.h"

namespace {

using ira::cplx;
typedef cplx<double> cd;

constexpr int FL_THREADS = 256;
// In-LDS sub-FFTs in radix-16 passes (3 passes for 1024 points).  Radix 8 (FL_LR = 3: 4 passes, ~80 VGPRs instead of 125)
// was measured too: 1.66 ms against 1.61 ms for the fr/filter spectra -- at C = R = 2 these kernels are bound by LDS capacity
// (4 workgroups per CU) and by memory (K2 moves 3.2 TB/s), not by registers; smaller tiles (C = R = 1) were slower still.
constexpr int FL_LR = 4;

// M = N1 x N2 with N2 = 2^log2n2 and N1 = rad * Q, Q = 2^log2q, rad = 1 or 3: convolution sizes 2^k and 3 * 2^k, so that
// M exceeds what a job needs by at most a third instead of up to 2x (half of all real-file lengths are odd and need
// M >= 1.5 L: 3 * 2^18 instead of 2^20 at 10 s).  A column of N1 = 3 Q points is ONE radix-3 decimation-in-frequency stage
// (three blocks of Q: block b holds the frequencies k1 = 3 k' + b) around the power-of-two sub-FFTs of ira_fft_lds.h, which
// then run on rad * C blocks of Q points.  In LDS a block is padded to Q + 1 entries.
struct Geom {
  long long m;
  int n1, rad, log2q, log2n2;
  const cd* t1;  // exp(-2 pi i k / N1), k < N1   (full circle; first half doubles as the sub-FFT table, period N1 = rad * Q)
  const cd* t2;  // exp(-2 pi i k / N2), k < N2
  const cd* tf;  // exp(-2 pi i k / M),  k < N2
  int ablate;    // diagnostics (IRA_FFT_ABLATE): 1 K1 plain input, 2 K1 no FFT, 4 K1 no store, 8 K2 no FFTs,
                 // 16 K2 no filter multiply, 32 K3 no FFT, 64 K3 plain epilogue, 128 K2 no store, 256 K3 no load, 512 K3 (LDS-DMA) no store
};

// LDS slot (inside one column) of column element / work-array row `r` < N1, and the column frequency k1 that row r of the
// work array holds after the forward column pass (rows stay in this permuted order through K2; K3's DIT consumes it).
__device__ __forceinline__ unsigned col_slot(const Geom& g, unsigned r) {
  return (r >> g.log2q) * ((1u << g.log2q) + 1u) + (r & ((1u << g.log2q) - 1u));
}
__device__ __forceinline__ unsigned row_k1(const Geom& g, unsigned r) {
  return (unsigned)g.rad * ira::lds_brev(r & ((1u << g.log2q) - 1u), g.log2q) + (r >> g.log2q);
}

// forward radix-3 butterfly (W = exp(-2 pi i / 3)), natural order in and out
__device__ __forceinline__ void bfly3(cd& a0, cd& a1, cd& a2) {
  const cd t1 = ira::cadd(a1, a2);
  const cd t2 = {a0.re - 0.5 * t1.re, a0.im - 0.5 * t1.im};
  const cd d = ira::csub(a1, a2);
  const cd r = {0.86602540378443864676 * d.im, -0.86602540378443864676 * d.re};     // -i sin(pi/3) (a1 - a2)
  a0 = ira::cadd(a0, t1);
  a1 = ira::cadd(t2, r);
  a2 = ira::csub(t2, r);
}

// The radix-3 stage of a column of N1 = 3 Q points held as three blocks of Q (+1 pad) in LDS, C columns `stride` apart.
// Forward (decimation in frequency): y_b[j] = W_N1^(j b) sum_r x[j + r Q] W_3^(r b) replaces x[j + b Q]; block b is then
// the input of the Q-point transform that yields X[3 k' + b].  Inverse: the exact reverse (unnormalised).
template <bool INVERSE>
__device__ __forceinline__ void radix3_stage(const Geom& g, cd* lds, const cd* twl, int C, unsigned stride, int tid) {
  const unsigned Q = 1u << g.log2q, qs = Q + 1u, half = (unsigned)g.n1 >> 1;
  for (unsigned idx = tid; idx < Q * (unsigned)C; idx += FL_THREADS) {
    const unsigned c = idx >> g.log2q, j = idx & (Q - 1u);
    cd* p = lds + c * stride + j;
    cd a0 = p[0], a1 = p[qs], a2 = p[2 * qs];
    cd w1 = ira::tw_get<double, true>(twl, j);                       // W_N1^j       (j < Q < N1/2)
    cd w2 = ira::tw_lookup<double, true>(twl, 2u * j, half);         // W_N1^(2 j)   (2 j < N1)
    if (INVERSE) {
      w1.im = -w1.im; w2.im = -w2.im;
      a1 = ira::cmul(a1, w1);
      a2 = ira::cmul(a2, w2);
      a0.im = -a0.im; a1.im = -a1.im; a2.im = -a2.im;                // inverse 3-point DFT = conj(DFT(conj .))
      bfly3(a0, a1, a2);
      a0.im = -a0.im; a1.im = -a1.im; a2.im = -a2.im;
    } else {
      bfly3(a0, a1, a2);
      if (j != 0) {
        a1 = ira::cmul(a1, w1);
        a2 = ira::cmul(a2, w2);
      }
    }
    p[0] = a0; p[qs] = a1; p[2 * qs] = a2;
  }
  __syncthreads();
}

// Phase of a chirp value: q / L half-turns reduced to [0, 2), for an exact non-negative integer q < 2^53 held in a double
// (n^2, 2 n dn + dn^2, 2 dn^2 with n, dn < 2^22).  q mod 2L comes out EXACTLY from one fma -- the quotient estimate is off
// by at most one, the remainder is an integer below 2^33 -- instead of a 64-bit integer division (85 instructions; the
// three per-thread chirp start values of K1 / K3 were a third of those kernels' VALU work).
struct ChirpScale {
  double l2, inv_l2, inv_l;       // 2L, 1/(2L), 1/L
};
__device__ __forceinline__ ChirpScale chirp_scale(long long L) {
  const double l = (double)L;
  return {2.0 * l, 0.5 / l, 1.0 / l};
}
__device__ __forceinline__ double chirp_angle(double q, const ChirpScale& cs) {
  const double k = floor(q * cs.inv_l2);
  double r = fma(-k, cs.l2, q);
  r = r < 0.0 ? r + cs.l2 : (r >= cs.l2 ? r - cs.l2 : r);
  return r * cs.inv_l;
}
// exp(-i*pi*q/L)
__device__ __forceinline__ cd unit_q(double q, const ChirpScale& cs) {
  double s, c;
  sincospi(chirp_angle(q, cs), &s, &c);
  return {c, -s};
}
// exp(-i*pi*n^2/L), n < 2^26
__device__ __forceinline__ cd chirp(long long n, const ChirpScale& cs) {
  const double x = (double)n;
  return unit_q(x * x, cs);
}

// value of lane `lane` of the wave in every lane (scalar registers)
__device__ __forceinline__ double lane_value(double v, int lane) {
  int w[2];
  __builtin_memcpy(w, &v, 8);
  w[0] = __builtin_amdgcn_readlane(w[0], lane);
  w[1] = __builtin_amdgcn_readlane(w[1], lane);
  __builtin_memcpy(&v, w, 8);
  return v;
}

// numpy.hanning(L)[i] = 0.5 + 0.5*cos(pi*(2i + 1 - L)/(L - 1)), hanning(1) = 1: evaluated in cols_fwd_kernel as a rotation
// along each thread's elements (exact sincospi start, then one complex multiply per element).

using ira::BandMask;
using ira::mask_at;

// ---- per-element job description (device arrays, one entry per batch element) ----------------------------------
struct Jobs {
  int e0;                   // first element of this launch (run_convolution may launch the three passes over sub-ranges of jobs)
  const int32_t* L;         // transform length of element e
  // signal input
  const float* x;
  const int64_t* xoff;
  const int64_t* x2off;     // optional: second real signal of the same length (-1: none) -> z = x1 + i x2
  int use_hann;
  // optional (null = the transform length L): samples actually read / Hann window length, per signal of the pair.
  // numpy.fft.rfft(x * hanning(len(x)), n=P) zero-pads (len < P) or truncates (len > P) a segment whose window was
  // built for its OWN length (reference group_delay.py:95-109).
  const int32_t* data_len; const int32_t* win_len;
  const int32_t* data_len2; const int32_t* win_len2;
  // optional: 1 = the two "signals" of element e are the EVEN and ODD samples of ONE real signal (x2off = xoff + 1,
  // both read with stride 2; Hann index 2n / 2n+1): a real transform of even length 2L as one complex transform of L
  const int32_t* interleave;
  // masked-spectrum input
  const cd* spec;           // half spectra, complex f64
  const int64_t* spec_off;  // element e reads spec + spec_off[e], (L/2+1) bins
  const int64_t* spec_off2; // optional: the SECOND band's spectrum (another channel of the same length); null = same
  const BandMask* bands;    // 2 per element
  const double* freq_val;   // rfftfreq step of element e: bin k -> float32(k * freq_val[e])
  // filter spectra
  const cd* bfilt;          // [nfilt][M]
  const int32_t* bidx;      // which filter element e uses
  // outputs
  cd* spec_out;
  const int64_t* spec_out_off;
  const int64_t* spec_out_off2;  // second signal's half spectrum (paired elements)
  cd* zpair;                // full-length DFT of x1 + i x2 (paired elements), L entries at zpair_off[e]
  const int64_t* zpair_off;
  float* y;
  const int64_t* y1_off;    // first band signal of element e (length L)
  const int64_t* y2_off;    // second band signal or -1
};

enum InMode { IN_SIGNAL = 0, IN_FILTER = 1, IN_SPECTRUM = 2 };
enum RowMode { ROW_FWD = 0, ROW_CONV = 1 };
enum OutMode { OUT_SPECTRUM = 0, OUT_BANDS = 1 };

// The job's own values, read ONCE per workgroup into scalar registers (ira::uniform, all loads first: one round trip).
// Written as J.xoff[e] / J.bands[2 e] inside the tile loops each of them is a vector load followed by s_waitcnt vmcnt(0),
// i.e. a memory round trip per element (see ira_common.h, uniform()).
struct Ctx {
  long long L;
  long long o1, o2;                 // IN_SIGNAL: sample offsets (o2 < 0: one signal); IN_SPECTRUM: spectrum offsets
  long long st;                     // IN_SIGNAL: sample stride (2 = even / odd samples of one real signal)
  long long nd1, nd2, lw1, lw2;     // IN_SIGNAL: samples actually read / Hann window lengths of the two signals
  bool two;                         // IN_SPECTRUM: the two bands come from two different spectra
  // the same in the form the tile loops use: 32-bit counts, wave-uniform bases
  unsigned Lu, stu;
  unsigned lim1, lim2;              // IN_SIGNAL: samples of signal 1 / 2 that lie inside the transform (0: none)
  const float* p1; const float* p2; // IN_SIGNAL: first sample of each signal (J.x itself when there is nothing to read)
  const cd* s1; const cd* s2;       // IN_SPECTRUM: the two spectra
  BandMask b1, b2;
  ira::MaskCuts k1, k2;             // first bins past each mask edge (ira_bandmask.h)
  double fv;
};

__device__ __forceinline__ BandMask uniform_band(const BandMask& p) {
  BandMask b{};
  b.kind = ira::uniform(p.kind);
  b.hp_x0 = ira::uniform(p.hp_x0); b.hp_x1 = ira::uniform(p.hp_x1);
  b.lp_x0 = ira::uniform(p.lp_x0); b.lp_x1 = ira::uniform(p.lp_x1);
  return b;
}

template <int MODE>
__device__ __forceinline__ Ctx job_ctx(const Jobs& J, int e) {
  Ctx c{};
  const int L = J.L[e];
  if (MODE == IN_SIGNAL) {
    const long long o1 = J.xoff[e];
    const long long o2 = J.x2off ? (long long)J.x2off[e] : -1ll;
    const int il = J.interleave ? J.interleave[e] : 0;
    const int nd1 = J.data_len ? J.data_len[e] : -1, nd2 = J.data_len2 ? J.data_len2[e] : -1;
    const int lw1 = J.win_len ? J.win_len[e] : -1, lw2 = J.win_len2 ? J.win_len2[e] : -1;
    c.L = ira::uniform(L);
    c.o1 = ira::uniform(o1);
    c.o2 = ira::uniform(o2);
    c.st = ira::uniform(il) ? 2 : 1;
    c.nd1 = J.data_len ? (long long)ira::uniform(nd1) : c.L;
    c.nd2 = J.data_len2 ? (long long)ira::uniform(nd2) : c.nd1;
    c.lw1 = J.win_len ? (long long)ira::uniform(lw1) : c.L;
    c.lw2 = J.win_len2 ? (long long)ira::uniform(lw2) : c.lw1;
    c.Lu = (unsigned)c.L; c.stu = (unsigned)c.st;
    c.lim1 = (unsigned)(c.nd1 < c.L ? (c.nd1 > 0 ? c.nd1 : 0) : c.L);
    c.lim2 = c.o2 >= 0 ? (unsigned)(c.nd2 < c.L ? (c.nd2 > 0 ? c.nd2 : 0) : c.L) : 0u;
    c.p1 = c.lim1 ? J.x + c.o1 : J.x;
    c.p2 = c.lim2 ? J.x + c.o2 : J.x;
  } else if (MODE == IN_SPECTRUM) {
    const long long o1 = J.spec_off[e];
    const long long o2 = J.spec_off2 ? (long long)J.spec_off2[e] : -1ll;
    const BandMask b1 = J.bands[2 * e], b2 = J.bands[2 * e + 1];
    const double fv = J.freq_val[e];
    c.L = ira::uniform(L);
    c.o1 = ira::uniform(o1);
    c.o2 = J.spec_off2 ? ira::uniform(o2) : c.o1;
    c.two = c.o2 != c.o1;
    c.b1 = uniform_band(b1);
    c.b2 = uniform_band(b2);
    c.fv = ira::uniform(fv);
    ira::band_cuts(c.b1, c.b2, c.fv, (int)(c.L / 2), c.k1, c.k2);
    c.Lu = (unsigned)c.L;
    c.s1 = J.spec + c.o1; c.s2 = J.spec + c.o2;
  } else {
    c.L = ira::uniform(L);
    c.Lu = (unsigned)c.L;
  }
  return c;
}

// Input generation in two phases, so that ALL of a thread's global loads are in flight before the first one is used (a
// load that is waited for inside the element loop costs one full memory round trip per element: 8 per workgroup tile):
// fetch_input does nothing but the loads -- unconditional ones: a lane beyond the data reads sample 0 of the batch buffer
// and value_input drops it; nothing is converted or branched on here -- value_input the arithmetic.  w = chirp(n, L) and
// h = hann_at(n, L) come from the caller (recurrences along a thread's elements, see cols_fwd_kernel); IN_FILTER
// evaluates its mirrored chirp directly (filters are plan data, built once and cached).
struct RawL { double a, b, c, d; float fa, fb; };
constexpr int FL_UI = 4;    // K1: fetches in flight per thread and batch (a fetch is up to four doubles)
constexpr int FL_U = 8;     // K2 / K3: 16-byte loads in flight per thread and batch

template <int MODE>
__device__ __forceinline__ RawL fetch_input(const Jobs& J, const Ctx& c, unsigned n) {
  RawL r{0.0, 0.0, 0.0, 0.0, 0.0f, 0.0f};
  if (MODE == IN_SIGNAL) {
    // 32-bit element indices on wave-uniform 64-bit bases: one compare, one select and one multiply per load (as 64-bit
    // pointer selects the sixteen loads of a tile were a quarter of K1's VALU instructions)
    r.fa = c.p1[n < c.lim1 ? c.stu * n : 0u];
    r.fb = c.p2[n < c.lim2 ? c.stu * n : 0u];
  } else if (MODE == IN_SPECTRUM) {
    const unsigned k = n > c.Lu / 2 ? c.Lu - n : n;
    const unsigned kk = n < c.Lu ? k : 0u;
    const cd xk = c.s1[kk];
    r.a = xk.re; r.b = xk.im;
    if (c.two) {
      const cd x2 = c.s2[kk];
      r.c = x2.re; r.d = x2.im;
    }
  }
  return r;
}

template <int MODE>
__device__ __forceinline__ cd value_input(const Jobs& J, const Ctx& c, unsigned n, unsigned M, cd w, double h,
                                          double h2, const RawL& r, bool hann, bool second) {
  const unsigned L = c.Lu;
  if (MODE == IN_SIGNAL) {
    // lim1 / lim2 = samples of the two signals inside the transform (0: no such signal): a lane beyond them dropped its load
    double v = (double)(n < c.lim1 ? r.fa : 0.0f), v2 = (double)(n < c.lim2 ? r.fb : 0.0f);
    if (hann) {
      v *= h; v2 *= h2;
    }
    if (!second) return {v * w.re, v * w.im};
    return {v * w.re - v2 * w.im, v * w.im + v2 * w.re};
  } else if (MODE == IN_FILTER) {
    // b[j] = conj(w[|j|]) for -(L-1) <= j <= L-1 at position j mod M.  With M >= 2L - 1 the two sides do not meet; a
    // smaller M (>= L + L/2, single real signals whose bins k <= L/2 alone are wanted) lets them overlap, and then the
    // NEGATIVE side wins: output k needs j = k - n in [-(L-1), L/2], and positions above L/2 can only mean j < 0.
    unsigned m = M - n;
    if (m >= L) {
      m = n;
      if (m >= L) return {0.0, 0.0};
    }
    const cd wm = chirp((long long)m, chirp_scale((long long)L));
    return {wm.re, -wm.im};  // conj(w)
  } else {
    if (n >= L) return {0.0, 0.0};
    // Hermitian extension of X * (m1 + i m2); the inverse DFT is conj(DFT(conj(.)))/L, so feed conj(W) * chirp
    const bool upper = n > L / 2;
    const unsigned k = upper ? L - n : n;
    const cd xk = {r.a, upper ? -r.b : r.b};
    const double m1 = (double)ira::mask_cut(c.b1, c.k1, (int)k, c.fv);
    const double m2 = (double)ira::mask_cut(c.b2, c.k2, (int)k, c.fv);
    cd wk;
    if (!c.two) {
      wk = ira::cmul(xk, cd{m1, m2});
    } else {                                        // X1 m1 + i X2 m2: band 1 of one channel, band 2 of another
      const cd x2 = {r.c, upper ? -r.d : r.d};
      wk = {xk.re * m1 - x2.im * m2, xk.im * m1 + x2.re * m2};
    }
    const cd cw = {wk.re, -wk.im};
    return ira::cmul(cw, w);
  }
}

// XCD-aware remap (speed only): give each XCD a contiguous range of (element, tile) pairs so that neighbouring
// column tiles -- which touch the two halves of the same 128-byte lines -- meet in one L2.
__device__ __forceinline__ void remap_xcd(unsigned& bx, unsigned& by) {
  const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
  const unsigned orig = blockIdx.y * gx + blockIdx.x;
  const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
  const unsigned wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
  bx = wg % gx; by = wg / gx;
}

// ---------------------------------------------------------------------------------------------------------
// K1: columns forward.  grid (N2 / C, nb); LDS C * (N1 + 1) complex.
// ---------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(FL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void cols_fwd_kernel(Geom g, Jobs J, cd* __restrict__ work, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  cd* lds = reinterpret_cast<cd*>(smem_raw);
  unsigned bx, by;
  remap_xcd(bx, by);
  const int e = (int)by + J.e0;
  const Ctx ctx = job_ctx<MODE>(J, e);
  const long long L = ctx.L;
  const unsigned N1 = (unsigned)g.n1, N2 = 1u << g.log2n2;
  const long long M = g.m;
  const unsigned n2_0 = bx * C;
  const unsigned lc = 31u - (unsigned)__builtin_clz((unsigned)C), cm = (unsigned)C - 1u;   // C is a power of two (make_plan)
  const unsigned stride = (unsigned)g.rad * ((1u << g.log2q) + 1u);    // rad blocks of Q + 1
  const int tid = threadIdx.x;
  cd* twl = lds + (size_t)C * stride;                                  // two-level twiddle table of the sub-FFT (ira_fft_lds.h)
  const cd twv = ira::tw_split_fetch<double>(g.t1, N1 >> 1, tid);      // load issued first, LDS write after the tile's loads
  // A thread's elements are n_k = n_0 + k*dn (same column, rows FL_THREADS/C apart).  The chirp exp(-i pi n^2/L) along
  // them obeys  w_{k+1} = w_k d_k,  d_{k+1} = d_k e2  with  d_k = exp(-i pi (2 n_k dn + dn^2)/L),  e2 = exp(-i pi 2 dn^2/L):
  // exactly reduced start values per thread instead of one sincospi per element; the Hann window is a plain rotation.
  // (<= N1*C/FL_THREADS = 8 steps, so the recurrences add a few 1e-16.)
  {
    const unsigned c = tid & cm, n1_0 = tid >> lc;
    const unsigned dn = (FL_THREADS >> lc) * N2;                          // all indices < M < 2^31
    const unsigned n0 = n1_0 * N2 + n2_0 + c;
    const unsigned total = N1 * (unsigned)C;
    const unsigned cnt_mine = total > (unsigned)tid ? (total - (unsigned)tid + FL_THREADS - 1) / FL_THREADS : 0u;   // my elements
    // Whole tiles (every M but the very smallest): each thread owns total / FL_THREADS elements, and a count the compiler
    // can see is wave-uniform turns the element loop's `j < cnt` tests into scalar branches -- as per-lane tests they were
    // ~300 v_cndmask per tile around the recurrences' state.  Likewise the job's two yes/no properties (Hann window, second
    // signal): as run-time values every use is a select on a 64-bit double, as constants the unused half of the element
    // loop is not there.  KNOWN bit 0: Hann windows of more than one sample; bit 1: two signals; bit 2: ask at run time.
    // Several copies of the phase, one executed.
    auto load_phase = [&](auto known, const unsigned cnt) __attribute__((always_inline)) {
      constexpr int KNOWN = decltype(known)::value;
      // The first batch of loads goes out BEFORE the trigonometric set-up below (~400 instructions that need no memory).
      constexpr int UI = (MODE == IN_SIGNAL) ? 2 * FL_UI : FL_UI;             // a signal fetch is two floats: all eight at once
      RawL raw[UI];
      if (cnt > 0) {
  #pragma unroll
        for (int u = 0; u < UI; ++u) {
          const unsigned j = (unsigned)u < cnt ? (unsigned)u : cnt - 1;       // clamp: unconditional loads
          raw[u] = fetch_input<MODE>(J, ctx, n0 + j * dn);
        }
      }
      cd w = {1.0, 0.0}, d = {1.0, 0.0}, e2 = {1.0, 0.0};
      // Hann windows of the (up to) two signals: window length and sample count may differ from the transform length
      long long lw1 = L, lw2 = L;
      if (MODE == IN_SIGNAL) {
        lw1 = ctx.lw1;
        lw2 = ctx.lw2;
      }
      const bool second = (KNOWN & 4) ? (MODE == IN_SIGNAL && ctx.o2 >= 0) : (KNOWN & 2) != 0;   // a second signal (and window)
      const bool hann = (KNOWN & 4) ? (MODE == IN_SIGNAL && J.use_hann) : (KNOWN & 1) != 0;
      const bool win1 = (KNOWN & 4) ? lw1 > 1 : true, win2 = (KNOWN & 4) ? lw2 > 1 : true;      // windows longer than one sample
      double hc = 1.0, hs = 0.0, rc = 1.0, rs = 0.0, hc2 = 1.0, hs2 = 0.0, rc2 = 1.0, rs2 = 0.0;
      if (MODE != IN_FILTER) {
        const ChirpScale cs = chirp_scale(L);
        const double n0d = (double)n0, dnd = (double)dn;
        w = unit_q(n0d * n0d, cs);
        d = unit_q(2.0 * n0d * dnd + dnd * dnd, cs);
        const double st = (double)ctx.st;
        const double inv1 = win1 ? 1.0 / (double)(lw1 - 1) : 0.0, inv2 = win2 ? 1.0 / (double)(lw2 - 1) : 0.0;
        // The three values every thread of the job shares -- e2 and the two window rotation steps -- in ONE sincospi: lanes
        // 0 / 1 / 2 of each wave evaluate one of them each, everybody reads the results from those lanes.
        {
          const int lane = tid & 63;
          const double a = lane == 0 ? chirp_angle(2.0 * dnd * dnd, cs) : (2.0 * st * dnd) * (lane == 1 ? inv1 : inv2);
          double s3, c3;
          sincospi(a, &s3, &c3);
          e2 = {lane_value(c3, 0), -lane_value(s3, 0)};
          if (hann && win1) { rs = lane_value(s3, 1); rc = lane_value(c3, 1); }
          if (hann && second && win2) { rs2 = lane_value(s3, 2); rc2 = lane_value(c3, 2); }
        }
        if (hann) {
          // window index of transform index n: n (plain), or 2n / 2n+1 for the even / odd samples of an interleaved job
          const double i1 = st * n0d, i2 = st * n0d + (st - 1.0);
          if (win1) sincospi((2.0 * i1 + 1.0 - (double)lw1) * inv1, &hs, &hc);
          if (second && win2) sincospi((2.0 * i2 + 1.0 - (double)lw2) * inv2, &hs2, &hc2);
        }
      }
      // (nothing of the value phase -- not even the float -> double conversions of the loaded samples, which the optimiser
      // otherwise hoists to right behind the loads -- may come before this line: the first use of a load is where the wave
      // starts to wait for memory; the empty asm pins the loaded values here)
      if (MODE == IN_SIGNAL) {
  #pragma unroll
        for (int u = 0; u < UI; ++u) asm volatile("" : "+v"(raw[u].fa), "+v"(raw[u].fb));
      }
      __builtin_amdgcn_sched_barrier(0);
      for (unsigned jb = 0; jb < cnt; jb += UI) {
        if (jb > 0) {
  #pragma unroll
          for (int u = 0; u < UI; ++u) {
            const unsigned j = jb + u < cnt ? jb + u : cnt - 1;
            raw[u] = fetch_input<MODE>(J, ctx, n0 + j * dn);
          }
        }
  #pragma unroll
        for (int u = 0; u < UI; ++u) {
          const unsigned j = jb + u;
          if (j < cnt) {
            const unsigned n1 = (tid + FL_THREADS * j) >> lc;
            const unsigned n = n0 + j * dn;
            const double h = win1 ? 0.5 + 0.5 * hc : 1.0;
            const double h2 = win2 ? 0.5 + 0.5 * hc2 : 1.0;
            lds[c * stride + col_slot(g, n1)] = (IRA_ABL(g.ablate & 1)) ? cd{(double)n, 1.0} : value_input<MODE>(J, ctx, n, (unsigned)M, w, h, h2, raw[u], hann, second);
            w = ira::cmul(w, d);
            d = ira::cmul(d, e2);
            if (hann) {
              const double nc = hc * rc - hs * rs;
              hs = hs * rc + hc * rs;
              hc = nc;
            }
            if (hann && second) {
              const double nc2 = hc2 * rc2 - hs2 * rs2;
              hs2 = hs2 * rc2 + hc2 * rs2;
              hc2 = nc2;
            }
          }
        }
      }
    };
    using std::integral_constant;
    const unsigned cnt_all = total / FL_THREADS;
    if (total % FL_THREADS != 0) {
      load_phase(integral_constant<int, 4>{}, cnt_mine);
    } else if (MODE != IN_SIGNAL) {
      load_phase(integral_constant<int, 4>{}, cnt_all);
    } else {
      const bool two = ctx.o2 >= 0;
      if (!J.use_hann) {
        if (two) load_phase(integral_constant<int, 2>{}, cnt_all); else load_phase(integral_constant<int, 0>{}, cnt_all);
      } else if (ctx.lw1 > 1 && (!two || ctx.lw2 > 1)) {
        if (two) load_phase(integral_constant<int, 3>{}, cnt_all); else load_phase(integral_constant<int, 1>{}, cnt_all);
      } else {
        load_phase(integral_constant<int, 4>{}, cnt_all);
      }
    }
  }
  ira::tw_split_put(twl, twv, tid);
  __syncthreads();
  if (!(IRA_ABL(g.ablate & 2))) {
    if (g.rad == 3) radix3_stage<false>(g, lds, twl, C, stride, tid);
    ira::lds_fft_dif<double, FL_LR, true>(lds, g.log2q, twl, (unsigned)g.rad, tid, FL_THREADS, C * g.rad, (1u << g.log2q) + 1u);
  }
  cd* w = work + (long long)e * M;
  const unsigned total_o = N1 * (unsigned)C;
  for (unsigned base = 0; base < total_o; base += FL_THREADS * FL_UI) {
    cd th[FL_UI], tl[FL_UI];
#pragma unroll
    for (int u = 0; u < FL_UI; ++u) {
      unsigned i = base + tid + FL_THREADS * u;
      i = i < total_o ? i : total_o - 1;
      const unsigned p = (n2_0 + (i & cm)) * row_k1(g, i >> lc);
      th[u] = g.t1[p >> g.log2n2];
      tl[u] = g.tf[p & (N2 - 1u)];
    }
#pragma unroll
    for (int u = 0; u < FL_UI; ++u) {
      const unsigned i = base + tid + FL_THREADS * u;
      if (i >= total_o) continue;
      const unsigned c = i & cm, r = i >> lc;
      const cd v = ira::cmul(lds[c * stride + col_slot(g, r)], ira::cmul(th[u], tl[u]));
      if ((IRA_ABL(g.ablate & 4)) && v.re != 12345.678) continue;
      w[(long long)r * N2 + n2_0 + c] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// K2: rows.  grid (N1 / R, nb); LDS R * N2 complex.
// ---------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(FL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void rows_kernel(Geom g, Jobs J, cd* __restrict__ work, int R) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  cd* lds = reinterpret_cast<cd*>(smem_raw);
  unsigned bx, by;
  remap_xcd(bx, by);
  const int e = (int)by + J.e0;
  const unsigned N2 = 1u << g.log2n2;
  const long long M = g.m;
  const unsigned r0 = bx * R;
  const int tid = threadIdx.x;
  const int filt = MODE == ROW_CONV ? ira::uniform(J.bidx[e]) : 0;            // before the tile loop (see Ctx)
  cd* w = work + (long long)e * M + (long long)r0 * N2;
  const unsigned total = N2 * (unsigned)R;
  cd* twl = lds + total;                                               // two-level twiddle table of the sub-FFT
  const cd twv = ira::tw_split_fetch<double>(g.t2, N2 >> 1, tid);
  for (unsigned base = 0; base < total; base += FL_THREADS * FL_U) {
    // All loads of the batch in flight together.  Index clamped, and the LDS store NOT guarded (a lane past the end stores
    // the last element's own value onto itself): behind an `if (i < total)` the compiler sinks each load into its store's
    // branch and the eight loads become eight serial round trips (seen in the ISA: load, s_waitcnt vmcnt(0), ds_write, x8).
    cd raw[FL_U];
#pragma unroll
    for (int u = 0; u < FL_U; ++u) {
      const unsigned i = base + tid + FL_THREADS * u;
      raw[u] = w[i < total ? i : total - 1];
    }
#pragma unroll
    for (int u = 0; u < FL_U; ++u) {
      const unsigned i = base + tid + FL_THREADS * u;
      lds[i < total ? i : total - 1] = raw[u];
    }
  }
  ira::tw_split_put(twl, twv, tid);
  __syncthreads();
  if (!(IRA_ABL(g.ablate & 8))) ira::lds_fft_dif<double, FL_LR, true>(lds, g.log2n2, twl, 1u, tid, FL_THREADS, R, N2);
  if (MODE == ROW_CONV) {
    const cd* b = J.bfilt + (long long)filt * M + (long long)r0 * N2;
    if (!(IRA_ABL(g.ablate & 16)))
      for (unsigned base = 0; base < total; base += FL_THREADS * FL_U) {
        cd fb[FL_U];
#pragma unroll
        for (int u = 0; u < FL_U; ++u) {
          unsigned i = base + tid + FL_THREADS * u;
          fb[u] = b[i < total ? i : total - 1];
        }
#pragma unroll
        for (int u = 0; u < FL_U; ++u) {
          const unsigned i = base + tid + FL_THREADS * u;
          if (i < total) lds[i] = ira::cmul(lds[i], fb[u]);
        }
      }
    __syncthreads();
    if (!(IRA_ABL(g.ablate & 8))) ira::lds_fft_dit<double, FL_LR, true>(lds, g.log2n2, twl, 1u, true, tid, FL_THREADS, R, N2);
    for (unsigned base = 0; base < total; base += FL_THREADS * FL_UI) {
      cd th[FL_UI], tl[FL_UI];
#pragma unroll
      for (int u = 0; u < FL_UI; ++u) {
        unsigned i = base + tid + FL_THREADS * u;
        i = i < total ? i : total - 1;
        const unsigned p = (i & (N2 - 1)) * row_k1(g, r0 + (i >> g.log2n2));
        th[u] = g.t1[p >> g.log2n2];
        tl[u] = g.tf[p & (N2 - 1u)];
      }
#pragma unroll
      for (int u = 0; u < FL_UI; ++u) {
        const unsigned i = base + tid + FL_THREADS * u;
        if (i >= total) continue;
        cd t = ira::cmul(th[u], tl[u]);
        t.im = -t.im;
        const cd v = ira::cmul(lds[i], t);
        if ((IRA_ABL(g.ablate & 128)) && v.re != 12345.678) continue;
        w[i] = v;
      }
    }
  } else {
    for (unsigned i = tid; i < total; i += FL_THREADS) w[i] = lds[i];
  }
}

// ---------------------------------------------------------------------------------------------------------
// K3: columns inverse + epilogue.  grid (N2 / C, nb).
// ---------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(FL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void cols_inv_kernel(Geom g, Jobs J, const cd* __restrict__ work, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  cd* lds = reinterpret_cast<cd*>(smem_raw);
  unsigned bx, by;
  remap_xcd(bx, by);
  const int e = (int)by + J.e0;
  // the job's length and output offsets, once (scalar registers; see Ctx)
  long long L, out1, out2 = -1;
  bool paired = false;
  {
    const int l = J.L[e];
    if (MODE == OUT_SPECTRUM) {
      const long long x2 = J.x2off ? (long long)J.x2off[e] : -1ll, zo = J.x2off ? (long long)J.zpair_off[e] : 0ll;
      const long long so = J.spec_out_off[e];
      paired = ira::uniform(x2) >= 0;
      out1 = paired ? ira::uniform(zo) : ira::uniform(so);
    } else {
      const long long y1 = J.y1_off[e], y2 = J.y2_off[e];
      out1 = ira::uniform(y1);
      out2 = ira::uniform(y2);
    }
    L = ira::uniform(l);
  }
  const unsigned N1 = (unsigned)g.n1, N2 = 1u << g.log2n2;
  const long long M = g.m;
  const unsigned n2_0 = bx * C;
  const unsigned lc = 31u - (unsigned)__builtin_clz((unsigned)C), cm = (unsigned)C - 1u;   // C is a power of two (make_plan)
  const unsigned stride = (unsigned)g.rad * ((1u << g.log2q) + 1u);    // rad blocks of Q + 1
  const int tid = threadIdx.x;
  // outputs needed: n <= L/2 (spectrum) or n < L (bands); rows beyond that are computed but not stored
  const long long n_need = (MODE == OUT_SPECTRUM && !paired) ? L / 2 + 1 : L;
  const cd* w = work + (long long)e * M;
  const unsigned total = N1 * (unsigned)C;
  cd* twl = lds + (size_t)C * stride;                                  // two-level twiddle table of the sub-FFT
  const cd twv = ira::tw_split_fetch<double>(g.t1, N1 >> 1, tid);
  for (unsigned base = 0; base < total; base += FL_THREADS * FL_U) {
    cd raw[FL_U];
#pragma unroll
    for (int u = 0; u < FL_U; ++u) {
      unsigned i = base + tid + FL_THREADS * u;
      i = i < total ? i : total - 1;
      raw[u] = (IRA_ABL(g.ablate & 256)) ? cd{(double)i, 1.0} : w[(long long)(i >> lc) * N2 + n2_0 + (i & cm)];
    }
#pragma unroll
    for (int u = 0; u < FL_U; ++u) {
      const unsigned i = base + tid + FL_THREADS * u;
      if (i < total) lds[(i & cm) * stride + col_slot(g, i >> lc)] = raw[u];
    }
  }
  ira::tw_split_put(twl, twv, tid);
  __syncthreads();
  if (!(IRA_ABL(g.ablate & 32))) {
    ira::lds_fft_dit<double, FL_LR, true>(lds, g.log2q, twl, (unsigned)g.rad, true, tid, FL_THREADS, C * g.rad, (1u << g.log2q) + 1u);
    if (g.rad == 3) radix3_stage<true>(g, lds, twl, C, stride, tid);
  }
  const double inv_m = 1.0 / (double)M;
  // The output chirp exp(-i pi n^2 / L) along a thread's elements n_j = n_0 + j dn by the same recurrence as in K1
  // (w_{j+1} = w_j d_j, d_{j+1} = d_j e2: three exactly reduced sincospi per thread instead of a 64-bit modulo and a
  // sincospi per element, which were most of this kernel's instructions; <= 8 steps, a few 1e-16).
  const long long dn = (long long)(FL_THREADS >> lc) * N2;
  const long long n0 = (long long)((unsigned)tid >> lc) * N2 + n2_0 + ((unsigned)tid & cm);
  const ChirpScale cs = chirp_scale(L);
  const double n0d = (double)n0, dnd = (double)dn;
  cd cw = unit_q(n0d * n0d, cs), cdl = unit_q(2.0 * n0d * dnd + dnd * dnd, cs);
  const cd ce2 = unit_q(2.0 * dnd * dnd, cs);
  for (unsigned i = tid; i < N1 * (unsigned)C; i += FL_THREADS) {
    const unsigned c = i & cm, n1 = i >> lc;
    const long long n = (long long)n1 * N2 + n2_0 + c;
    const cd wn = cw;
    cw = ira::cmul(cw, cdl);
    cdl = ira::cmul(cdl, ce2);
    if (n >= n_need) continue;
    cd v = lds[c * stride + col_slot(g, n1)];
    if (!(IRA_ABL(g.ablate & 64))) v = ira::cmul(v, wn);
    if (MODE == OUT_SPECTRUM) {
      v.re *= inv_m; v.im *= inv_m;
      if (paired) {
        J.zpair[out1 + n] = v;                   // split into the two half spectra by pair_split_kernel
      } else {
        if (n == 0 || (2 * n == L)) v.im = 0.0;  // DC / Nyquist of a real signal
        J.spec_out[out1 + n] = v;
      }
    } else {
      const double sc = inv_m / (double)L;
      // y1 + i y2 = conj(v) / L
      J.y[out1 + n] = (float)(v.re * sc);
      if (out2 >= 0) J.y[out2 + n] = (float)(-v.im * sc);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// K3 with the register hop taken out of its memory phase (round 5): persistent workgroups, two LDS buffers, tiles fetched by
// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no data VGPRs, counted by vmcnt).  A workgroup walks
// its share of the launch's (job, column tile) pairs; while tile t is transformed and written out, tile t + 1 lands in the
// other buffer.  An LDS-DMA instruction writes its 64 x 16 bytes CONTIGUOUSLY, so the tile keeps its arrival order
// [row][column] (rows of C adjacent columns = the contiguous pieces of the work array) and the transform runs on that
// layout (lds_fft_dit_rc); no column padding.  Barriers inside the walk are lds_barrier() -- __syncthreads() would drain the
// DMA.  Everything the compiler cannot see (the DMA and its waits) is inline assembly with a memory clobber.
//   per tile:  s_waitcnt vmcnt(0) + barrier  (tile t landed, stores of tile t - 1 retired, every wave done with the other buffer)
//              -> issue tile t + 1 -> inverse column transforms of tile t -> output chirp, epilogue, stores
// grid: a multiple of 8 workgroups (XCD x owns a contiguous range of tiles, its workgroups walk it side by side).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_address(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

template <bool INVERSE>
__device__ __forceinline__ void radix3_stage_rc(const Geom& g, cd* buf, const cd* twl, int lc, int tid) {
  const unsigned Q = 1u << g.log2q, half = (unsigned)g.n1 >> 1, cm = (1u << lc) - 1u, bs = Q << lc;
  for (unsigned idx = tid; idx < (Q << lc); idx += FL_THREADS) {
    const unsigned c = idx & cm, j = idx >> lc;
    cd* p = buf + (j << lc) + c;
    cd a0 = p[0], a1 = p[bs], a2 = p[2 * bs];
    cd w1 = ira::tw_get<double, true>(twl, j);
    cd w2 = ira::tw_lookup<double, true>(twl, 2u * j, half);
    if (INVERSE) {
      w1.im = -w1.im; w2.im = -w2.im;
      a1 = ira::cmul(a1, w1);
      a2 = ira::cmul(a2, w2);
      a0.im = -a0.im; a1.im = -a1.im; a2.im = -a2.im;
      bfly3(a0, a1, a2);
      a0.im = -a0.im; a1.im = -a1.im; a2.im = -a2.im;
    } else {
      bfly3(a0, a1, a2);
      if (j != 0) {
        a1 = ira::cmul(a1, w1);
        a2 = ira::cmul(a2, w2);
      }
    }
    p[0] = a0; p[bs] = a1; p[2 * bs] = a2;
  }
  ira::lds_barrier();
}

struct K3Raw { int l; long long a, b, c; };          // a job's values as loaded (vector registers), and once uniform
template <int MODE>
__device__ __forceinline__ K3Raw k3_fetch(const Jobs& J, int e) {
  K3Raw r;
  r.l = J.L[e];
  if (MODE == OUT_SPECTRUM) {
    r.a = J.x2off ? (long long)J.x2off[e] : -1ll;
    r.b = J.x2off ? (long long)J.zpair_off[e] : 0ll;
    r.c = J.spec_out_off[e];
  } else {
    r.a = J.y1_off[e];
    r.b = J.y2_off[e];
    r.c = 0;
  }
  return r;
}

template <int MODE>
__global__ __launch_bounds__(FL_THREADS) void cols_inv_glds_kernel(Geom g, Jobs J, const cd* __restrict__ work, int lc,
                                                                   unsigned tiles_per_job, unsigned total_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  cd* lds = reinterpret_cast<cd*>(smem_raw);
  const unsigned N1 = (unsigned)g.n1, N2 = 1u << g.log2n2;
  const long long M = g.m;
  const unsigned cm = (1u << lc) - 1u;
  const unsigned tile = N1 << lc;                                       // complex values per tile, a multiple of FL_THREADS
  const int tid = threadIdx.x;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6), lane = (unsigned)tid & 63u;
  cd* twl = lds + 2 * (size_t)tile;                                     // both DMA buffers below it (low LDS addresses)
  const cd twv = ira::tw_split_fetch<double>(g.t1, N1 >> 1, tid);
  // my tiles: XCD x = blockIdx % 8 owns tiles [t0, t0 + cnt), its workgroups take them round robin
  const unsigned nx = gridDim.x >> 3, xcd = blockIdx.x & 7u, lw = blockIdx.x >> 3;
  const unsigned tq = total_tiles >> 3, tr = total_tiles & 7u;
  const unsigned t0 = xcd < tr ? xcd * (tq + 1u) : tr * (tq + 1u) + (xcd - tr) * tq, cnt = xcd < tr ? tq + 1u : tq;
  const unsigned chunks = (tile >> 6) / 4u;                             // DMA instructions per wave and tile
  const unsigned lds0 = lds_address(lds);
  auto issue = [&](unsigned t, unsigned which) __attribute__((always_inline)) {
    const unsigned e = t / tiles_per_job, bx = t - e * tiles_per_job;
    const cd* src = work + (long long)e * M + (bx << lc) + (long long)((wave * 64u + lane) >> lc) * N2 + (lane & cm);
    const long long step = (long long)(256u >> lc) * N2;                // rows between a wave's consecutive pieces
    unsigned dst = lds0 + which * tile * 16u + wave * 1024u;
    if (IRA_ABL(g.ablate & 256)) return;
    for (unsigned u = 0; u < chunks; ++u) {
      glds16(src, dst);
      src += step;
      dst += 4096u;
    }
  };
  ira::tw_split_put(twl, twv, tid);
  K3Raw raw{};
  if (lw < cnt) {
    issue(t0 + lw, 0u);
    raw = k3_fetch<MODE>(J, (int)((t0 + lw) / tiles_per_job) + J.e0);
  }
  const double inv_m = 1.0 / (double)M;
  unsigned which = 0;
  for (unsigned s = lw; s < cnt; s += nx, which ^= 1u) {
    const unsigned t = t0 + s;
    const unsigned e_rel = t / tiles_per_job, bx = t - e_rel * tiles_per_job;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");        // tile t has landed; the other buffer is free
    asm volatile("" : "+v"(raw.l), "+v"(raw.a), "+v"(raw.b), "+v"(raw.c));      // first use of the job's values: after the wait
    const long long L = ira::uniform(raw.l);
    long long out1, out2 = -1;
    bool paired = false;
    if (MODE == OUT_SPECTRUM) {
      paired = ira::uniform(raw.a) >= 0;
      out1 = paired ? ira::uniform(raw.b) : ira::uniform(raw.c);
    } else {
      out1 = ira::uniform(raw.a);
      out2 = ira::uniform(raw.b);
    }
    if (s + nx < cnt) {
      issue(t + nx, which ^ 1u);
      raw = k3_fetch<MODE>(J, (int)((t + nx) / tiles_per_job) + J.e0);
    }
    cd* buf = lds + (size_t)which * tile;
    if (!(IRA_ABL(g.ablate & 32))) {
      ira::lds_fft_dit_rc<double, FL_LR, true>(buf, g.log2q, twl, (unsigned)g.rad, true, tid, FL_THREADS, g.rad, lc);
      if (g.rad == 3) radix3_stage_rc<true>(g, buf, twl, lc, tid);
    }
    const unsigned n2_0 = bx << lc;
    const long long n_need = (MODE == OUT_SPECTRUM && !paired) ? L / 2 + 1 : L;
    const long long dn = (long long)(FL_THREADS >> lc) * N2;
    const long long n0 = (long long)((unsigned)tid >> lc) * N2 + n2_0 + ((unsigned)tid & cm);
    const ChirpScale cs = chirp_scale(L);
    const double n0d = (double)n0, dnd = (double)dn;
    cd cw = unit_q(n0d * n0d, cs), cdl = unit_q(2.0 * n0d * dnd + dnd * dnd, cs);
    const cd ce2 = unit_q(2.0 * dnd * dnd, cs);
    for (unsigned i = tid; i < tile; i += FL_THREADS) {
      const long long n = (long long)(i >> lc) * N2 + n2_0 + (i & cm);
      const cd wn = cw;
      cw = ira::cmul(cw, cdl);
      cdl = ira::cmul(cdl, ce2);
      if (n >= n_need) continue;
      cd v = buf[i];
      if (!(IRA_ABL(g.ablate & 64))) v = ira::cmul(v, wn);
      if ((IRA_ABL(g.ablate & 512)) && v.re != 12345.678) continue;
      if (MODE == OUT_SPECTRUM) {
        v.re *= inv_m; v.im *= inv_m;
        if (paired) {
          J.zpair[out1 + n] = v;
        } else {
          if (n == 0 || (2 * n == L)) v.im = 0.0;
          J.spec_out[out1 + n] = v;
        }
      } else {
        const double sc = inv_m / (double)L;
        J.y[out1 + n] = (float)(v.re * sc);
        if (out2 >= 0) J.y[out2 + n] = (float)(-v.im * sc);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Two real signals per transform: with Z = DFT(x1 + i x2),
//   X1[k] = (Z[k] + conj Z[L-k]) / 2,   X2[k] = (Z[k] - conj Z[L-k]) / (2i),   k = 0 .. L/2.
// k = 0 and k = L/2 pair a bin with itself, so their imaginary parts come out exactly zero like numpy's rfft.
__global__ __launch_bounds__(256) void pair_split_kernel(Jobs J) {
  const int e = blockIdx.y;
  if (J.x2off[e] < 0 || (J.interleave && J.interleave[e])) return;
  const long long L = J.L[e];
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > L / 2) return;
  const cd* z = J.zpair + J.zpair_off[e];
  const long long o1 = J.spec_out_off[e], o2 = J.spec_out_off2[e];      // before the stores (see Ctx)
  const cd zk = z[k], zl = z[k == 0 ? 0 : L - k];
  J.spec_out[o1 + k] = {0.5 * (zk.re + zl.re), 0.5 * (zk.im - zl.im)};
  J.spec_out[o2 + k] = {0.5 * (zk.im + zl.im), 0.5 * (zl.re - zk.re)};
}

// Interleaved jobs: z[m] = x[2m] + i x[2m+1], Z = DFT_L(z); the real signal's spectrum of length 2L is
//   X[k] = E[k] + W_2L^k O[k],  E = (Z[k] + conj Z[L-k]) / 2,  O = (Z[k] - conj Z[L-k]) / (2i),  k = 0 .. L  (Z index mod L)
__global__ __launch_bounds__(256) void half_split_kernel(Jobs J) {
  const int e = blockIdx.y;
  if (!J.interleave[e]) return;
  const long long L = J.L[e];
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > L) return;
  const cd* z = J.zpair + J.zpair_off[e];
  const cd zk = z[k == L ? 0 : k], zl = z[(k == 0 || k == L) ? 0 : L - k];
  const cd ev = {0.5 * (zk.re + zl.re), 0.5 * (zk.im - zl.im)};
  const cd od = {0.5 * (zk.im + zl.im), 0.5 * (zl.re - zk.re)};
  double sn, cs;
  sincospi(-(double)k / (double)L, &sn, &cs);                 // W_2L^k = exp(-i pi k / L)
  cd x = {ev.re + (cs * od.re - sn * od.im), ev.im + (cs * od.im + sn * od.re)};
  if (k == 0 || k == L) x.im = 0.0;                            // DC / Nyquist of a real signal
  J.spec_out[J.spec_out_off[e] + k] = x;
}

// M = N1 x N2.  The column passes (K1, K3) touch C adjacent columns of every row, i.e. C*16-byte pieces at a stride
// of N2*16 bytes, and C is what fits in LDS: a SHORT column (small N1) buys wide pieces.  IRA_FFT_SPLIT overrides
// log2(N1) for tuning (powers of two); split_of() is the single source of truth the host sizes its tables from
// (ira_fft_split).  Sizes: 2^k (16 <= M <= 2^22) and 3 * 2^k (96 <= M <= 3 * 2^20).
struct Split { int rad, log2q, log2n2; };

bool split_of(long long m, Split* out) {
  if (m < 16 || m > (1ll << 22)) return false;
  int rad = 1;
  if (m % 3 == 0) { rad = 3; m /= 3; }
  if ((m & (m - 1)) != 0) return false;
  int log2p = 0;
  while ((1ll << log2p) < m) ++log2p;
  int lq;                                       // log2 of the power-of-two part of N1
  if (rad == 1) {
    const int forced = ira_tune_int("IRA_FFT_SPLIT", 0);
    lq = (log2p + 1) / 2;
    if (forced > 0) lq = forced;
    if (lq < 2) lq = 2;
    if (lq > log2p - 2) lq = log2p - 2;
    if (log2p - lq > 13) lq = log2p - 13;      // one row (N2 complex f64) must fit in LDS: N2 <= 8192
  } else {
    if (log2p < 5) return false;                // M >= 96
    lq = (log2p - 1) / 2;                       // N1 = 3 * 2^lq just below N2: 768 x 1024 at M = 3 * 2^18
    { const int forced = ira_tune_int("IRA_FFT_SPLIT3", 0); if (forced >= 2 && forced <= log2p - 2) lq = forced; }
    if (log2p - lq > 13) lq = log2p - 13;
  }
  out->rad = rad; out->log2q = lq; out->log2n2 = log2p - lq;
  return true;
}

struct Plan {
  Geom g;
  int C, R;
  size_t lds_cols, lds_rows;
  int C3; size_t lds_cols3;          // the inverse column pass may take its own tile width (tuning: IRA_FFT_C3)
  int glds_lc; size_t lds_glds;      // K3 by LDS-DMA (cols_inv_glds_kernel): log2 of its tile width, or -1 where it does not apply
};

int32_t make_plan(int32_t m, const void* t1, const void* t2, const void* tf, Plan* p) {
  Split sp;
  if (!split_of(m, &sp)) return IRA_E_SIZE;
  p->g.m = m;
  p->g.rad = sp.rad;
  p->g.log2q = sp.log2q;
  p->g.n1 = sp.rad << sp.log2q;
  p->g.log2n2 = sp.log2n2;
  p->g.t1 = static_cast<const cd*>(t1);
  p->g.t2 = static_cast<const cd*>(t2);
  p->g.tf = static_cast<const cd*>(tf);
  p->g.ablate = ira_tune_int("IRA_FFT_ABLATE", 0);
  const int N1 = p->g.n1, N2 = 1 << p->g.log2n2;
  const size_t col = (size_t)sp.rad * ((1u << sp.log2q) + 1);            // LDS entries of one column (blocks of Q + 1)
  // ~32 KB of LDS per workgroup (C = R = 2 at N1 = N2 = 1024): measured fastest on MI355X -- 4 workgroups per CU
  // hide each other's barriers (rfft_any, 64 x 2^20: C/R = 4/4 2.75 ms, 2/2 2.39 ms, 1/1 2.92 ms)
  int C = 8;
  while (C > 1 && (size_t)C * col * sizeof(cd) > 33 * 1024) C >>= 1;
  if (C > N2) C = N2;
  int R = 1;
  while (R * 2 * N2 * (int)sizeof(cd) <= 32 * 1024 && N1 % (R * 2) == 0 && R < 16) R <<= 1;
  { const int v = ira_tune_int("IRA_FFT_C", 0); if (v >= 1 && v <= N2 && v <= 64 && (v & (v - 1)) == 0) C = v; }   // tuning (power of two: K1 relies on FL_THREADS % C == 0)
  { const int v = ira_tune_int("IRA_FFT_R", 0); if (v >= 1 && v <= N1 && N1 % v == 0) R = v; }
  p->C = C;
  p->R = R;
  p->lds_cols = ((size_t)C * col + ira::TW_SPLIT_ENTRIES) * sizeof(cd);          // tile + two-level twiddle table
  p->C3 = C;
  { const int v = ira_tune_int("IRA_FFT_C3", 0); if (v >= 1 && v <= N2 && v <= 64 && (v & (v - 1)) == 0) p->C3 = v; }
  p->lds_cols3 = ((size_t)p->C3 * col + ira::TW_SPLIT_ENTRIES) * sizeof(cd);
  p->lds_rows = ((size_t)R * N2 + ira::TW_SPLIT_ENTRIES) * sizeof(cd);
  // K3 by LDS-DMA: two unpadded buffers of N1 x C values below 64 KB of LDS (the DMA's destination register), whole
  // 1 KB pieces per wave (tile a multiple of FL_THREADS values), a radix the in-LDS transform can split (log2q >= 2)
  p->glds_lc = -1; p->lds_glds = 0;
  {
    int c = C;
    { const int v = ira_tune_int("IRA_FFT_CG", 0); if (v >= 1 && v <= N2 && v <= 64 && (v & (v - 1)) == 0) c = v; }
    while (c > 1 && 2 * (size_t)c * N1 * sizeof(cd) > 64 * 1024) c >>= 1;
    const size_t tile = (size_t)c * N1;
    if (ira_tune_int("IRA_FFT_GLDS", 0) != 0 && 2 * tile * sizeof(cd) <= 64 * 1024 && tile % FL_THREADS == 0 && N2 % c == 0 &&
        sp.log2q >= 2) {
      int lc = 0;
      while ((1 << lc) < c) ++lc;
      p->glds_lc = lc;
      p->lds_glds = (2 * tile + ira::TW_SPLIT_ENTRIES) * sizeof(cd);
    }
  }
  return IRA_OK;
}

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

#define IRA_TRY_HIP(expr)                          \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) return ira_hip_status(_e); \
  } while (0)

template <int IN, int OUT>
int32_t run_convolution(const Plan& p, const Jobs& J, cd* work, int nb, hipStream_t st) {
  const int N1 = p.g.n1, N2 = 1 << p.g.log2n2;
  IRA_TRY_HIP(allow_lds(cols_fwd_kernel<IN>, p.lds_cols));
  IRA_TRY_HIP(allow_lds(rows_kernel<ROW_CONV>, p.lds_rows));
  IRA_TRY_HIP(allow_lds(cols_inv_kernel<OUT>, p.lds_cols3));
  int cus = 256, per_cu = 2;
  if (p.glds_lc >= 0) {
    // persistent grid: as many workgroups as are resident at once (no state kept: asked of the runtime on every call)
    IRA_TRY_HIP(allow_lds(cols_inv_glds_kernel<OUT>, p.lds_glds));
    int dev = 0;
    IRA_TRY_HIP(hipGetDevice(&dev));
    IRA_TRY_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    IRA_TRY_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cols_inv_glds_kernel<OUT>, FL_THREADS, p.lds_glds));
    { const int v = ira_tune_int("IRA_FFT_GLDS_WG", 0); if (v >= 1 && v < per_cu) per_cu = v; }
    if (per_cu < 1) per_cu = 1;
  }
  // IRA_FFT_CHUNK (tuning): the three passes over sub-ranges of the jobs, so that a sub-range's work arrays (16 M bytes per
  // job, written by one pass and read by the next) may stay in the 256 MB Infinity Cache.  0 = one launch per pass.
  int chunk = ira_tune_int("IRA_FFT_CHUNK", 0);
  if (chunk <= 0 || chunk > nb) chunk = nb;
  Jobs Jc = J;
  for (int j0 = 0; j0 < nb; j0 += chunk) {
    const int cnt = nb - j0 < chunk ? nb - j0 : chunk;
    Jc.e0 = j0;
    cols_fwd_kernel<IN><<<dim3(N2 / p.C, cnt), FL_THREADS, p.lds_cols, st>>>(p.g, Jc, work, p.C);
    rows_kernel<ROW_CONV><<<dim3(N1 / p.R, cnt), FL_THREADS, p.lds_rows, st>>>(p.g, Jc, work, p.R);
    if (p.glds_lc >= 0) {
      const unsigned tiles_per_job = (unsigned)(N2 >> p.glds_lc), total = tiles_per_job * (unsigned)cnt;
      unsigned grid = (unsigned)(cus * per_cu) & ~7u;
      if (grid > ((total + 7u) & ~7u)) grid = (total + 7u) & ~7u;
      cols_inv_glds_kernel<OUT><<<dim3(grid), FL_THREADS, p.lds_glds, st>>>(p.g, Jc, work, p.glds_lc, tiles_per_job, total);
    } else {
      cols_inv_kernel<OUT><<<dim3(N2 / p.C3, cnt), FL_THREADS, p.lds_cols3, st>>>(p.g, Jc, work, p.C3);
    }
  }
  IRA_RETURN_LAUNCH();
}

}  // namespace

extern "C" int32_t ira_bluestein_filter(const int32_t* L_dev, int32_t nfilt, int32_t m, const void* t1_dev,
                                        const void* t2_dev, const void* tf_dev, double* bfilt_dev, void* stream) {
  IRA_CHECK_PTR(L_dev); IRA_CHECK_PTR(t1_dev); IRA_CHECK_PTR(t2_dev); IRA_CHECK_PTR(tf_dev); IRA_CHECK_PTR(bfilt_dev);
  if (nfilt <= 0) return nfilt == 0 ? IRA_OK : IRA_E_SIZE;
  Plan p;
  int32_t rc = make_plan(m, t1_dev, t2_dev, tf_dev, &p);
  if (rc != IRA_OK) return rc;
  Jobs J{};
  J.L = L_dev;
  hipStream_t st = (hipStream_t)stream;
  const int N1 = p.g.n1, N2 = 1 << p.g.log2n2;
  IRA_TRY_HIP(allow_lds(cols_fwd_kernel<IN_FILTER>, p.lds_cols));
  IRA_TRY_HIP(allow_lds(rows_kernel<ROW_FWD>, p.lds_rows));
  cd* b = reinterpret_cast<cd*>(bfilt_dev);
  cols_fwd_kernel<IN_FILTER><<<dim3(N2 / p.C, nfilt), FL_THREADS, p.lds_cols, st>>>(p.g, J, b, p.C);
  rows_kernel<ROW_FWD><<<dim3(N1 / p.R, nfilt), FL_THREADS, p.lds_rows, st>>>(p.g, J, b, p.R);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_rfft_any(const float* x_dev, const int64_t* xoff_dev, const int32_t* L_dev, int32_t nb,
                                int32_t use_hann, int32_t m, const void* t1_dev, const void* t2_dev,
                                const void* tf_dev, const double* bfilt_dev, const int32_t* bidx_dev,
                                double* work_dev, double* spec_out_dev, const int64_t* spec_off_dev,
                                const int64_t* x2off_dev, const int64_t* spec_off2_dev, double* zpair_dev,
                                const int64_t* zpair_off_dev, int32_t max_len, const int32_t* data_len_dev,
                                const int32_t* win_len_dev, const int32_t* data_len2_dev,
                                const int32_t* win_len2_dev, const int32_t* interleave_dev, int32_t keep_packed,
                                void* stream) {
  IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(xoff_dev); IRA_CHECK_PTR(L_dev); IRA_CHECK_PTR(t1_dev); IRA_CHECK_PTR(t2_dev);
  IRA_CHECK_PTR(tf_dev); IRA_CHECK_PTR(bfilt_dev); IRA_CHECK_PTR(bidx_dev); IRA_CHECK_PTR(work_dev);
  IRA_CHECK_PTR(spec_out_dev); IRA_CHECK_PTR(spec_off_dev);
  if (nb <= 0) return nb == 0 ? IRA_OK : IRA_E_SIZE;
  Plan p;
  int32_t rc = make_plan(m, t1_dev, t2_dev, tf_dev, &p);
  if (rc != IRA_OK) return rc;
  Jobs J{};
  J.L = L_dev; J.x = x_dev; J.xoff = xoff_dev; J.use_hann = use_hann;
  J.bfilt = reinterpret_cast<const cd*>(bfilt_dev); J.bidx = bidx_dev;
  J.spec_out = reinterpret_cast<cd*>(spec_out_dev); J.spec_out_off = spec_off_dev;
  J.data_len = data_len_dev; J.win_len = win_len_dev; J.data_len2 = data_len2_dev; J.win_len2 = win_len2_dev;
  if (interleave_dev != nullptr && x2off_dev == nullptr) return IRA_E_NULL;
  // keep_packed leaves the third pass's output Z = DFT(x1 + i x2) where it wrote it: right for INTERLEAVED elements (the
  // consumer untangles), wrong for a PAIRED one (two signals of two channels), which would silently return two tangled
  // spectra.  Without the interleave table every second signal is a paired one: refuse.  With it, elements
  // that are not interleaved must be single (x2off < 0) -- the caller's contract, stated in ira.h.
  if (keep_packed && x2off_dev != nullptr && interleave_dev == nullptr) return IRA_E_UNSUPPORTED;
  J.interleave = interleave_dev;
  if (x2off_dev != nullptr) {
    if (spec_off2_dev == nullptr || zpair_dev == nullptr || zpair_off_dev == nullptr) return IRA_E_NULL;
    if (max_len <= 0) return IRA_E_SIZE;
    J.x2off = x2off_dev; J.spec_out_off2 = spec_off2_dev;
    J.zpair = reinterpret_cast<cd*>(zpair_dev); J.zpair_off = zpair_off_dev;
  }
  rc = run_convolution<IN_SIGNAL, OUT_SPECTRUM>(p, J, reinterpret_cast<cd*>(work_dev), nb, (hipStream_t)stream);
  if (rc != IRA_OK || x2off_dev == nullptr) return rc;
  // keep_packed calls carry no PAIRED elements (every second signal is the odd half of an interleaved one): nothing to split
  if (keep_packed) return rc;
  pair_split_kernel<<<dim3((max_len / 2 + 1 + 255) / 256, nb), 256, 0, (hipStream_t)stream>>>(J);
  // keep_packed: the half-length transforms Z of interleaved elements stay as the column pass wrote them (zpair); the
  // consumer untangles them on the fly (ira_spectrum_mag_phase with packed_dev) -- no read-modify-write pass over Z
  if (interleave_dev != nullptr && !keep_packed)
    half_split_kernel<<<dim3((max_len + 1 + 255) / 256, nb), 256, 0, (hipStream_t)stream>>>(J);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_band_irfft(const double* spec_dev, const int64_t* spec_off_dev, const int32_t* L_dev,
                                  int32_t nb, const double* band_params_dev, const double* freq_val_dev,
                                  int32_t m, const void* t1_dev, const void* t2_dev, const void* tf_dev,
                                  const double* bfilt_dev, const int32_t* bidx_dev, double* work_dev, float* y_dev,
                                  const int64_t* y1_off_dev, const int64_t* y2_off_dev,
                                  const int64_t* spec_off2_dev, void* stream) {
  IRA_CHECK_PTR(spec_dev); IRA_CHECK_PTR(spec_off_dev); IRA_CHECK_PTR(L_dev); IRA_CHECK_PTR(band_params_dev);
  IRA_CHECK_PTR(freq_val_dev); IRA_CHECK_PTR(t1_dev); IRA_CHECK_PTR(t2_dev); IRA_CHECK_PTR(tf_dev);
  IRA_CHECK_PTR(bfilt_dev); IRA_CHECK_PTR(bidx_dev); IRA_CHECK_PTR(work_dev); IRA_CHECK_PTR(y_dev);
  IRA_CHECK_PTR(y1_off_dev); IRA_CHECK_PTR(y2_off_dev);
  if (nb <= 0) return nb == 0 ? IRA_OK : IRA_E_SIZE;
  Plan p;
  int32_t rc = make_plan(m, t1_dev, t2_dev, tf_dev, &p);
  if (rc != IRA_OK) return rc;
  static_assert(sizeof(BandMask) == 8 * sizeof(double), "band parameter record is 8 doubles");
  Jobs J{};
  J.L = L_dev;
  J.spec = reinterpret_cast<const cd*>(spec_dev); J.spec_off = spec_off_dev; J.spec_off2 = spec_off2_dev;
  J.bands = reinterpret_cast<const BandMask*>(band_params_dev); J.freq_val = freq_val_dev;
  J.bfilt = reinterpret_cast<const cd*>(bfilt_dev); J.bidx = bidx_dev;
  J.y = y_dev; J.y1_off = y1_off_dev; J.y2_off = y2_off_dev;
  return run_convolution<IN_SPECTRUM, OUT_BANDS>(p, J, reinterpret_cast<cd*>(work_dev), nb, (hipStream_t)stream);
}

extern "C" int32_t ira_fft_split(int32_t m, int32_t* n1, int32_t* n2) {
  IRA_CHECK_PTR(n1); IRA_CHECK_PTR(n2);
  Split sp;
  if (!split_of(m, &sp)) return IRA_E_SIZE;
  *n1 = sp.rad << sp.log2q;
  *n2 = 1 << sp.log2n2;
  return IRA_OK;
}
