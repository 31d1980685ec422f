// STFT v2: register-resident FFT, one TEAM of TW waves per frame (TW = 1: n_fft 4096, TW = 2: n_fft 8192).
//
// Packed real FFT: z[n] = xw[2n] + i xw[2n+1], M = n_fft/2 = 2048*TW complex points, 32 per lane.
// M = 16 * 16 * R3 (R3 = 8 or 16), decimation in frequency, natural-order input:
//   step 1  16-point DFTs over n1 (stride M/16) straight from global memory (coalesced), twiddle W_M^(k1*m)
//   step 2  16-point DFTs over n2 after exchange 1 through LDS,                twiddle W_M^(16*k2*n3)
//   step 3  R3-point DFTs over n3 after exchange 2 -> lane holds Z[r + 256*k3], r = k2*16 + k1
//   post    exchange 3 (natural order), every (Z[k], Z[M-k]) pair yields BOTH X[k] and X[M-k]:
//           E = (Zk + conj Zp)/2, P = W_N^k * (-i)(Zk - conj Zp)/2, |X[k]| = |E + P|, |X[M-k]| = |E - P|
// LDS layouts are padded so that every exchange access is bank-conflict free (see the address comments).
// Each team runs FS frames back to back keeping its dB outputs in registers; then the whole workgroup writes its
// NT*FS columns into one [F][TB+1] float tile that ALIASES the exchange buffers and stores TB-float runs of the
// C-contiguous (F, T) matrix (64-byte runs for TB = 16).
#include <cmath>
#include <cstdlib>

#include "ira_fft_reg.h"
#include "ira_log.h"

namespace {

using ira::cplx;
using ira::brev_bits;
using ira::dft_dif;
using ira::powers16;
using ira::LogTabEntry;
using ira::LOGTAB_N;
using ira::build_log_table;
using ira::log2_table;

template <typename T>
__device__ __forceinline__ float power_to_db(T re, T im, T floor_lin, float floor_db, const LogTabEntry* tab);
template <>
__device__ __forceinline__ float power_to_db<float>(float re, float im, float floor_lin, float floor_db,
                                                    const LogTabEntry*) {
  const float p = re * re + im * im;
  if (!(p > floor_lin * floor_lin)) return floor_db;
  return 3.0102999566398120f * __log2f(p);   // 10*log10(p) = 20*log10(|X|)
}
template <>
__device__ __forceinline__ float power_to_db<double>(double re, double im, double floor_lin, float floor_db,
                                                     const LogTabEntry* tab) {
  const double p = fma(re, re, im * im);
  if (!(p > floor_lin * floor_lin)) return floor_db;                        // also catches NaN
  if (!(p < 1.0e300)) return (float)(20.0 * log10(hypot(re, im)));          // overflow / infinity: slow exact path
  return (float)(3.0102999566398120 * log2_table(p, tab));
}

template <typename T, int TW>
struct Cfg {
  static constexpr int TL = 64 * TW;            // lanes per team
  static constexpr int M = 2048 * TW;           // complex points
  static constexpr int N = 2 * M;               // n_fft
  static constexpr int F = M + 1;
  static constexpr int MB = M / 16;             // (n2, n3) pairs = 2 * TL
  static constexpr int R3 = M / 256;            // 8 or 16
  // Exchange-1 row stride: MB + 24 keeps the four k1 rows a half-wave reads on disjoint bank quarters AND puts
  // rows 8..15 behind everything the first half of exchange 2 writes (8*ROW1 >= 128*ROW2), so step 2 can
  // consume E1 half by half and never holds more than 16 complex values.
  static constexpr int PAD1 = 24;
  static constexpr int ROW1 = MB + PAD1;        // exchange-1 row stride (complex)
  static constexpr int ROW2 = R3 + 1;           // exchange-2 row stride (complex), row = k1*16 + k2
  static constexpr int E3 = M + M / 16;         // exchange-3: Z[k] at k + (k >> 4)
  static constexpr int EX = (16 * ROW1 > 256 * ROW2) ? ((16 * ROW1 > E3) ? 16 * ROW1 : E3)
                                                     : ((256 * ROW2 > E3) ? 256 * ROW2 : E3);   // complex per team
  static_assert(8 * ROW1 >= 128 * ROW2, "second half of exchange 1 must survive the first half of exchange 2");
  static constexpr int NPAIR = M / 2 / TL;      // (k, M-k) pairs per lane = 16
  static constexpr int H3 = 256 / TL;           // step-3 butterflies per lane (4 or 2)
};

// Synchronisation between the waves of one TEAM.  A one-wave team needs no barrier at all: the exchange buffer
// is private to the wave and a wave's LDS instructions execute in program order, so only the compiler has to be
// told not to move the reads above the writes.  That lets the 8 waves of a workgroup drift apart and overlap each
// other's global-load and LDS latencies.  Two-wave teams use the workgroup barrier.
template <int TW>
__device__ __forceinline__ void team_sync() {
  if (TW == 1) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

template <typename T, int TW, int NT, int FS>
__global__ __launch_bounds__(64 * TW * NT) void stft2_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int32_t* __restrict__ nframes, int hop,
    const T* __restrict__ window, const cplx<T>* __restrict__ tw, T floor_lin, float floor_db,
    float* __restrict__ out, const int64_t* __restrict__ out_off, const int32_t* __restrict__ frame_sel,
    const int64_t* __restrict__ sel_off, int ablate, unsigned lds_main) {
  using C = Cfg<T, TW>;
  constexpr int TB = NT * FS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // XCD-aware remap (speed only): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.
  // Give every XCD a CONTIGUOUS range of (segment, frame-group) pairs so that neighbouring frame groups -- which
  // write adjacent 64-byte halves of the same output lines and re-read 7/8 of each other's samples -- share an L2.
  const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
  const unsigned orig = blockIdx.y * gx + blockIdx.x;
  const unsigned xq = nwg / 8, xr = nwg % 8, xcd = orig % 8;
  const unsigned wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;   // bijective
  const int seg = (int)(wg / gx);
  const int T_out = nframes[seg];
  const int col0 = (int)(wg % gx) * TB;
  if (col0 >= T_out) return;
  const int tid = threadIdx.x;
  // wave-uniform team index (lets the compiler use scalar base + 32-bit lane offsets for the frame loads)
  const int team = __builtin_amdgcn_readfirstlane(tid / C::TL), q0 = tid % C::TL;
  cplx<T>* ex = reinterpret_cast<cplx<T>*>(smem_raw) + (size_t)team * C::EX;
  const float* xs = x + off[seg];
  // float64 only: log table behind everything else in LDS (never aliased by the tile)
  LogTabEntry* ltab = reinterpret_cast<LogTabEntry*>(smem_raw + lds_main);
  if (sizeof(T) == 8) {
    build_log_table(ltab, tid);
    __syncthreads();
  }

  // per-lane twiddle bases (same for every frame)
  cplx<T> base1[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) base1[h] = tw[2 * (q0 + C::TL * h)];            // W_M^m = W_N^(2m)
  const int n3_lane = q0 % C::R3;                                             // identical for bb and bb + TL
  const cplx<T> base2 = tw[32 * n3_lane];                                     // W_M^(16 n3) = W_N^(32 n3)
  const cplx<T> wlane = tw[q0];                                               // W_N^q

  // diagnostic (ablate bit 256): per-phase cycle stamps of wave 0 of workgroup (0,0), written over out[0..7]
  unsigned long long st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define IRA_STAMP(i) do { if (IRA_ABL(ablate & 256)) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); st[i] = t_; } } while (0)
  static_assert(FS == 1 || FS == 2, "FS == 2 holds frame 0's outputs in registers while frame 1 runs");
  float keep_lo[FS == 2 ? C::NPAIR : 1], keep_hi[FS == 2 ? C::NPAIR : 1], keep_mid = 0.f;
  float lo[C::NPAIR], hi[C::NPAIR], mid = 0.f;

  // The frame loop stays ROLLED (register pressure); each frame's outputs are copied into one of two statically
  // indexed register sets at the end of the iteration.
#pragma unroll 1
  for (int fs = 0; fs < FS; ++fs) {
    // Opaque copy of the lane index: stops LICM from hoisting ~200 address registers out of the frame loop
    // (they would all be spilled); recomputing them per frame is a handful of integer ops.
    int ql = q0;
    asm volatile("" : "+v"(ql));
    const int q = ql;
    const int col = col0 + team * FS + fs;
    const bool live = col < T_out;
    const int64_t frame = live ? (frame_sel ? (int64_t)frame_sel[sel_off[seg] + col] : (int64_t)col) : 0;
    const float* fx = xs + frame * hop;

    if (fs == FS - 1) IRA_STAMP(0);
    // ---- step 1: two 16-point DFTs over n1 straight from global memory ------------------------------------
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const int m = q + C::TL * h;
      cplx<T> v[16];
      // Issue ALL loads of this half first and wait once: left to itself hipcc chains them with vmcnt(1) waits
      // (16 serialised L2 round trips).  Columns past the end re-read frame 0 and are dropped at the store (a
      // per-load select would make hipcc branch around every load).
      float xa[16], xb[16];
      T wa[16], wb[16];
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const int n = n1 * C::MB + m;
        xa[n1] = fx[2 * n]; xb[n1] = fx[2 * n + 1];
        wa[n1] = window[2 * n]; wb[n1] = window[2 * n + 1];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) v[n1] = {(T)xa[n1] * wa[n1], (T)xb[n1] * wb[n1]};
      dft_dif<T, 16>(v);
      cplx<T> p[16];
      powers16<T>(h == 0 ? base1[0] : base1[1], p);
      // exchange 1 write: E1[k1][m], row stride MB + 8 complex; for a fixed k1 lanes hit consecutive m
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        const cplx<T> a = v[brev_bits(k1, 4)];
        ex[k1 * C::ROW1 + m] = (k1 == 0) ? a : ira::cmul(a, p[k1]);
      }
    }
    team_sync<TW>();

    if (fs == FS - 1) IRA_STAMP(1);
    // ---- step 2: two 16-point DFTs over n2, half by half (see Cfg::PAD1) ---------------------------------------------
    {
      cplx<T> p[16];
      powers16<T>(base2, p);
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        const int bb = q + C::TL * h;
        const int k1 = bb / C::R3, n3 = bb % C::R3;
        cplx<T> b2[16];
        // read E1[k1][n2*R3 + n3]: within 8 (16) lanes n3 is consecutive, the next k1 row sits 48 banks further
        if (h == 1) IRA_STAMP(8);
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) b2[n2] = ex[k1 * C::ROW1 + n2 * C::R3 + n3];
        if (h == 1) IRA_STAMP(9);
        dft_dif<T, 16>(b2);
        if (h == 1) IRA_STAMP(10);
        team_sync<TW>();   // every lane's E1 reads of this half are done before the rows are overwritten
        // exchange 2 write: row (k1*16 + k2), stride R3 + 1 complex; half h only touches rows of its own k1 range
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
          const cplx<T> a = b2[brev_bits(k2, 4)];
          ex[(k1 * 16 + k2) * C::ROW2 + n3] = (k2 == 0) ? a : ira::cmul(a, p[k2]);
        }
        if (h == 1) IRA_STAMP(11);
      }
    }
    team_sync<TW>();

    if (fs == FS - 1) IRA_STAMP(2);
    // ---- step 3: R3-point DFTs over n3; lane ends up holding Z[k1 + 16*k2 + 256*k3] for its rows r = k1*16 + k2 -----
    cplx<T> z3[C::H3][C::R3];
#pragma unroll
    for (int h = 0; h < C::H3; ++h) {
      const int r = q + C::TL * h;
      // row r is R3 contiguous complex, rows 9 (17) complex apart: conflict free
#pragma unroll
      for (int n3 = 0; n3 < C::R3; ++n3) z3[h][n3] = ex[r * C::ROW2 + n3];
      dft_dif<T, C::R3>(z3[h]);
    }
    team_sync<TW>();
#pragma unroll
    for (int h = 0; h < C::H3; ++h) {
      const int r = q + C::TL * h;
      const int k1 = r >> 4, k2 = r & 15;
      // exchange 3: Z[k] at k + (k >> 4) = k1 + 17*k2 + 272*k3 -> lanes (k2 fastest) are 34 banks apart: conflict free
#pragma unroll
      for (int k3 = 0; k3 < C::R3; ++k3)
        ex[k1 + 17 * k2 + 272 * k3] = z3[h][brev_bits(k3, C::R3 == 16 ? 4 : 3)];
    }
    team_sync<TW>();

    if (fs == FS - 1) IRA_STAMP(3);
    // ---- post: X[k], X[M-k] from (Z[k], Z[M-k]) -------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < C::NPAIR; ++i) {
      const int k = q + C::TL * i;
      const int kp = (C::M - k) & (C::M - 1);
      const cplx<T> zk = ex[k + (k >> 4)], zp = ex[kp + (kp >> 4)];
      const cplx<T> e = {(T)0.5 * (zk.re + zp.re), (T)0.5 * (zk.im - zp.im)};
      const cplx<T> d = {(T)0.5 * (zk.re - zp.re), (T)0.5 * (zk.im + zp.im)};
      const cplx<T> o = {d.im, -d.re};
      const cplx<T> wk = ira::cmul(wlane, tw[C::TL * i]);       // W_N^k = W_N^q * W_N^(TL*i); second factor wave-uniform
      const cplx<T> pp = ira::cmul(wk, o);
      lo[i] = power_to_db<T>(e.re + pp.re, e.im + pp.im, floor_lin, floor_db, ltab);
      hi[i] = power_to_db<T>(e.re - pp.re, e.im - pp.im, floor_lin, floor_db, ltab);
    }
    {
      const cplx<T> zm = ex[C::M / 2 + (C::M / 32)];
      mid = power_to_db<T>(zm.re, zm.im, floor_lin, floor_db, ltab);
    }
    {
      // A NaN (or infinite) sample anywhere in the frame makes every bin of numpy's rfft NaN (spectrogram.py:150): it shows
      // in Z[0] = sum of the packed inputs (0 * NaN at the Hann end points is NaN too).  One check per frame instead
      // of a NaN test per bin: the floor test above maps NaN to the floor, which is right for every other frame.
      const cplx<T> z0 = ex[0];
      if (!(z0.re - z0.re == (T)0 && z0.im - z0.im == (T)0)) {
        const float qn = __uint_as_float(0x7fc00000u);
#pragma unroll
        for (int i = 0; i < C::NPAIR; ++i) { lo[i] = qn; hi[i] = qn; }
        mid = qn;
      }
    }
    if (fs == FS - 1) IRA_STAMP(4);
    if (FS == 2 && fs == 0) {
#pragma unroll
      for (int i = 0; i < (FS == 2 ? C::NPAIR : 1); ++i) { keep_lo[i] = lo[i]; keep_hi[i] = hi[i]; }
      keep_mid = mid;
    }
    team_sync<TW>();   // exchange buffer is reused by the next frame
  }

  IRA_STAMP(5);
  __syncthreads();   // every team is done with its exchange buffer: the tile may overwrite them
  // ---- all teams' columns into one [F][TB+1] float tile (aliases the exchange buffers) ------------------------------------
  float* tile = reinterpret_cast<float*>(smem_raw);
  {
    const int c = team * FS;
#pragma unroll
    for (int i = 0; i < C::NPAIR; ++i) {
      const int k = q0 + C::TL * i;
      if (FS == 2) {
        tile[k * (TB + 1) + c] = keep_lo[i];
        tile[(C::M - k) * (TB + 1) + c] = keep_hi[i];        // k = 0 -> bin M (Nyquist)
      }
      tile[k * (TB + 1) + c + FS - 1] = lo[i];
      tile[(C::M - k) * (TB + 1) + c + FS - 1] = hi[i];
    }
    if (q0 == 0) {
      if (FS == 2) tile[(C::M / 2) * (TB + 1) + c] = keep_mid;
      tile[(C::M / 2) * (TB + 1) + c + FS - 1] = mid;
    }
  }
  __syncthreads();
  IRA_STAMP(6);
  const int ncol = (T_out - col0 < TB) ? T_out - col0 : TB;
  float* o = out + out_off[seg];
  constexpr int NTHR = 64 * TW * NT;
  // 16-byte stores: TB/4 lanes cover one output row, so a wave instruction writes 64*4/TB rows x TB floats.
  // Rows of the (F, T) matrix are only 4-byte aligned (T is arbitrary); global dwordx4 stores accept that.
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  constexpr int QR = TB / 4;
  static_assert(TB % 4 == 0, "tile columns come in groups of four");
  for (int idx = tid; idx < C::F * QR; idx += NTHR) {
    const int k = idx / QR, c4 = (idx % QR) * 4;
    const float* tp = tile + k * (TB + 1) + c4;
    float* gp = o + (int64_t)k * T_out + col0 + c4;
    if (c4 + 3 < ncol) {
      f4u v = {tp[0], tp[1], tp[2], tp[3]};
      *reinterpret_cast<f4u*>(gp) = v;
    } else {
      for (int c = 0; c < 4; ++c)
        if (c4 + c < ncol) gp[c] = tp[c];
    }
  }
  if ((IRA_ABL(ablate & 256)) && wg == 1 && tid == 0) {
    IRA_STAMP(7);
    printf("STAMPS step1 %llu step2 %llu step3 %llu post %llu keep %llu barrier %llu tile+store %llu | step2(h=1): reads %llu dft16 %llu tw+writes %llu\n",
           st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6],
           st[9] - st[8], st[10] - st[9], st[11] - st[10]);
  }
#undef IRA_STAMP
}

template <typename T, int TW, int NT, int FS>
int32_t launch2(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg, int32_t max_frames,
                int32_t hop, const void* window, const void* tw, double floor_db, float* out, const int64_t* out_off,
                const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st) {
  using C = Cfg<T, TW>;
  constexpr int TB = NT * FS;
  size_t lds_ex = (size_t)NT * C::EX * sizeof(cplx<T>);
  size_t lds_tile = (size_t)C::F * (TB + 1) * sizeof(float);
  const size_t lds_main = ((lds_ex > lds_tile ? lds_ex : lds_tile) + 15) & ~(size_t)15;
  const size_t lds = lds_main + (sizeof(T) == 8 ? LOGTAB_N * sizeof(LogTabEntry) : 0);
  if (lds > 160 * 1024) return IRA_E_SIZE;
  auto kern = stft2_kernel<T, TW, NT, FS>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return ira_hip_status(e);
  }
  const double floor_lin = std::pow(10.0, floor_db / 20.0);
  const int ablate = ira_tune_int("IRA_STFT2_ABLATE", 0);
  dim3 grid((max_frames + TB - 1) / TB, nseg);
  kern<<<grid, 64 * TW * NT, lds, st>>>(x, off, nframes, hop, static_cast<const T*>(window),
                                        static_cast<const cplx<T>*>(tw), (T)floor_lin, (float)floor_db, out, out_off,
                                        frame_sel, sel_off, ablate, (unsigned)lds_main);
  IRA_RETURN_LAUNCH();
}

}  // namespace

// Returns IRA_E_UNSUPPORTED when (n_fft, precision) has no register-resident configuration; the caller then
// falls back to the generic LDS kernel in ira_stft.hip.
int32_t ira_stft2_dispatch(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                           int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                           int32_t precision, double floor_db, float* out, const int64_t* out_off,
                           const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st) {
  if (precision == 32 && n_fft == 4096)
    return launch2<float, 1, 8, 2>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off,
                                   frame_sel, sel_off, st);
  if (precision == 32 && n_fft == 8192)
    return launch2<float, 2, 4, 2>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off,
                                   frame_sel, sel_off, st);
  if (precision == 64 && n_fft == 4096)
    return launch2<double, 1, 4, 2>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off,
                                    frame_sel, sel_off, st);
  if (precision == 64 && n_fft == 8192)
    return launch2<double, 2, 2, 2>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off,
                                    frame_sel, sel_off, st);
  return IRA_E_UNSUPPORTED;
}
