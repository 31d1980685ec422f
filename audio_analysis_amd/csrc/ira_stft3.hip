// STFT v3: the float32 / n_fft = 4096 configuration (the reference's spectrogram default, spectrogram.py:107-160)
// at SIXTEEN one-wave teams per CU.
//
// Why a third kernel: tools/pk_f32_rate.hip shows that on gfx950 one wave can issue an f32 VALU instruction only
// every ~5 cycles while the SIMD retires one every ~2.5, and that packed f32 math saturates the SIMD with a single
// wave.  v2 (ira_stft2.hip) holds a whole 2048-point exchange (16 KB + padding) per wave, which caps the CU at 8
// waves = 2 per SIMD: its ~2200 VALU instructions per frame then cost ~5 cycles each plus every exposed load/LDS
// latency.  The lever is occupancy, and occupancy is LDS: v3 moves the same data through HALF-size exchanges
// (8.4 KB per wave), so 16 waves = 4 per SIMD fit beside nothing else, with <= 128 VGPRs each.
//
// Same transform as v2: packed real FFT, z[n] = xw[2n] + i xw[2n+1], M = 2048 = 16 * 16 * 8, DIF,
//   n = n1*128 + n2*8 + n3,  k = k1 + 16*k2 + 256*k3.
//   step 1  lane m = q + 64h (h = 0, 1): 16-point DFT over n1 from global memory, twiddle W_M^(k1 m)
//   E1      half h at a time: [16 k1][64 m'] complex, row stride 66   -> (k1 = q & 15, n3 = (q >> 4) + 4 hb) reads n2
//   step 2  two 16-point DFTs over n2 (hb = 0, 1), twiddle W_M^(16 k2 n3)
//   E2      half hb at a time: k1 + 16 k2 + 272 n3' complex           -> row r = q + 64 hh = k1 + 16 k2 reads n3
//   step 3  four 8-point DFTs over n3 -> lane holds Z[r + 256 k3]
//   E3      natural order, real parts then imaginary parts through one 2048-float buffer
//   post    (Z[k], Z[M-k]) -> |X[k]|, |X[M-k]| in dB, exactly as v2
// Every LDS access is bank-conflict free: 8-byte accesses go half a wave at a time over 32 bank pairs (the strides
// 66 = 2 mod 32 and 272 = 16 mod 32 spread the 16 x 2 lanes of a half wave), 4-byte accesses are unit stride.
// The 16 teams of a workgroup then write their 16 columns into one [F][17] float tile that aliases the exchange
// buffers and store 64-byte runs of the C-contiguous (F, T) matrix.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "ira_fft_reg.h"

namespace {

using ira::brev_bits;
using ira::cplx;
using ira::dft_dif;
using ira::powers16;

typedef cplx<float> cf;

constexpr int M3 = 2048, F3 = M3 + 1;
constexpr int ROWH = 66;     // E1 half: row stride (complex)
constexpr int E2N3 = 272;    // E2 half: n3' stride (complex)
constexpr int EXC = 1072;    // complex slots per team: max(16*66, 15 + 240 + 3*272 + 1, 2048 floats / 2)
static_assert(EXC >= 16 * ROWH && EXC >= 15 + 16 * 15 + 3 * E2N3 + 1 && EXC * 2 >= M3, "exchange buffer too small");

__device__ __forceinline__ void wave_sync() {
  // One-wave team: LDS instructions of a wave execute in order; only the compiler must not reorder across this.
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// 10 log10((re^2 + im^2) / 4) floored at floor_db.  The quarter belongs to the untangling below, which works on UN-halved
// sums (32 multiplies by 0.5 fewer per frame); the floor is ONE v_max in the dB domain (numpy.maximum(|X|, 10^(floor/20))
// before the logarithm, spectrogram.py:152-156: the same value for every bin at or below the floor; a v_max returns the
// other operand for a NaN like the former `!(p > floor)` compare + select did, and log2(0) = -inf is absorbed too).
// POLY (tuning build, IRA_STFT6_ABLATE=16): the logarithm without the transcendental unit -- p = m 2^(E-127) taken apart
// with integer instructions, 10 log10 m as a degree-7 polynomial in m - 3/2 (1.2e-6 dB).  Why it exists and why it is NOT
// the product path: in tools/micro/stft_epilogue_rate.hip (lock-step waves, no memory) the frame's 33 v_log_f32 cost 1770
// of 7300 SIMD-cycles and the polynomial form issues 36 % faster; in the kernel, whose waves drift apart, v_log_f32 rides
// the transcendental pipe BESIDE the other waves' arithmetic and the polynomial's 9 extra full-rate instructions per bin
// make the kernel 4 % slower (profiles/r05_stft_epilogue.txt).
template <bool POLY>
__device__ __forceinline__ float db_quarter(float re, float im, float floor_db) {
  const float p = re * re + im * im;
  float db;
  if (POLY) {
    const unsigned bits = __float_as_uint(p);
    const float t = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u) - 1.5f;
    const float ef = (float)(bits >> 23);
    float q = 4.3469792483e-02f;
    q = fmaf(q, t, -7.5589056220e-02f);
    q = fmaf(q, t, 1.1321877900e-01f);
    q = fmaf(q, t, -2.1251078291e-01f);
    q = fmaf(q, t, 4.2899189857e-01f);
    q = fmaf(q, t, -9.6519812982e-01f);
    q = fmaf(q, t, 2.8952960832e+00f);
    q = fmaf(q, t, 1.7609133684e+00f - 129.0f * 3.0102999566398120f);    // - 127 (bias) - 2 (the quarter), in units of 10 log10 2
    db = fmaf(ef, 3.0102999566398120f, q);
  } else {
    db = fmaf(3.0102999566398120f, __log2f(p), -6.0205999132796240f);     // 10 log10(p / 4)
  }
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(db), "v"(floor_db));        // (fmaxf adds a canonicalising v_max per operand)
  return r;
}

// Untangling of one bin pair of the packed real transform + dB:  X[k] = (E + P)/2, X[M-k] = conj((E - P)/2) with
// E = Zk + conj Zp, P = W_N^k (-i)(Zk - conj Zp); the halves go into db_quarter's constant.  Shared by the (F, T) and the
// frame-major kernels, which therefore produce the same float32 values (test_frame_major_stft_is_the_exact_transpose).
template <bool POLY = false>
__device__ __forceinline__ void untangle_db(float zkr, float zpr, float zki, float zpi, cf wk, float floor_db, float& lo, float& hi) {
  const cf e = {zkr + zpr, zki - zpi};
  const cf o = {zki + zpi, zpr - zkr};
  const cf pp = ira::cmul(wk, o);
  lo = db_quarter<POLY>(e.re + pp.re, e.im + pp.im, floor_db);
  hi = db_quarter<POLY>(e.re - pp.re, e.im - pp.im, floor_db);
}

// p[k] = W_N^(k idx), k < 16, read from the table (tw[j] = exp(-2 pi i j / N), j < N / 2; the other half by symmetry) instead of
// formed by powers16's product tree: a MEASUREMENT aid (tuning build, IRA_STFT6_ABLATE=128) -- is the float32 spectrogram's
// worst-bin error (2.6e-3 dB) set by the up-to-four chained float32 products per twiddle, or by the transform itself?
__device__ __forceinline__ void powers16_exact(const cf* __restrict__ tw, unsigned idx, cf (&p)[16]) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const unsigned j = ((unsigned)k * idx) & 4095u;
    const cf w = tw[j & 2047u];
    p[k] = (j & 2048u) ? cf{-w.re, -w.im} : w;
  }
}

template <int NT3, bool TF>
__global__ __launch_bounds__(64 * NT3) void stft3_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int32_t* __restrict__ nframes, int hop,
    const float* __restrict__ window, const cf* __restrict__ tw, float floor_lin, float floor_db,
    float* __restrict__ out, const int64_t* __restrict__ out_off, const int32_t* __restrict__ frame_sel,
    const int64_t* __restrict__ sel_off, int ablate, unsigned win_lds_off) {
  constexpr int TB3 = NT3;   // one frame per team -> NT3 output columns per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // XCD-aware bijective remap: each XCD (own L2) gets a contiguous range of (segment, frame group) pairs, so the
  // groups that share 7/8 of their samples and adjacent halves of the same output lines meet in one L2.
  const unsigned gx = gridDim.x, nwg = gridDim.x * gridDim.y;
  const unsigned orig = blockIdx.y * gx + blockIdx.x;
  const unsigned xq = nwg / 8, xr = nwg % 8, xcd = orig % 8;
  const unsigned wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + orig / 8;
  if ((IRA_ABL(ablate & 48)) && orig < 256u * (16 / NT3) * 1u) {
    // diagnostic: stagger the first round of workgroups so that co-resident ones are out of phase
    const unsigned slot = (NT3 == 16) ? ((IRA_ABL(ablate & 16)) ? ((orig >> 3) & 1u) : ((orig >> 3) & 3u))
                                      : ((IRA_ABL(ablate & 16)) ? (orig / 256u) % (16 / NT3) : orig % (16 / NT3));
    for (unsigned i = 0; i < slot * (unsigned)IRA_ABL(ablate >> 8); ++i) __builtin_amdgcn_s_sleep(100);
  }
  const int seg = (int)(wg / gx);
  const int T_out = nframes[seg];
  const int col0 = (int)(wg % gx) * TB3;
  if (col0 >= T_out) return;
  const int tid = threadIdx.x;
  const int team = __builtin_amdgcn_readfirstlane(tid >> 6), q = tid & 63;
  cf* ex = reinterpret_cast<cf*>(smem_raw) + (size_t)team * EXC;
  unsigned long long st[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define IRA_STAMP(i) do { if (IRA_ABL(ablate & 128)) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); st[i] = t_; } } while (0)
  IRA_STAMP(0);
  float* exf = reinterpret_cast<float*>(ex);
  // The Hann window (16 KB) is the same for every frame: one copy in LDS per workgroup instead of 16 KB of L1 traffic
  // per frame (the frame loads were L1-bound).  Lives behind the exchange buffers / tile.
  float* winl = reinterpret_cast<float*>(smem_raw + win_lds_off);
  for (int i = tid; i < 2 * M3; i += 64 * NT3) winl[i] = window[i];
  __syncthreads();

  const int col = col0 + team;
  // Columns past the end transform frame 0 and are dropped at the store (a per-load select makes hipcc branch
  // around every load).
  const int64_t frame = (col < T_out) ? (frame_sel ? (int64_t)frame_sel[sel_off[seg] + col] : (int64_t)col) : 0;
  const float* fx = x + off[seg] + frame * hop;
  if (IRA_ABL(ablate & 64)) fx = x + ((size_t)(wg * TB3 + team) * 4096u) % (size_t)(30720000u - 8192u);   // diagnostic: disjoint frames
  const int k1l = q & 15, n3a = q >> 4;

  // ---- step 1 -------------------------------------------------------------------------------------------------
  cf a1[16];   // half h = 1, held in registers until E1 is free again
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int m = q + 64 * h;
    float xa[16], xb[16], wa[16], wb[16];
    // all loads first, one wait (left alone hipcc serialises them behind vmcnt(1) waits)
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int n = n1 * 128 + m;
      if (IRA_ABL(ablate & 1)) { xa[n1] = (float)n; xb[n1] = (float)(n + 1); } else { xa[n1] = fx[2 * n]; xb[n1] = fx[2 * n + 1]; }
      if (IRA_ABL(ablate & 2)) { wa[n1] = 0.5f; wb[n1] = 0.25f; } else { wa[n1] = winl[2 * n]; wb[n1] = winl[2 * n + 1]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    cf v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = {xa[n1] * wa[n1], xb[n1] * wb[n1]};
    dft_dif<float, 16>(v);
    cf p[16];
    powers16<float>(tw[2 * m], p);                      // W_M^m = W_N^(2m)
    if (h == 0) {
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        const cf a = v[brev_bits(k1, 4)];
        ex[k1 * ROWH + q] = (k1 == 0) ? a : ira::cmul(a, p[k1]);
      }
    } else {
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        const cf a = v[brev_bits(k1, 4)];
        a1[k1] = (k1 == 0) ? a : ira::cmul(a, p[k1]);
      }
    }
  }
  wave_sync();
  IRA_STAMP(1);

  // ---- E1 -> step-2 operands: n2 = 0..7 come from half 0, n2 = 8..15 from half 1 ------------------------------------
  cf b2[2][16];
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

  // ---- step 2 and E2 (half hb = n3 in [4hb, 4hb + 4)) -> step-3 operands ---------------------------------------------
  cf z3[4][8];
  {
    cf p[16];
    dft_dif<float, 16>(b2[0]);
    powers16<float>(tw[32 * n3a], p);                    // W_M^(16 n3) = W_N^(32 n3)
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      const cf a = b2[0][brev_bits(k2, 4)];
      ex[k1l + 16 * k2 + E2N3 * n3a] = (k2 == 0) ? a : ira::cmul(a, p[k2]);
    }
    dft_dif<float, 16>(b2[1]);
    powers16<float>(tw[32 * (n3a + 4)], p);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) b2[1][brev_bits(k2, 4)] = ira::cmul(b2[1][brev_bits(k2, 4)], p[k2]);
  }
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

  // ---- step 3: lane holds Z[r + 256 k3], r = q + 64 hh -----------------------------------------------------------------
#pragma unroll
  for (int hh = 0; hh < 4; ++hh) dft_dif<float, 8>(z3[hh]);

  // ---- E3: real parts, then imaginary parts, natural order ------------------------------------------------------------
  float zkr[16], zpr[16], zki[16], zpi[16], midr, midi;
#pragma unroll
  for (int hh = 0; hh < 4; ++hh)
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) exf[q + 64 * hh + 256 * k3] = z3[hh][brev_bits(k3, 3)].re;
  wave_sync();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = q + 64 * i;
    zkr[i] = exf[k];
    zpr[i] = exf[(M3 - k) & (M3 - 1)];
  }
  midr = exf[M3 / 2];
  wave_sync();
#pragma unroll
  for (int hh = 0; hh < 4; ++hh)
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) exf[q + 64 * hh + 256 * k3] = z3[hh][brev_bits(k3, 3)].im;
  wave_sync();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = q + 64 * i;
    zki[i] = exf[k];
    zpi[i] = exf[(M3 - k) & (M3 - 1)];
  }
  midi = exf[M3 / 2];

  // ---- post: X[k] = E + P, X[M-k] = conj(E - P) with E = (Zk + conj Zp)/2, P = W_N^k (-i)(Zk - conj Zp)/2 ---------------
  const cf wlane = tw[q];
  float lo[16], hi[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const cf wk = ira::cmul(wlane, tw[64 * i]);          // W_N^k = W_N^q W_N^(64 i); second factor wave-uniform
    if (IRA_ABL(ablate & 8)) { lo[i] = zkr[i] + wk.re * zpr[i]; hi[i] = zki[i] - wk.im * zpi[i]; continue; }
    untangle_db(zkr[i], zpr[i], zki[i], zpi[i], wk, floor_db, lo[i], hi[i]);
  }
  float mid = db_quarter<false>(2.0f * midr, 2.0f * midi, floor_db);
  {
    // A NaN (or infinite) sample anywhere in the frame makes every bin of numpy's rfft NaN (spectrogram.py:150): it shows
    // in Z[0] = sum of the packed inputs, which lane 0 holds as its first pair (0 * NaN at the Hann end points is NaN
    // too).  One check per frame instead of a NaN test per bin (db_of's floor test maps NaN to the floor).
    const float z0 = (zkr[0] - zkr[0]) + (zki[0] - zki[0]);              // 0 if finite, NaN otherwise
    const float flag = __shfl(z0, 0, 64);
    if (flag != 0.0f) {
      const float qn = __uint_as_float(0x7fc00000u);
#pragma unroll
      for (int i = 0; i < 16; ++i) { lo[i] = qn; hi[i] = qn; }
      mid = qn;
    }
  }
  IRA_STAMP(2);

  if (TF) {
    // Frame-major output (T, F): the frame's 2049 values are contiguous, so every wave stores its own frame straight
    // from registers in 256-byte runs -- no tile, no workgroup barrier, no partial-line write requests.
    if (col < T_out) {
      float* fo = out + out_off[seg] + (int64_t)col * F3;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = q + 64 * i;
        fo[k] = lo[i];
        fo[M3 - k] = hi[i];                                   // k = 0 -> bin M (Nyquist)
      }
      if (q == 0) fo[M3 / 2] = mid;
    }
    return;
  }

  __syncthreads();
  IRA_STAMP(3);   // every team is done with its exchange buffer: the tile may overwrite them
  float* tile = reinterpret_cast<float*>(smem_raw);      // [F][TB + 1]
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = q + 64 * i;
    tile[k * (TB3 + 1) + team] = lo[i];
    tile[(M3 - k) * (TB3 + 1) + team] = hi[i];            // k = 0 -> bin M (Nyquist)
  }
  if (q == 0) tile[(M3 / 2) * (TB3 + 1) + team] = mid;
  __syncthreads();
  IRA_STAMP(4);

  const int ncol = (T_out - col0 < TB3) ? T_out - col0 : TB3;
  float* o = out + out_off[seg];
  // 16-byte stores: four lanes cover one 64-byte output row segment.  Rows of the (F, T) matrix are only 4-byte
  // aligned (T is arbitrary); global dwordx4 stores accept that.
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  constexpr int QR = TB3 / 4;
  for (int idx = tid; idx < ((IRA_ABL(ablate & 4)) ? 1 : F3 * QR); idx += 64 * NT3) {
    const int k = idx / QR, c4 = (idx % QR) * 4;
    const float* tp = tile + k * (TB3 + 1) + c4;
    float* gp = o + (int64_t)k * T_out + col0 + c4;
    if (c4 + 3 < ncol) {
      f4u v = {tp[0], tp[1], tp[2], tp[3]};
      *reinterpret_cast<f4u*>(gp) = v;
    } else {
      for (int c = 0; c < 4; ++c)
        if (c4 + c < ncol) gp[c] = tp[c];
    }
  }
  if (IRA_ABL(ablate & 128)) {
    unsigned long long t5, t6;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t5) :: "memory");          // stores issued, not yet acknowledged
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t6) :: "memory");
    if ((wg == nwg / 2 || wg == nwg / 2 + 1) && q == 0 && (team == 0 || team == NT3 - 1))
      printf("STAMP wg %u team %d: step1(load+dft) %llu  steps2-3+post %llu  barrier1 %llu  tile %llu  store-issue %llu  store-ack %llu\n",
             wg, team, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], t5 - st[4], t6 - t5);
  }
#undef IRA_STAMP
}

// float32 / n_fft 4096 only; anything else returns IRA_E_UNSUPPORTED and the caller falls through to v2 / v1.
template <int NT3, bool TF>
int32_t launch3(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg, int32_t max_frames,
                int32_t hop, const void* window, const void* tw, double floor_db, float* out, const int64_t* out_off,
                const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st) {
  constexpr int TB3 = NT3;
  constexpr size_t lds_ex = (size_t)NT3 * EXC * sizeof(cf);
  constexpr size_t lds_tile = (size_t)F3 * (TB3 + 1) * sizeof(float);
  constexpr size_t lds_main = ((TF ? lds_ex : (lds_ex > lds_tile ? lds_ex : lds_tile)) + 15) & ~(size_t)15;
  constexpr size_t lds = lds_main + (size_t)2 * M3 * sizeof(float);          // + the window copy
  static_assert(lds <= 160 * 1024, "one workgroup must fit the CU's LDS");
  const hipError_t attr = lds <= 64 * 1024 ? hipSuccess
                                            : hipFuncSetAttribute(reinterpret_cast<const void*>(&stft3_kernel<NT3, TF>),
                                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return ira_hip_status(attr);
  const double floor_lin = std::pow(10.0, floor_db / 20.0);
  const int ablate = ira_tune_int("IRA_STFT3_ABLATE", 0);   // diagnostics
  dim3 grid((max_frames + TB3 - 1) / TB3, nseg);
  stft3_kernel<NT3, TF><<<grid, 64 * NT3, lds, st>>>(x, off, nframes, hop, static_cast<const float*>(window),
                                            static_cast<const cf*>(tw), (float)floor_lin, (float)floor_db, out,
                                            out_off, frame_sel, sel_off, ablate, (unsigned)lds_main);
  IRA_RETURN_LAUNCH();
}


// ------------------------------------------------------------------------------------------------------------
// STFT v6: the frame-major (T, F) variant as a PERSISTENT kernel -- one 16-wave workgroup per CU that walks many tiles of
// 16 frames, every wave transforming "its" frame of each tile on its own, with NO workgroup barrier inside the loop.
//
// Why (round 3, tools/micro/dft16_rate.hip, profiles/r03_dft16_rate.txt): the frame's own instruction stream (1915 VALU
// instructions, the LDS exchanges included) issues at 1.27 wave-instructions per CU-cycle when 16 waves free-run through
// it, but stft3_kernel<16, true> ran at 0.65 -- and still at 0.79 with every load and store ablated.  A 16-wave / 150 KB
// workgroup is the only one its CU can hold, so with one workgroup PER TILE each CU went through launch -> window staging
// -> barrier -> 16 waves loading at once (L1-bound) -> 16 waves computing in lock-step -> stores -> drain, one phase at a
// time, ~230 times per launch.  Here the window is staged once per CU, waves drift out of phase within a few tiles, and
// one wave's loads / stores / LDS round trips run under the arithmetic of the other three on its SIMD.
// The grid is the CU count (every wave has a fixed trip count: no work queue, nothing to drain).
// ------------------------------------------------------------------------------------------------------------
// element at a 32-bit BYTE offset from a wave-uniform base: written this way the access compiles to the scalar-base +
// 32-bit-lane-offset form (an element index is widened to 64 bits first: two extra VALU instructions per access)
template <typename T>
__device__ __forceinline__ T& at32(T* base, unsigned byte_off) {
  return *reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + byte_off);
}

// NT: one-wave teams per workgroup (16 = four waves per SIMD and <= 128 registers; 12 = three per SIMD and <= 170).
// PF: software pipeline -- the NEXT frame's half-0 samples are requested before this frame's results are computed and
//     stored, half 1 before half 0 is transformed.  gfx950 counts loads and stores in ONE in-order counter, so a load
//     issued after a frame's 33 stores cannot be consumed before those stores are acknowledged; issued before them it can.
// AB: ablation bits, instantiated only by the tuning build (IRA_STFT6_ABLATE): 1 no sample loads, 2 no window reads,
//     4 no stores, 8 post-stage twiddles without the scalar table loads, 16 polynomial logarithm, 32 E3 through LDS (the
//     pre-round-5 mirror exchange), 64 samples from LDS + emulated staging (timing only: see load_half),
//     128 every twiddle read from the table instead of formed by products (error measurement).  The product runs AB = 0.
template <int NT, bool PF, int AB>
__global__ __launch_bounds__(64 * NT) void stft6_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ off, const int32_t* __restrict__ nframes, int hop,
    const float* __restrict__ window, const cf* __restrict__ tw, float floor_lin, float floor_db,
    float* __restrict__ out, const int64_t* __restrict__ out_off, const int32_t* __restrict__ frame_sel,
    const int64_t* __restrict__ sel_off, unsigned gx, unsigned ntiles, unsigned win_lds_off, int stagger, int resync) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int team = __builtin_amdgcn_readfirstlane(tid >> 6), q0 = tid & 63;
  cf* ex = reinterpret_cast<cf*>(smem_raw) + (size_t)team * EXC;
  float* exf = reinterpret_cast<float*>(ex);
  float* winl = reinterpret_cast<float*>(smem_raw + win_lds_off);
  for (int i = tid; i < 2 * M3; i += 64 * NT) winl[i] = window[i];
  __syncthreads();                                        // the only workgroup barrier of the kernel

  // Tiles of NT frames in (segment, frame group) order.  Workgroups are dealt round-robin over the 8 XCDs (each with its
  // own L2): XCD c takes the contiguous range [c * per, (c + 1) * per) and its workgroups walk it interleaved, so that the
  // tiles in flight at any time are neighbours (7/8 of a frame's samples are shared with the next one).
  const unsigned nwg = gridDim.x, wg = blockIdx.x;
  const unsigned xcd = wg & 7u, lane_wg = wg >> 3, wg_per_xcd = (nwg + 7u - xcd) >> 3;
  const unsigned per = (ntiles + 7u) >> 3;
  const unsigned t_end = (xcd + 1u) * per < ntiles ? (xcd + 1u) * per : ntiles;
  const float qn = __uint_as_float(0x7fc00000u);

  auto locate = [&](unsigned t, const float*& fxp, float*& fop) -> bool {       // wave-uniform
    const int seg = (int)(t / gx);
    const int col = (int)(t - (unsigned)seg * gx) * NT + team;
    if (col >= nframes[seg]) return false;
    const int64_t frame = frame_sel ? (int64_t)frame_sel[sel_off[seg] + col] : (int64_t)col;
    // wave-uniform bases in scalar registers: every access below is base + 32-bit lane offset
    fxp = x + ira::uniform((long long)(off[seg] + frame * hop));
    fop = out + ira::uniform((long long)(out_off[seg] + (int64_t)col * F3));
    return true;
  };
  auto load_half = [&](const float* fxp, int qq, int h, float (&xa)[16], float (&xb)[16]) {
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const unsigned n = (unsigned)(n1 * 128 + qq + 64 * h);        // unsigned 32-bit lane offsets: scalar base + offset
      if (AB & 64) {
        // timing only: the frame's samples as 8-byte LDS reads (from the window copy: wrong values, right instruction mix) --
        // what a tile's sample span staged ONCE in LDS would make of the 32 global loads per frame-wave; the staging itself
        // is emulated below (three 16-byte global loads + LDS writes per frame-wave)
        const float2 v2 = *reinterpret_cast<const float2*>(winl + ((2u * n) & 4094u));
        xa[n1] = v2.x; xb[n1] = v2.y;
      } else if (AB & 1) { xa[n1] = __uint_as_float(0x3f800000u + n); xb[n1] = __uint_as_float(0x3f000000u + n); }
      else { xa[n1] = at32(fxp, 8u * n); xb[n1] = at32(fxp, 8u * n + 4u); }
    }
  };
  // optional one-off phase offset between the waves of a SIMD (tuning build; measured: no effect, default 0)
  for (int i = 0; i < (team >> 2) * stagger; ++i) __builtin_amdgcn_s_sleep(16);

  const unsigned t_step = wg_per_xcd;
  unsigned t = xcd * per + lane_wg;
  // resync > 0: the sixteen waves meet at a workgroup barrier every `resync` positions of the walk.  Free-running waves drift
  // apart by several tiles, and then the frames in flight on an XCD (32 workgroups x 16 waves x 16 KB) no longer fit its
  // 4 MB L2: every sample was fetched 3.9 times (profiles/r03_traffic_report.json).  Every wave executes the same NUMBER of
  // barriers whatever tiles it skips (a wave past the last frame of a segment has nothing to do there).
  const unsigned t_first = t;
  const unsigned n_pos = t_end > t_first ? (t_end - t_first + t_step - 1u) / t_step : 0u;
  unsigned syncs_done = 0;
  auto sync_up_to = [&](unsigned pos) {                       // pos = positions of the walk this wave has left behind
    if (resync <= 0) return;
    for (const unsigned want = pos / (unsigned)resync; syncs_done < want; ++syncs_done) __syncthreads();
  };
  const float* fx = x;
  float* fo = out;
  bool have = false;
  while (t < t_end && !(have = locate(t, fx, fo))) t += t_step;
  float xa0[16], xb0[16];                                  // half 0 of the frame about to be transformed (PF)
  if (PF && have) load_half(fx, q0, 0, xa0, xb0);

  // The sixteen wave-uniform factors W_N^(64 i) of the post step live in scalar registers for the whole walk: inside the loop
  // they were sixteen scalar loads per frame in four to eight serial round trips (s_waitcnt lgkmcnt(0)).
  cf wuni[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) wuni[i] = tw[64 * i];

  // INVARIANT: the barriers of sync_up_to() are matched by COUNT, not textually -- every wave of the workgroup
  // must execute exactly n_pos / resync of them over its whole walk, whatever tiles it skips.  No exit path of this loop
  // (no `return`, `break` or `continue`) may bypass the sync_up_to() at its end or the final sync_up_to(n_pos) behind it:
  // a wave that left early would leave the other fifteen waiting at a barrier for ever.
  while (have) {
    unsigned tn = t + t_step;
    const float* fxn = fx;
    float* fon = fo;
    bool have_n = false;
    while (tn < t_end && !(have_n = locate(tn, fxn, fon))) tn += t_step;
    // The lane's twiddle factors and their powers are the same for every frame; left to itself the optimiser hoists all of
    // them out of this loop (~190 registers) and spills them.  An opaque copy of the lane index keeps them per-frame work,
    // as in the one-frame kernels: three 8-byte L1 hits and ~220 multiply-adds per frame instead of scratch traffic.
    int q = q0;
    asm volatile("" : "+v"(q));
    if (AB & 64) {
      // the emulated share of the cooperative staging: 3 x 16 bytes per lane from the frame's own samples into the free tail of LDS
      typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
      float* tail = reinterpret_cast<float*>(smem_raw + win_lds_off) + 2 * M3;       // behind the window copy (4 KB are free)
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const f4 v4 = *reinterpret_cast<const f4*>(fx + 4u * (unsigned)(q + 64 * u));
        *reinterpret_cast<f4*>(tail + 4 * (q + 64 * (u & 3))) = v4;
      }
    }
    const int k1l = q & 15, n3a = q >> 4;
    const cf wlane = tw[(unsigned)q];

    // ---- step 1 ---------------------------------------------------------------------------------------------------
    cf a1[16];
    float xa1[16], xb1[16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = q + 64 * h;
      float xa[16], xb[16], wa[16], wb[16];
      if (!PF) load_half(fx, q, h, xa, xb);                // (all loads first, one wait)
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const unsigned n = (unsigned)(n1 * 128 + m);
        if (PF) { xa[n1] = h ? xa1[n1] : xa0[n1]; xb[n1] = h ? xb1[n1] : xb0[n1]; }
        if (AB & 2) { wa[n1] = 0.5f; wb[n1] = 0.25f; }
        else { wa[n1] = winl[2u * n]; wb[n1] = winl[2u * n + 1u]; }
      }
      __builtin_amdgcn_sched_barrier(0);
      cf v[16];
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) v[n1] = {xa[n1] * wa[n1], xb[n1] * wb[n1]};
      if (PF && h == 0) {                                  // half 1's samples: in flight under half 0's DFT
        __builtin_amdgcn_sched_barrier(0);
        load_half(fx, q, 1, xa1, xb1);
        __builtin_amdgcn_sched_barrier(0);
      }
      dft_dif<float, 16>(v);
      cf p[16];
      if (AB & 128) powers16_exact(tw, (unsigned)(2 * m), p);
      else powers16<float>(tw[(unsigned)(2 * m)], p);      // W_M^m = W_N^(2m)
      if (h == 0) {
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
          const cf a = v[brev_bits(k1, 4)];
          ex[k1 * ROWH + q] = (k1 == 0) ? a : ira::cmul(a, p[k1]);
        }
      } else {
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
          const cf a = v[brev_bits(k1, 4)];
          a1[k1] = (k1 == 0) ? a : ira::cmul(a, p[k1]);
        }
      }
    }
    wave_sync();

    // ---- E1 -> step-2 operands ----------------------------------------------------------------------------------------
    cf b2[2][16];
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

    // ---- step 2 and E2 ---------------------------------------------------------------------------------------------------
    cf z3[4][8];
    {
      cf p[16];
      dft_dif<float, 16>(b2[0]);
      if (AB & 128) powers16_exact(tw, (unsigned)(32 * n3a), p);
      else powers16<float>(tw[(unsigned)(32 * n3a)], p);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        const cf a = b2[0][brev_bits(k2, 4)];
        ex[k1l + 16 * k2 + E2N3 * n3a] = (k2 == 0) ? a : ira::cmul(a, p[k2]);
      }
      dft_dif<float, 16>(b2[1]);
      if (AB & 128) powers16_exact(tw, (unsigned)(32 * (n3a + 4)), p);
      else powers16<float>(tw[(unsigned)(32 * (n3a + 4))], p);
#pragma unroll
      for (int k2 = 1; k2 < 16; ++k2) b2[1][brev_bits(k2, 4)] = ira::cmul(b2[1][brev_bits(k2, 4)], p[k2]);
    }
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

    // ---- step 3 ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) dft_dif<float, 8>(z3[hh]);

    // ---- E3: the mirror partner Z[M - k] of every bin the lane holds.  After step 3 the lane already holds Z[q + 64 i'],
    // i' = hh + 4 k3 < 32, in natural order; the partner of k = q + 64 i (i < 16) is register 31 - i of lane (64 - q) & 63 --
    // and for lane 0 its OWN register 32 - i (register 0 for i = 0).  Round 5: a lane permutation through the LDS crossbar
    // (ds_bpermute_b32: no LDS memory, no wave_sync) instead of writing real parts, then imaginary parts, through a
    // 2048-float buffer and reading both orders back: 32 permutes + 32 selects for 130 LDS accesses and four round trips
    // (tools/micro/stft_epilogue_rate.hip: -540 SIMD-cycles per frame; profiles/r05_stft_epilogue.txt).  Pure data movement:
    // the same float32 values as before.  AB & 32 (tuning build) keeps the LDS form as the A/B.
    float zkr[16], zpr[16], zki[16], zpi[16], midr, midi;
    if (AB & 32) {
#pragma unroll
      for (int hh = 0; hh < 4; ++hh)
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) exf[q + 64 * hh + 256 * k3] = z3[hh][brev_bits(k3, 3)].re;
      wave_sync();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = q + 64 * i;
        zkr[i] = exf[k];
        zpr[i] = exf[(M3 - k) & (M3 - 1)];
      }
      midr = exf[M3 / 2];
      wave_sync();
#pragma unroll
      for (int hh = 0; hh < 4; ++hh)
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) exf[q + 64 * hh + 256 * k3] = z3[hh][brev_bits(k3, 3)].im;
      wave_sync();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = q + 64 * i;
        zki[i] = exf[k];
        zpi[i] = exf[(M3 - k) & (M3 - 1)];
      }
      midi = exf[M3 / 2];
      wave_sync();                                          // the next frame's step 1 writes this buffer again
    } else {
      // (the permutes are issued inside the post loop below, bin pair by bin pair: holding all 64 partner values beside the
      // lane's own 64 spills)
      zkr[0] = z3[0][0].re; zki[0] = z3[0][0].im;           // the NaN flag below reads Z[0]
      midr = z3[0][brev_bits(4, 3)].re;                     // Z[1024] = lane 0's register 16 (only lane 0 stores the bin)
      midi = z3[0][brev_bits(4, 3)].im;
      wave_sync();                                          // the next frame's step 1 writes the exchange buffer again
    }
    const int perm_src = ((64 - q) & 63) << 2;              // byte address of the partner lane for ds_bpermute

    // ---- next frame, half 0 (PF): requested BEFORE this frame's results are computed and stored ---------------------------
    if (PF) {
      __builtin_amdgcn_sched_barrier(0);
      if (have_n) load_half(fxn, q, 0, xa0, xb0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- post: X[k] = E + P, X[M-k] = conj(E - P) with E = (Zk + conj Zp)/2, P = W_N^k (-i)(Zk - conj Zp)/2 -> dB -> store.
    // A NaN (or infinite) sample anywhere in the frame makes every bin of numpy's rfft NaN (spectrogram.py:150): it shows
    // in Z[0] = sum of the packed inputs, which lane 0 holds as its first pair -- one flag per frame.  Every result is
    // stored as soon as it exists (nothing of it stays live).
    const float z0 = (zkr[0] - zkr[0]) + (zki[0] - zki[0]);                // 0 if finite, NaN otherwise
    const bool bad = __shfl(z0, 0, 64) != 0.0f;                            // wave-uniform: a branch, not a select per bin
    // per-lane pointers once per frame: every store below is base + immediate offset (as 32-bit lane offsets added to a
    // scalar base each store took an integer add of its own: an unsigned offset cannot be folded into the immediate)
    float* const flo = fo + q;
    float* const fhi = fo + (M3 - q);
    if (__builtin_expect(bad, 0)) {
      if (!(AB & 4)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { flo[64 * i] = qn; fhi[-64 * i] = qn; }
        if (q == 0) fo[M3 / 2] = qn;
      }
    } else {
      float sacc = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const cf wk = (AB & 128) ? tw[(unsigned)(q + 64 * i)]
                                 : ira::cmul(wlane, (AB & 8) ? cf{wlane.im * (float)(i + 1), wlane.re} : wuni[i]);
        float lo, hi;
        if (AB & 32) {
          untangle_db<(AB & 16) != 0>(zkr[i], zpr[i], zki[i], zpi[i], wk, floor_db, lo, hi);
        } else {
          const int ip = 31 - i, il = (32 - i) & 31;        // partner register of lanes q > 0 / of lane 0
          const cf own = z3[i & 3][brev_bits(i >> 2, 3)];
          const cf pv = z3[ip & 3][brev_bits(ip >> 2, 3)], lv = z3[il & 3][brev_bits(il >> 2, 3)];
          const float pr = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_src, __float_as_int(pv.re)));
          const float pi = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_src, __float_as_int(pv.im)));
          untangle_db<(AB & 16) != 0>(own.re, q == 0 ? lv.re : pr, own.im, q == 0 ? lv.im : pi, wk, floor_db, lo, hi);
        }
        if (AB & 4) { sacc += lo + hi; continue; }
        flo[64 * i] = lo;
        fhi[-64 * i] = hi;                                    // k = 0 -> bin M (Nyquist)
      }
      const float mid = db_quarter<(AB & 16) != 0>(2.0f * midr, 2.0f * midi, floor_db);
      if (AB & 4) { if (sacc + mid == 12345.678f) fo[q] = sacc; }
      else if (q == 0) fo[M3 / 2] = mid;
    }
    sync_up_to(have_n ? (tn - t_first) / t_step : n_pos);
    t = tn; fx = fxn; fo = fon; have = have_n;
  }
  sync_up_to(n_pos);
}

// measured (profiles/r04_stft6_resync.txt): 2 brings the fetch traffic from 3.7x the samples down to 1.08x at the same run time
constexpr int IRA_STFT6_RESYNC_DEFAULT = 2;

template <int NT, bool PF>
int32_t launch6(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg, int32_t max_frames, int32_t hop,
                const void* window, const void* tw, double floor_db, float* out, const int64_t* out_off,
                const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st) {
  constexpr size_t lds_main = ((size_t)NT * EXC * sizeof(cf) + 15) & ~(size_t)15;
#ifdef IRA_TUNING_BUILD
  constexpr size_t lds = lds_main + (size_t)2 * M3 * sizeof(float) + 4096;   // + the scratch tail of the AB & 64 emulation
#else
  constexpr size_t lds = lds_main + (size_t)2 * M3 * sizeof(float);
#endif
  static_assert(lds <= 160 * 1024, "one workgroup must fit the CU's LDS");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    cus = 256;
  const unsigned gx = (unsigned)((max_frames + NT - 1) / NT);
  const unsigned long long tiles = (unsigned long long)gx * (unsigned long long)nseg;
  if (tiles > 0xffffffffull) return IRA_E_SIZE;
  unsigned grid = (unsigned)cus;
  if (tiles < grid) grid = (unsigned)tiles;
  if (grid == 0) return IRA_OK;
  const double floor_lin = std::pow(10.0, floor_db / 20.0);
  const int stagger = ira_tune_int("IRA_STFT6_STAGGER", 0);
#define IRA_LAUNCH6(AB)                                                                                                   \
  do {                                                                                                                    \
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&stft6_kernel<NT, PF, AB>),                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
    if (attr != hipSuccess) return ira_hip_status(attr);                                                                  \
    stft6_kernel<NT, PF, AB><<<grid, 64 * NT, lds, st>>>(x, off, nframes, hop, static_cast<const float*>(window),         \
                                              static_cast<const cf*>(tw), (float)floor_lin, (float)floor_db, out, out_off, \
                                              frame_sel, sel_off, gx, (unsigned)tiles, (unsigned)lds_main, stagger,       \
                                              ira_tune_int("IRA_STFT6_RESYNC", IRA_STFT6_RESYNC_DEFAULT));                \
  } while (0)
#ifdef IRA_TUNING_BUILD
  switch (ira_tune_int("IRA_STFT6_ABLATE", 0)) {
    case 1: IRA_LAUNCH6(1); break;
    case 4: IRA_LAUNCH6(4); break;
    case 7: IRA_LAUNCH6(7); break;
    case 16: IRA_LAUNCH6(16); break;
    case 23: IRA_LAUNCH6(23); break;
    case 32: IRA_LAUNCH6(32); break;
    case 39: IRA_LAUNCH6(39); break;
    case 64: IRA_LAUNCH6(64); break;
    case 128: IRA_LAUNCH6(128); break;
    case 68: IRA_LAUNCH6(68); break;
    default: IRA_LAUNCH6(0); break;
  }
#else
  IRA_LAUNCH6(0);
#endif
#undef IRA_LAUNCH6
  IRA_RETURN_LAUNCH();
}

}  // namespace

int32_t ira_stft3_dispatch(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                           int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                           int32_t precision, double floor_db, float* out, const int64_t* out_off,
                           const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st) {
  if (precision != 32 || n_fft != 4096) return IRA_E_UNSUPPORTED;
  const int nt = ira_tune_int("IRA_STFT3_NT", 16);   // tuning
  if (nt == 8)
    return launch3<8, false>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off, frame_sel, sel_off, st);
  if (nt == 4)
    return launch3<4, false>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off, frame_sel, sel_off, st);
  return launch3<16, false>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off, frame_sel, sel_off, st);
}

// Frame-major variant: out[e] is a (T, F) matrix (each frame's F values contiguous).
int32_t ira_stft3_dispatch_tf(const float* x, const int64_t* off, const int32_t* nframes, int32_t nseg,
                              int32_t max_frames, int32_t n_fft, int32_t hop, const void* window, const void* tw,
                              int32_t precision, double floor_db, float* out, const int64_t* out_off,
                              const int32_t* frame_sel, const int64_t* sel_off, hipStream_t st) {
  if (precision != 32 || n_fft != 4096) return IRA_E_UNSUPPORTED;
  if (ira_tune_flag("IRA_STFT_V3"))                       // tuning build: the one-workgroup-per-tile kernel (A/B, ablations)
    return launch3<16, true>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off, frame_sel, sel_off, st);
#ifdef IRA_TUNING_BUILD
  switch (ira_tune_int("IRA_STFT6_VARIANT", 0)) {          // A/B: teams per workgroup x software pipeline
    case 1: return launch6<16, false>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off, frame_sel, sel_off, st);
    case 2: return launch6<12, false>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off, frame_sel, sel_off, st);
    case 3: return launch6<12, true>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off, frame_sel, sel_off, st);
    case 4: return launch6<16, true>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off, frame_sel, sel_off, st);
    default: break;
  }
#endif
  return launch6<16, false>(x, off, nframes, nseg, max_frames, hop, window, tw, floor_db, out, out_off, frame_sel, sel_off, st);
}
