// a19-a21: z-plane AR pole fit (reference analyse/zplane.py:83-158).
//
// The reference solves the covariance-method least squares  min || A a + x ||,  A[n,k] = x[n-k] (n = p..N-1,
// k = 1..p) with an SVD (numpy.linalg.lstsq) over an (N-p) x p matrix -- 4 s of LAPACK per 10 s channel.
// Here:
//   1. ar_lag_kernel    (default) the normal equations of the covariance method have SHIFT structure: with
//                       Phi[a][b] = sum_{n=p}^{N-1} s[n-a] s[n-b],  G[i][j] = Phi[i+1][j+1],  r[j] = -Phi[0][j+1] and
//                         Phi[a+1][b+1] = Phi[a][b] + s[p-1-a] s[p-1-b] - s[N-1-a] s[N-1-b],
//                       so p+1 lag sums Phi[0][0..p] (one pass over the samples, float64 FMAs on exact products of
//                       float32 samples) plus O(p^2) head/tail corrections give all of G: 2(p+1)N flops instead
//                       of p(p+1)N.  The corrections are accumulated with compensated (Neumaier) sums.
//   1'. ar_gram_kernel  (IRA_AR_DENSE=1, cross-check) G = A^T A and r = A^T y as a dense float64 contraction on the matrix cores
//                       (v_mfma_f64_16x16x4_f64).  A is an implicit Hankel view of the signal: every operand
//                       fragment is a shifted window of the LDS-staged samples, nothing is materialised.
//   2. ar_solve_kernel  deterministic reduction of the per-chunk partial Grams, optional ridge, Cholesky,
//                       two triangular solves -> a[0..p] (a[0] = 1).
//   3. poly_roots_kernel  all roots of the monic polynomial by Aberth-Ehrlich iteration in float64
//                       (numpy.roots uses companion-matrix eigenvalues; root ORDER is unspecified there, so
//                       parity is on the sorted set and on the radius statistics).
//   4. fir_numerator_kernel  b[n] = sum_k a[k] h[n-k], n <= Q   (zplane.py:123-142, --zeros option).
// Float32 is not an option for G: on coloured IRs a float32 Gram loses the poles entirely (SURVEY.md
// section 7, hard part 1).
#include <cmath>

#include "ira_common.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int GR_KC = 1024;       // rows (samples n) staged in LDS per tile
constexpr int GR_CHUNK = 8192;    // rows per workgroup (one wave); ~4 waves per SIMD at batch 64 x 10 s
constexpr int GR_GROUP = 64;      // a workgroup accumulates one 64 x 64 block of G
constexpr int GR_PART = GR_GROUP * GR_GROUP + GR_GROUP;   // doubles per partial record (block + rhs slice)
constexpr int GR_MAX_P = 1024;

__host__ __device__ inline int groups_side(int p) { return (p + GR_GROUP - 1) / GR_GROUP; }
__host__ __device__ inline int groups_total(int p) { const int g = groups_side(p); return g * (g + 1) / 2; }

// Operand fragments of one k-step (4 rows n0..n0+3): lane (i = lane & 15, kk = lane >> 4) holds s[n0+kk-1-(16b+i)]
// for each 16-wide block b; the A fragment of block row b and the B fragment of block column b are the SAME value.
template <bool DIAG>
__device__ __forceinline__ void load_frags(const double* __restrict__ lds, long long t0, int ks, long long origin,
                                           long long n_end, int gi, int gj, double (&fa)[4], double (&fb)[4]) {
  const int lane = threadIdx.x;
  const int i = lane & 15, kk = lane >> 4;
  const long long n = t0 + ks + kk;
  const bool valid = n < n_end;                     // rows past the end of the chunk contribute nothing
  const long long base = (n - 1 - i) - origin;      // LDS index of s[n-1-i]
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const double v = lds[base - (GR_GROUP * gi + 16 * a)];
    fa[a] = valid ? v : 0.0;
  }
  if (!DIAG) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double v = lds[base - (GR_GROUP * gj + 16 * b)];
      fb[b] = valid ? v : 0.0;
    }
  }
}

// DIAG and off-diagonal groups are SEPARATE kernels: a fused one carries both accumulator sets (80 + 128 AGPRs)
// and drops to one wave per SIMD.  The operand fetch is software-pipelined one k-step ahead of the MFMAs.
template <bool DIAG>
__global__ __launch_bounds__(64) void ar_gram_kernel(const float* __restrict__ x, const double* __restrict__ x64,
                                                     const int64_t* __restrict__ xoff,
                                                     const int32_t* __restrict__ nlen,
                                                     const double* __restrict__ divisor, int p, int nchunks_max,
                                                     double* __restrict__ part) {
  __shared__ double lds[GR_KC + GR_MAX_P + 8];
  const int e = blockIdx.z;
  const int chunk = blockIdx.x;
  const long long N = nlen[e];
  const long long row0 = (long long)p + (long long)chunk * GR_CHUNK;
  if (row0 >= N) return;
  const long long n_end = (row0 + GR_CHUNK < N) ? row0 + GR_CHUNK : N;
  int gi, gj;
  if (DIAG) {
    gi = gj = blockIdx.y;
  } else {                                           // strictly-lower groups: id -> (gi > gj)
    gi = 1;
    while (gi * (gi + 1) / 2 <= (int)blockIdx.y) ++gi;
    gj = blockIdx.y - gi * (gi - 1) / 2;
  }
  const int gid = gi * (gi + 1) / 2 + gj;            // slot in the lower-triangular partial record
  const int halo = GR_GROUP * groups_side(p) + 1;    // largest lag touched (padded) + 1
  const float* xs = x ? x + xoff[e] : nullptr;
  const double* xd = x64 ? x64 + xoff[e] : nullptr;
  const double div = divisor ? divisor[e] : 1.0;
  const int lane = threadIdx.x;

  constexpr int NACC = DIAG ? 10 : 16;
  d4 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
  double rhs = 0.0;

  for (long long t0 = row0; t0 < n_end; t0 += GR_KC) {
    const long long origin = t0 - halo;
    const int kc = (int)((n_end - t0 < GR_KC) ? n_end - t0 : GR_KC);
    const int kc4 = (kc + 3) & ~3;
    __syncthreads();
    for (int m = lane; m < halo + kc4; m += 64) {
      const long long idx = origin + m;
      lds[m] = (idx >= 0 && idx < N) ? (xd ? xd[idx] : (double)xs[idx]) / div : 0.0;
    }
    __syncthreads();

    double fa[4], fb[4], na[4], nb[4];
    load_frags<DIAG>(lds, t0, 0, origin, n_end, gi, gj, fa, fb);
    for (int ks = 0; ks < kc4; ks += 4) {
      if (ks + 4 < kc4) load_frags<DIAG>(lds, t0, ks + 4, origin, n_end, gi, gj, na, nb);
      int t = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (DIAG && b > a) continue;
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], DIAG ? fa[b] : fb[b], acc[t], 0, 0, 0);
          ++t;
        }
      }
      if (DIAG) {
        // rhs slice r[64 gi + lane] = - sum_n s[n - 1 - j'] s[n]  (VALU, rides under the MFMAs)
        const long long jb = (long long)GR_GROUP * gi + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long long nn = t0 + ks + q;
          const double sn = (nn < n_end) ? lds[nn - origin] : 0.0;
          rhs -= lds[nn - 1 - jb - origin] * sn;
        }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) { fa[a] = na[a]; fb[a] = nb[a]; }
    }
  }

  // D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* out = part + (((long long)e * nchunks_max + chunk) * groups_total(p) + gid) * GR_PART;
  const int col = lane & 15, rq = lane >> 4;
  int t = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (DIAG && b > a) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rq + 4 * r;
        out[(16 * a + row) * GR_GROUP + 16 * b + col] = acc[t][r];
      }
      ++t;
    }
  }
  if (DIAG) out[GR_GROUP * GR_GROUP + lane] = rhs;
}

// ------------------------------------------------------------------------------------------------------------
// Lag sums.  grid (chunks of LAG_CHUNK rows, 1, nb), 256 threads.  The chunk (plus a halo of earlier samples) is staged
// in LDS as float64; a thread owns LPT consecutive lags and a sub-range of the rows and slides an LPT-value window, so
// that a row costs TWO LDS reads for LPT FMAs.  The kernel is bound by LDS data return, not by the FMAs: with four lags
// per thread (the first version) the 12 waves of a CU asked LDS for twice the cycles their FMAs took, and neither
// unrolling nor fewer instructions moved its 0.28 ms (64 x 10 s, p = 64); LPT = 13 (65 lags = 5 x 13) reads 3.25x less.
// The window lives in registers under compile-time renaming (an LPT-row unrolled body, no moves).  Every accumulator sees
// its rows in order, so the sums do not depend on LPT.  Sub-range sums are combined in a fixed order.
// Partial record of element e (doubles):  [nchunks_max][p+1] lag sums | head s[0..p] | tail s[N-1], s[N-2] .. s[N-1-p]
// ------------------------------------------------------------------------------------------------------------
constexpr int LAG_CHUNK = 4096;
constexpr int LAG_THREADS = 256;

__host__ __device__ inline int lag_chunks(long long max_len, int p) {
  return (int)((max_len - p + LAG_CHUNK - 1) / LAG_CHUNK);
}
__host__ __device__ inline long long lag_record_doubles(long long max_len, int p) {
  return (long long)lag_chunks(max_len, p) * (p + 1) + 2ll * (p + 1);
}
// lags per thread: the candidate that wastes the fewest lag slots (ties: the wider one)
inline int lag_per_thread(int nlag) {
  int best = 8, best_waste = 1 << 30;
  for (int c : {16, 13, 12, 10, 8}) {
    const int waste = (nlag + c - 1) / c * c - nlag;
    if (waste < best_waste) { best_waste = waste; best = c; }
  }
  return best;
}

template <int LPT>
__global__ __launch_bounds__(LAG_THREADS) void ar_lag_kernel(const float* __restrict__ x,
                                                             const double* __restrict__ x64,
                                                             const int64_t* __restrict__ xoff,
                                                             const int32_t* __restrict__ nlen,
                                                             const double* __restrict__ divisor, int p,
                                                             int nchunks_max, long long rec_doubles,
                                                             double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int e = blockIdx.z;
  const int chunk = blockIdx.x;
  const long long N = nlen[e];
  const long long row0 = (long long)p + (long long)chunk * LAG_CHUNK;
  if (row0 >= N) return;
  const long long n_end = (row0 + LAG_CHUNK < N) ? row0 + LAG_CHUNK : N;
  const int rows = (int)(n_end - row0);
  const int nlag = p + 1, ngroups = (nlag + LPT - 1) / LPT;
  const int halo = LPT * ngroups;                      // the last group's window reaches back LPT * ngroups - 1 samples
  const long long origin = row0 - halo;              // sample index of lds[0]
  double* lds = reinterpret_cast<double*>(smem_raw);                 // halo + LAG_CHUNK samples; reused for the partials
  const int nsub = ngroups >= LAG_THREADS ? 1 : LAG_THREADS / ngroups;
  const float* xs = x ? x + xoff[e] : nullptr;
  const double* xd = x64 ? x64 + xoff[e] : nullptr;
  const double div = divisor ? divisor[e] : 1.0;
  const int tid = threadIdx.x;
  double* rec = part + (long long)e * rec_doubles;

  // Staging in batches of nine loads (index clamped, value dropped afterwards: a load inside the range test is waited
  // for on the spot, i.e. one memory round trip per sample and thread -- 17 of them in a row before).
  constexpr int LAG_STAGE = 9;                                        // two batches cover halo + LAG_CHUNK at p <= 64
  auto stage = [&](auto src) {                                        // float or double samples: one code path each
    for (int m0 = tid; m0 < halo + rows; m0 += LAG_STAGE * LAG_THREADS) {
      decltype(src[0] + 0) v[LAG_STAGE];
#pragma unroll
      for (int u = 0; u < LAG_STAGE; ++u) {
        long long idx = origin + m0 + u * LAG_THREADS;
        idx = idx < 0 ? 0 : (idx < N ? idx : N - 1);
        v[u] = src[idx];
      }
#pragma unroll
      for (int u = 0; u < LAG_STAGE; ++u) asm volatile("" : "+v"(v[u]));   // no conversion hoisted up to the loads
#pragma unroll
      for (int u = 0; u < LAG_STAGE; ++u) {
        const int m = m0 + u * LAG_THREADS;
        const long long idx = origin + m;
        if (m < halo + rows) lds[m] = (idx >= 0 && idx < N) ? (double)v[u] / div : 0.0;
      }
    }
  };
  if (xd) stage(xd); else stage(xs);
  if (chunk == 0) {
    // head and tail samples for the O(p^2) corrections of the solve kernel
    double* head = rec + (long long)nchunks_max * nlag;
    double* tail = head + nlag;
    for (int m = tid; m < nlag; m += LAG_THREADS) {
      head[m] = (xd ? xd[m] : (double)xs[m]) / div;                         // m <= p < N
      tail[m] = (xd ? xd[N - 1 - m] : (double)xs[N - 1 - m]) / div;
    }
  }
  __syncthreads();

  const int sub_len = (rows + nsub - 1) / nsub;
  const int item = tid;                                 // ngroups * nsub <= LAG_THREADS items
  const bool active = item < ngroups * nsub;
  const int g = active ? item % ngroups : 0, sub = active ? item / ngroups : 0;
  const int b0 = LPT * g;
  double acc[LPT];
#pragma unroll
  for (int j = 0; j < LPT; ++j) acc[j] = 0.0;
  if (active) {
    const int r_begin = sub * sub_len;
    const int r_end = (r_begin + sub_len < rows) ? r_begin + sub_len : rows;
    if (r_begin < r_end) {
      const double* cur = lds + halo + r_begin;       // s[n] for the first row of the sub-range
      const double* lag = cur - b0;                   // s[n - b0]
      double w[LPT];                                  // window: at unrolled step u, slot (j - u) mod LPT holds s[n - b0 - j]
#pragma unroll
      for (int j = 1; j < LPT; ++j) w[j] = lag[-j];
      w[0] = 0.0;
      int r = r_begin;
      for (; r + LPT <= r_end; r += LPT) {
#pragma unroll
        for (int u = 0; u < LPT; ++u) {
          const double sn = cur[u];
          w[(LPT - u) % LPT] = lag[u];               // the newest value takes the slot of the oldest
#pragma unroll
          for (int j = 0; j < LPT; ++j) acc[j] = fma(sn, w[(j - u + LPT) % LPT], acc[j]);
        }
        cur += LPT; lag += LPT;
      }
      for (; r < r_end; ++r) {                        // fewer than LPT rows left: one at a time, the window moved
        const double sn = *cur++;
        w[0] = *lag++;
#pragma unroll
        for (int j = 0; j < LPT; ++j) acc[j] = fma(sn, w[j], acc[j]);
#pragma unroll
        for (int j = LPT - 1; j > 0; --j) w[j] = w[j - 1];
      }
    }
  }
  __syncthreads();                                      // every thread is done with the samples: lds becomes [nsub][LPT * ngroups]
  if (active) {
    double* o = lds + (size_t)sub * (LPT * ngroups) + b0;
#pragma unroll
    for (int j = 0; j < LPT; ++j) o[j] = acc[j];
  }
  __syncthreads();
  for (int b = tid; b < nlag; b += LAG_THREADS) {
    double sum = 0.0;
    for (int sb = 0; sb < nsub; ++sb) sum += lds[(size_t)sb * (LPT * ngroups) + b];
    rec[(long long)chunk * nlag + b] = sum;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Iterative refinement of the normal-equation solution (corrected semi-normal equations): for elements whose Cholesky
// pivots show cond(G) above a threshold, the residual rho = y - A a of the CURRENT coefficients is formed from the
// samples in float64 (rho[n] = -(s[n] + sum_k a_k s[n-k]): a short FIR over the LDS-staged chunk) and its gradient
// g_j = sum_n s[n-j] rho[n] accumulated exactly like the lag sums; the solve kernel then adds G^-1 g to a.  Each step
// multiplies the error by ~cond(G) eps, down to the cond(A) eps level of a QR/SVD solve (the reference's lstsq).
// Same grid and chunking as ar_lag_kernel; partial layout [nchunks_max][p+1] per element (entry 0 unused).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ar_needs_refinement(const double* info, int e, double cond_threshold) {
  const double status = info[IRA_AR_INFO_DOUBLES * e + 0];
  if (status != 0.0 && status != 2.0) return false;                    // no float64 factorisation to refine with (1, 3, 4), or
                                                                       // already solved in double-double (5)
  return info[IRA_AR_INFO_DOUBLES * e + 3] > cond_threshold;            // trace(G) * ||G^-1|| estimate
}

__global__ __launch_bounds__(LAG_THREADS) void ar_grad_kernel(const float* __restrict__ x, const double* __restrict__ x64,
                                                              const int64_t* __restrict__ xoff,
                                                              const int32_t* __restrict__ nlen,
                                                              const double* __restrict__ divisor, int p, int nchunks_max,
                                                              const double* __restrict__ coeffs,
                                                              const double* __restrict__ info, double cond_threshold,
                                                              double* __restrict__ gpart) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int e = blockIdx.z;
  if (!ar_needs_refinement(info, e, cond_threshold)) return;
  const int chunk = blockIdx.x;
  const long long N = nlen[e];
  const long long row0 = (long long)p + (long long)chunk * LAG_CHUNK;
  if (row0 >= N) return;
  const long long n_end = (row0 + LAG_CHUNK < N) ? row0 + LAG_CHUNK : N;
  const int rows = (int)(n_end - row0);
  const int halo = p + 3;
  const long long origin = row0 - halo;
  const int nlag = p + 1, ngroups = (nlag + 3) / 4;
  const int nsub = ngroups >= LAG_THREADS ? 1 : LAG_THREADS / ngroups;
  double* lds = reinterpret_cast<double*>(smem_raw);                 // halo + LAG_CHUNK samples
  double* res = lds + halo + LAG_CHUNK;                              // LAG_CHUNK residuals
  double* co = res + LAG_CHUNK;                                      // p + 1 coefficients
  double* red = co + nlag;                                           // [nsub][4 * ngroups]
  const float* xs = x ? x + xoff[e] : nullptr;
  const double* xd = x64 ? x64 + xoff[e] : nullptr;
  const double div = divisor ? divisor[e] : 1.0;
  const int tid = threadIdx.x;
  for (int m = tid; m < halo + rows; m += LAG_THREADS) {
    const long long idx = origin + m;
    lds[m] = (idx >= 0 && idx < N) ? (xd ? xd[idx] : (double)xs[idx]) / div : 0.0;
  }
  for (int m = tid; m < nlag; m += LAG_THREADS) co[m] = coeffs[(long long)e * nlag + m];
  __syncthreads();
  for (int r = tid; r < rows; r += LAG_THREADS) {
    const double* sn = lds + halo + r;
    double acc = 0.0;
    for (int k = p; k >= 1; --k) acc = fma(co[k], sn[-k], acc);      // small terms first
    res[r] = -(sn[0] + acc);
  }
  __syncthreads();
  const int sub_len = (rows + nsub - 1) / nsub;
  for (int item = tid; item < ngroups * nsub; item += LAG_THREADS) {
    const int g = item % ngroups, sub = item / ngroups;
    const int b0 = 4 * g;
    const int r_begin = sub * sub_len;
    const int r_end = (r_begin + sub_len < rows) ? r_begin + sub_len : rows;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (r_begin < r_end) {
      const double* cur = res + r_begin;
      const double* lag = lds + halo + r_begin - b0;                  // s[n - b0]
      double w1 = lag[-1], w2 = lag[-2], w3 = lag[-3];
      for (int r = r_begin; r < r_end; ++r) {
        const double rn = *cur++;
        const double w0 = *lag++;
        a0 = fma(rn, w0, a0);
        a1 = fma(rn, w1, a1);
        a2 = fma(rn, w2, a2);
        a3 = fma(rn, w3, a3);
        w3 = w2; w2 = w1; w1 = w0;
      }
    }
    double* o = red + (size_t)sub * (4 * ngroups) + b0;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
  __syncthreads();
  double* rec = gpart + ((long long)e * nchunks_max + chunk) * nlag;
  for (int b = tid; b < nlag; b += LAG_THREADS) {
    double sum = 0.0;
    for (int sub = 0; sub < nsub; ++sub) sum += red[(size_t)sub * (4 * ngroups) + b];
    rec[b] = sum;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Reduction of partials + Cholesky + solves.  One 256-thread workgroup per element.
// The p x p matrix lives in LDS when it fits (p <= 128), otherwise in caller-provided global scratch.
// ------------------------------------------------------------------------------------------------------------
constexpr int SV_THREADS = 256;
constexpr int SV_LDS_P = 128;

// L y = r (forward), L^T a = y (backward), column oriented, in place on vec; all threads of the workgroup call.
__device__ __forceinline__ void chol_solve_inplace(const double* G, double* vec, int p, int tid) {
  for (int k = 0; k < p; ++k) {
    if (tid == 0) vec[k] = vec[k] / G[k * p + k];
    __syncthreads();
    const double yk = vec[k];
    for (int i = k + 1 + tid; i < p; i += SV_THREADS) vec[i] -= G[i * p + k] * yk;
    __syncthreads();
  }
  for (int k = p - 1; k >= 0; --k) {
    if (tid == 0) vec[k] = vec[k] / G[k * p + k];
    __syncthreads();
    const double ak = vec[k];
    for (int i = tid; i < k; i += SV_THREADS) vec[i] -= G[k * p + i] * ak;
    __syncthreads();
  }
}

// ---- blocked variants for a Gram matrix that lives in GLOBAL scratch (p > SV_LDS_P; round 5) ---------------------------------
// The column-by-column loops above, run on global memory, cost three workgroup barriers and a read-modify-write of the
// trailing matrix through L2 per COLUMN (and two barriers + an uncoalesced column read per column of every triangular
// sweep): 8.1 ms per 256 fits at the reference CLI's default order 256 (cli.py:245), 70 % of config 4's step there.
// Blocked: a panel of nb columns is factored in LDS (same operations, LDS latency), the trailing matrix is updated ONCE per
// panel from the LDS copy in 4 x 4 register tiles, and a triangular sweep handles nb unknowns per round (one wave solves the
// diagonal block with shuffles, the other rows take the block's contribution with one pass over their nb entries).
__device__ __forceinline__ int sv_block(int p) { return p <= 384 ? 32 : 16; }
__host__ __device__ inline size_t sv_blocked_lds_doubles(int p) {
  const int nb = p <= 384 ? 32 : 16;
  return (size_t)p * (nb + 1) + (size_t)nb * (nb + 1);
}

__device__ void chol_factor_blocked(double* __restrict__ G, int p, double* pan, int tid, double* piv_s, int* fail_s,
                                    double& dmax, double& dmin) {
  const int nb = sv_block(p), ld = nb + 1;
  for (int kb = 0; kb < p; kb += nb) {
    const int w = (p - kb) < nb ? (p - kb) : nb, m = p - kb;
    for (int idx = tid; idx < m * w; idx += SV_THREADS) {
      const int i = idx / w, c = idx - i * w;
      pan[i * ld + c] = G[(long long)(kb + i) * p + kb + c];
    }
    __syncthreads();
    for (int c = 0; c < w; ++c) {
      if (tid == 0) {
        const double d = pan[c * ld + c];
        if (!(d > 0.0)) { *fail_s = 1; *piv_s = 1.0; }
        else *piv_s = sqrt(d);
      }
      __syncthreads();
      const double d = *piv_s;
      dmax = fmax(dmax, d); dmin = fmin(dmin, d);
      for (int i = c + tid; i < m; i += SV_THREADS) pan[i * ld + c] = (i == c) ? d : pan[i * ld + c] / d;
      __syncthreads();
      const int wc = w - c - 1;                                  // columns of the panel still to update
      for (int idx = tid; idx < (m - c - 1) * wc; idx += SV_THREADS) {
        const int ii = idx / wc, jj = idx - ii * wc;
        const int i = c + 1 + ii, j = c + 1 + jj;
        if (i >= j) pan[i * ld + j] -= pan[i * ld + c] * pan[j * ld + c];
      }
      __syncthreads();
    }
    for (int idx = tid; idx < m * w; idx += SV_THREADS) {
      const int i = idx / w, c = idx - i * w;
      if (i >= c) G[(long long)(kb + i) * p + kb + c] = pan[i * ld + c];
    }
    // trailing matrix (rows and columns >= kb + w, lower triangle): G[i][j] -= sum_c L[i][c] L[j][c], 4 x 4 tiles per thread
    const int m2 = m - w, nt = (m2 + 3) >> 2;
    for (int t = tid; t < nt * nt; t += SV_THREADS) {
      const int ti = t / nt, tj = t - ti * nt;
      if (tj > ti) continue;
      const int i0 = w + 4 * ti, j0 = w + 4 * tj;                // panel-row indices
      double acc[4][4] = {};
      for (int c = 0; c < w; ++c) {
        double a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a[r] = pan[(i0 + r < m ? i0 + r : m - 1) * ld + c];
          b[r] = pan[(j0 + r < m ? j0 + r : m - 1) * ld + c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[r][q] = fma(a[r], b[q], acc[r][q]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + r, j = j0 + q;
          if (i < m && j <= i) G[(long long)(kb + i) * p + kb + j] -= acc[r][q];
        }
    }
    __syncthreads();
  }
}

// L y = r, L^T a = y in place on vec (LDS), L in global memory; blk = nb x (nb + 1) doubles of LDS.  All threads call.
__device__ void chol_solve_blocked(const double* __restrict__ G, double* vec, int p, double* blk, int tid) {
  const int nb = sv_block(p), ld = nb + 1;
  for (int kb = 0; kb < p; kb += nb) {
    const int w = (p - kb) < nb ? (p - kb) : nb;
    for (int idx = tid; idx < w * w; idx += SV_THREADS) {
      const int i = idx / w, c = idx - i * w;
      blk[i * ld + c] = G[(long long)(kb + i) * p + kb + c];
    }
    __syncthreads();
    if (tid < IRA_WAVE) {                                        // one wave: the diagonal block, column by column
      double v = tid < w ? vec[kb + tid] : 0.0;
      for (int c = 0; c < w; ++c) {
        const double yc = __shfl(v, c, IRA_WAVE) / blk[c * ld + c];
        if (tid == c) v = yc;
        else if (tid > c && tid < w) v -= blk[tid * ld + c] * yc;
      }
      if (tid < w) vec[kb + tid] = v;
    }
    __syncthreads();
    for (int i = kb + w + tid; i < p; i += SV_THREADS) {
      const double* row = G + (long long)i * p + kb;
      double sacc = vec[i];
      for (int c = 0; c < w; ++c) sacc -= row[c] * vec[kb + c];
      vec[i] = sacc;
    }
    __syncthreads();
  }
  for (int kb = ((p - 1) / nb) * nb; kb >= 0; kb -= nb) {
    const int w = (p - kb) < nb ? (p - kb) : nb;
    for (int idx = tid; idx < w * w; idx += SV_THREADS) {
      const int i = idx / w, c = idx - i * w;
      blk[i * ld + c] = G[(long long)(kb + i) * p + kb + c];
    }
    __syncthreads();
    if (tid < IRA_WAVE) {
      double v = tid < w ? vec[kb + tid] : 0.0;
      for (int c = w - 1; c >= 0; --c) {
        const double ac = __shfl(v, c, IRA_WAVE) / blk[c * ld + c];
        if (tid == c) v = ac;
        else if (tid < c) v -= blk[c * ld + tid] * ac;
      }
      if (tid < w) vec[kb + tid] = v;
    }
    __syncthreads();
    for (int i = tid; i < kb; i += SV_THREADS) {
      double sacc = vec[i];
      for (int c = 0; c < w; ++c) sacc -= G[(long long)(kb + c) * p + i] * vec[kb + c];
      vec[i] = sacc;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(SV_THREADS) void ar_solve_kernel(const double* __restrict__ part,
                                                              const int32_t* __restrict__ nlen, int p,
                                                              int nchunks_max, double ridge,
                                                              double* __restrict__ gscratch,
                                                              double* __restrict__ coeffs,
                                                              double* __restrict__ info, int lag_mode,
                                                              int lag_nchunks_max, long long lag_rec_doubles,
                                                              const double* __restrict__ gpart, double cond_threshold) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ double piv, trace_s, nrm_s;
  __shared__ int fail;
  const int e = blockIdx.x;
  const int tid = threadIdx.x;
  // refinement call (gpart != null): only flagged elements; rhs = gradient of the current residual, a += G^-1 rhs
  if (gpart != nullptr && !ar_needs_refinement(info, e, cond_threshold)) return;
  const long long N = nlen[e];
  const int nchunks = (int)((N - p + GR_CHUNK - 1) / GR_CHUNK);
  const int ng = groups_total(p);
  double* vec = reinterpret_cast<double*>(smem_raw);            // rhs / solution, p doubles
  double* G = (p <= SV_LDS_P) ? vec + p : gscratch + (long long)e * p * p;
  const double* pe = part + (long long)e * nchunks_max * ng * GR_PART;

  if (lag_mode) {
    // ---- G and r from the p+1 lag sums and the head/tail samples (see ar_lag_kernel) -----------------------------------
    const int nlag = p + 1;
    const double* rec = part + (long long)e * lag_rec_doubles;
    const double* head = rec + (long long)lag_nchunks_max * nlag;
    const double* tail = head + nlag;
    const int lchunks = lag_chunks(N, p);
    for (int d = tid; d < nlag; d += SV_THREADS) {
      double c = 0.0;
      for (int ch = 0; ch < lchunks; ++ch) c += rec[(long long)ch * nlag + d];      // Phi[0][d]
      if (d >= 1) vec[d - 1] = -c;
      // walk down diagonal d:  Phi[a][a+d] = Phi[0][d] + sum_{m<a} (s[p-1-m] s[p-1-m-d] - s[N-1-m] s[N-1-m-d])
      double run = c, comp = 0.0;                                                     // Neumaier compensated sum
      for (int a = 1; a + d <= p; ++a) {
        const int m = a - 1;
        const double t1 = head[p - 1 - m] * head[p - 1 - m - d];
        const double t2 = -tail[m] * tail[m + d];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const double term = q == 0 ? t1 : t2;
          const double t = run + term;
          comp += (fabs(run) >= fabs(term)) ? (run - t) + term : (term - t) + run;
          run = t;
        }
        double v = run + comp;
        const int r = a + d - 1, cidx = a - 1;                                        // G[r][cidx], r >= cidx
        if (d == 0) v += ridge;
        G[r * p + cidx] = v;
        G[cidx * p + r] = v;
      }
    }
  } else {
    // ---- deterministic reduction over chunks (fixed order), lower triangle + mirror -------------------------
    for (int idx = tid; idx < p * p; idx += SV_THREADS) {
      const int r = idx / p, c = idx - r * p;
      if (c > r) continue;
      const int gi = r / GR_GROUP, gj = c / GR_GROUP;
      const int gid = gi * (gi + 1) / 2 + gj;
      const int lr = r - gi * GR_GROUP, lc = c - gj * GR_GROUP;
      double s = 0.0;
      for (int ch = 0; ch < nchunks; ++ch) s += pe[((long long)ch * ng + gid) * GR_PART + lr * GR_GROUP + lc];
      if (r == c) s += ridge;
      G[r * p + c] = s;
      G[c * p + r] = s;
    }
    for (int j = tid; j < p; j += SV_THREADS) {
      const int gi = j / GR_GROUP;
      const int gid = gi * (gi + 1) / 2 + gi;
      double s = 0.0;
      for (int ch = 0; ch < nchunks; ++ch)
        s += pe[((long long)ch * ng + gid) * GR_PART + GR_GROUP * GR_GROUP + (j - gi * GR_GROUP)];
      vec[j] = s;
    }
  }
  if (gpart != nullptr) {
    __syncthreads();                                              // vec was written by other threads above
    const int lchunks = lag_chunks(N, p);
    const double* ge = gpart + (long long)e * lag_nchunks_max * (p + 1);
    for (int j = tid; j < p; j += SV_THREADS) {
      double c = 0.0;
      for (int ch = 0; ch < lchunks; ++ch) c += ge[(long long)ch * (p + 1) + j + 1];
      vec[j] = c;
    }
  }
  if (tid == 0) fail = 0;
  __syncthreads();
  if (tid == 0) {
    double tr = 0.0;
    for (int k = 0; k < p; ++k) tr += G[k * p + k];
    trace_s = tr;
  }
  double dmax = 0.0, dmin = INFINITY;

  // ---- Cholesky, right-looking, lower triangle in place ---------------------------------------------------
  const bool in_lds = p <= SV_LDS_P;
  double* pan = vec + p;                                        // blocked path: panel, then the diagonal block (LDS)
  double* blk = pan + (size_t)p * (sv_block(p) + 1);
  if (!in_lds) {
    chol_factor_blocked(G, p, pan, tid, &piv, &fail, dmax, dmin);
  } else
  for (int k = 0; k < p; ++k) {
    if (tid == 0) {
      const double d = G[k * p + k];
      if (!(d > 0.0)) { fail = 1; piv = 1.0; }
      else piv = sqrt(d);
    }
    __syncthreads();
    const double d = piv;
    dmax = fmax(dmax, d); dmin = fmin(dmin, d);
    for (int i = k + tid; i < p; i += SV_THREADS) G[i * p + k] = (i == k) ? d : G[i * p + k] / d;
    __syncthreads();
    // trailing update: G[i][j] -= L[i][k] L[j][k] for k < j <= i
    const int m = p - k - 1;
    for (int idx = tid; idx < m * m; idx += SV_THREADS) {
      const int ii = idx / m, jj = idx - ii * m;
      if (jj > ii) continue;
      const int i = k + 1 + ii, j = k + 1 + jj;
      G[i * p + j] -= G[i * p + k] * G[j * p + k];
    }
    __syncthreads();
  }
  if (in_lds) chol_solve_inplace(G, vec, p, tid); else chol_solve_blocked(G, vec, p, blk, tid);
  double* co = coeffs + (long long)e * (p + 1);
  if (gpart != nullptr) {
    if (!fail)
      for (int j = tid; j < p; j += SV_THREADS) co[j + 1] += vec[j];
    if (tid == 0 && !fail) info[IRA_AR_INFO_DOUBLES * e + 0] = 2.0;  // refined
    return;
  }
  if (tid == 0) co[0] = 1.0;
  for (int j = tid; j < p; j += SV_THREADS) co[j + 1] = vec[j];
  if (info == nullptr) return;
  // cond(G) estimate: lambda_max <= trace(G); 1/lambda_min from two inverse iterations on the factor, started from the
  // alternating vector (the weakest direction of a low-passed signal's Gram is the one at Nyquist).  Within a factor p
  // above the true condition number -- it only decides whether ira_ar_refine has work to do.
  double cond_est = INFINITY;
  if (!fail) {
    __syncthreads();
    for (int j = tid; j < p; j += SV_THREADS) vec[j] = (j & 1) ? -1.0 : 1.0;
    __syncthreads();
    for (int it = 0; it < 2; ++it) {
      if (tid == 0) {
        double q = 0.0;
        for (int j = 0; j < p; ++j) q += vec[j] * vec[j];
        nrm_s = sqrt(q);
      }
      __syncthreads();
      const double inv = 1.0 / nrm_s;
      for (int j = tid; j < p; j += SV_THREADS) vec[j] *= inv;
      __syncthreads();
      if (in_lds) chol_solve_inplace(G, vec, p, tid); else chol_solve_blocked(G, vec, p, blk, tid);
    }
    if (tid == 0) {
      double q = 0.0;
      for (int j = 0; j < p; ++j) q += vec[j] * vec[j];
      cond_est = trace_s * sqrt(q);
    }
  }
  if (tid == 0) {
    info[IRA_AR_INFO_DOUBLES * e + 0] = (double)fail;
    info[IRA_AR_INFO_DOUBLES * e + 1] = dmax;   // largest / smallest Cholesky pivot
    info[IRA_AR_INFO_DOUBLES * e + 2] = dmin;
    info[IRA_AR_INFO_DOUBLES * e + 3] = cond_est;
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same solve for p <= 64 on ONE WAVE per element (round 4).  ar_solve_kernel spends its time at workgroup barriers: the
// factorisation and its six triangular sweeps (the solve and two inverse iterations for the condition estimate) are ~1000
// barriers of a 256-thread workgroup for 87 k multiply-adds -- 0.25 ms per 256 elements at 2-18 % VALU activity.  Here lane i
// owns row i of G in the wave's own LDS (row stride 65 doubles: a column walk touches every bank pair once), the right-hand
// side lives one value per lane in a register, and nothing waits for another wave:
//   * Cholesky LEFT-looking: column k of L is G[i][k] - sum_{m<k} L[i][m] L[k][m] for every lane i >= k at once -- the m loop
//     only READS (own row, and row k as a broadcast), so its loads pipeline; one write per lane and column.  The running
//     subtraction visits m in increasing order, exactly the order in which the right-looking kernel applies its rank-one
//     updates to that entry: the factor is the same, bit for bit.
//   * triangular sweeps: the pivot lane's value crosses the wave with v_readlane; the division by the diagonal, the
//     multiply-subtract and the sequential norm sums are the workgroup kernel's own operations in its own order.
// Same inputs, outputs, status codes and info record as ar_solve_kernel in lag mode (first solve and refinement mode).
// ------------------------------------------------------------------------------------------------------------
constexpr int SW_LD = 65;

__device__ __forceinline__ double wave_bcast(double v, int src_lane) {       // src_lane wave-uniform
  int w[2];
  __builtin_memcpy(w, &v, 8);
  w[0] = __builtin_amdgcn_readlane(w[0], src_lane);
  w[1] = __builtin_amdgcn_readlane(w[1], src_lane);
  __builtin_memcpy(&v, w, 8);
  return v;
}

// L y = r, then L^T a = y, on the lane-resident vector (lane j holds entry j)
__device__ __forceinline__ double wave_chol_solve(const double* G, double vec, int p, int lane) {
  for (int k = 0; k < p; ++k) {
    const double yk = wave_bcast(vec, k) / G[k * SW_LD + k];
    const double lik = G[(lane < p ? lane : 0) * SW_LD + k];
    if (lane == k) vec = yk;
    else if (lane > k && lane < p) vec -= lik * yk;
  }
  for (int k = p - 1; k >= 0; --k) {
    const double ak = wave_bcast(vec, k) / G[k * SW_LD + k];
    const double lki = G[k * SW_LD + (lane < p ? lane : 0)];
    if (lane == k) vec = ak;
    else if (lane < k) vec -= lki * ak;
  }
  return vec;
}

__global__ __launch_bounds__(64) void ar_solve_wave_kernel(const double* __restrict__ part, const int32_t* __restrict__ nlen,
                                                           int p, double ridge, double* __restrict__ coeffs,
                                                           double* __restrict__ info, int lag_nchunks_max,
                                                           long long lag_rec_doubles, const double* __restrict__ gpart,
                                                           double cond_threshold) {
  __shared__ double G[64 * SW_LD];
  __shared__ double rhs[64];
  const int e = blockIdx.x, lane = threadIdx.x;
  if (gpart != nullptr && !ar_needs_refinement(info, e, cond_threshold)) return;
  const long long N = nlen[e];
  const int nlag = p + 1;
  const double* rec = part + (long long)e * lag_rec_doubles;
  const double* head = rec + (long long)lag_nchunks_max * nlag;
  const double* tail = head + nlag;
  const int lchunks = lag_chunks(N, p);
  // ---- G (lower triangle) and r from the lag sums and the head / tail samples: lane d walks diagonal d, as ar_solve_kernel ----
  for (int d = lane; d < nlag; d += 64) {
    double c = 0.0;
    for (int ch = 0; ch < lchunks; ++ch) c += rec[(long long)ch * nlag + d];
    if (d >= 1) rhs[d - 1] = -c;
    double run = c, comp = 0.0;
    for (int a = 1; a + d <= p; ++a) {
      const int m = a - 1;
      const double t1 = head[p - 1 - m] * head[p - 1 - m - d];
      const double t2 = -tail[m] * tail[m + d];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const double term = q == 0 ? t1 : t2;
        const double t = run + term;
        comp += (fabs(run) >= fabs(term)) ? (run - t) + term : (term - t) + run;
        run = t;
      }
      double v = run + comp;
      if (d == 0) v += ridge;
      G[(a + d - 1) * SW_LD + (a - 1)] = v;                      // G[r][c], r >= c
    }
  }
  __builtin_amdgcn_wave_barrier();
  double vec = lane < p ? rhs[lane] : 0.0;
  if (gpart != nullptr) {
    const double* ge = gpart + (long long)e * lag_nchunks_max * (p + 1);
    double c = 0.0;
    if (lane < p)
      for (int ch = 0; ch < lchunks; ++ch) c += ge[(long long)ch * (p + 1) + lane + 1];
    vec = c;
  }
  double trace = 0.0;
  for (int k = 0; k < p; ++k) trace += G[k * SW_LD + k];
  // ---- Cholesky, left-looking ----------------------------------------------------------------------------------------------
  int fail = 0;
  double dmax = 0.0, dmin = INFINITY;
  const int row = (lane < p ? lane : 0) * SW_LD;
  for (int k = 0; k < p; ++k) {
    double s = G[row + k];
    const double* rk = G + k * SW_LD;
    const double* ri = G + row;
#pragma unroll 8
    for (int m = 0; m < k; ++m) s -= ri[m] * rk[m];
    const double dkk = wave_bcast(s, k);
    double piv;
    if (!(dkk > 0.0)) { fail = 1; piv = 1.0; }
    else piv = sqrt(dkk);
    dmax = fmax(dmax, piv); dmin = fmin(dmin, piv);
    if (lane >= k && lane < p) G[row + k] = (lane == k) ? piv : s / piv;
    __builtin_amdgcn_wave_barrier();
  }
  vec = wave_chol_solve(G, vec, p, lane);
  double* co = coeffs + (long long)e * (p + 1);
  if (gpart != nullptr) {
    if (!fail && lane < p) co[lane + 1] += vec;
    if (lane == 0 && !fail) info[IRA_AR_INFO_DOUBLES * e + 0] = 2.0;  // refined
    return;
  }
  if (lane == 0) co[0] = 1.0;
  if (lane < p) co[lane + 1] = vec;
  if (info == nullptr) return;
  double cond_est = INFINITY;
  if (!fail) {
    vec = lane < p ? ((lane & 1) ? -1.0 : 1.0) : 0.0;
    for (int it = 0; it < 2; ++it) {
      double q = 0.0;
      for (int j = 0; j < p; ++j) { const double vj = wave_bcast(vec, j); q += vj * vj; }
      const double inv = 1.0 / sqrt(q);
      vec *= inv;
      vec = wave_chol_solve(G, vec, p, lane);
    }
    double q = 0.0;
    for (int j = 0; j < p; ++j) { const double vj = wave_bcast(vec, j); q += vj * vj; }
    cond_est = trace * sqrt(q);
  }
  if (lane == 0) {
    info[IRA_AR_INFO_DOUBLES * e + 0] = (double)fail;
    info[IRA_AR_INFO_DOUBLES * e + 1] = dmax;
    info[IRA_AR_INFO_DOUBLES * e + 2] = dmin;
    info[IRA_AR_INFO_DOUBLES * e + 3] = cond_est;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Ill-conditioned fits: the normal equations in DOUBLE-DOUBLE arithmetic (~32 significant digits).
// The reference solves min ||A a + y|| by SVD (numpy.linalg.lstsq, zplane.py:117), accurate to ~cond(A) eps.  float64 normal
// equations lose cond(A)^2 eps: the corrected-semi-normal-equation steps of ira_ar_refine recover lstsq's accuracy while
// cond(G) eps < 1 (cond(G) <~ 1e12 in practice); beyond that the float64 Cholesky factor itself is noise.  A float32 recording
// low-passed at 500 Hz has cond(A) ~ 3e8, cond(G) ~ 1e17 (tests/golden/ar_illcond.npz, SURVEY.md section 7 hard part 1): its
// float64 normal equations are off by 35 % in the coefficients.  But G = A^T A is made of sums of products of the samples,
// and those can be formed EXACTLY: a product of two float64 values is hi + lo by one fma, and a double-double accumulator
// keeps ~106 bits.  With G and r good to 1e-30 and the Cholesky factorisation and the two triangular solves carried in the
// same arithmetic, the solution is good to cond(G) x 1e-32 -- an exact rational solve of that golden case agrees with it to
// 1e-15 and with the reference's lstsq to 2e-9 (pole radii 4e-8).  Only flagged elements (Cholesky pivot not positive, or
// the cond(G) estimate above the threshold) do any work: a rare path, written for clarity.
//   ar_lag_dd_kernel    grid (chunks of LAG_CHUNK rows, 1, nb): p+1 lag sums of the chunk, one or more lags per thread
//   ar_solve_dd_kernel  one workgroup per element: G from the lag sums by the same diagonal walk as ar_solve_kernel (exact
//                       products of head / tail samples), Cholesky + solves in global scratch (p^2 double-doubles).
// A pivot that is not positive RELATIVE to the trace (lstsq's own rank cut, squared; at least 1e-26) means G is singular
// to the reference's working precision -- the element is
// given status 1 (whatever brought it here) and ira_ar_minnorm returns lstsq's minimum-norm solution as before.
// ------------------------------------------------------------------------------------------------------------
struct dd { double hi, lo; };
// `#pragma clang fp contract(off)` in every routine: this file is compiled with fused multiply-add contraction on, and a
// product that gets fused into the addition of an error-free transformation (s = a + b*c computed as ONE fma while the error
// term assumes s = fl(a + fl(b*c))) silently turns double-double into plain float64 -- seen as a 6e-8 instead of 1e-15
// agreement with an exact rational solve.
__device__ __forceinline__ dd dd_two_sum(double a, double b) {
#pragma clang fp contract(off)
  const double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd dd_fast_two_sum(double a, double b) {      // |a| >= |b|
#pragma clang fp contract(off)
  const double s = a + b;
  return {s, b - (s - a)};
}
__device__ __forceinline__ dd dd_two_prod(double a, double b) {
#pragma clang fp contract(off)
  const double p = a * b;
  return {p, fma(a, b, -p)};
}
__device__ __forceinline__ dd dd_add(dd x, dd y) {
#pragma clang fp contract(off)
  const dd s = dd_two_sum(x.hi, y.hi);
  const dd t = dd_two_sum(x.lo, y.lo);
  const dd u = dd_fast_two_sum(s.hi, s.lo + t.hi);
  return dd_fast_two_sum(u.hi, u.lo + t.lo);
}
__device__ __forceinline__ dd dd_neg(dd x) { return {-x.hi, -x.lo}; }
__device__ __forceinline__ dd dd_mul(dd x, dd y) {
#pragma clang fp contract(off)
  dd p = dd_two_prod(x.hi, y.hi);
  p.lo += x.hi * y.lo + x.lo * y.hi;
  return dd_fast_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_div(dd x, dd y) {
#pragma clang fp contract(off)
  const double q1 = x.hi / y.hi;
  dd r = dd_add(x, dd_neg(dd_mul(y, dd{q1, 0.0})));
  const double q2 = r.hi / y.hi;
  r = dd_add(r, dd_neg(dd_mul(y, dd{q2, 0.0})));
  const double q3 = r.hi / y.hi;
  const dd q = dd_fast_two_sum(q1, q2);
  return dd_add(q, dd{q3, 0.0});
}
__device__ __forceinline__ dd dd_sqrt(dd x) {                            // x > 0
#pragma clang fp contract(off)
  const double s = sqrt(x.hi);
  const dd r = dd_add(x, dd_neg(dd_two_prod(s, s)));
  return dd_fast_two_sum(s, r.hi / (2.0 * s));
}

constexpr double AR_DD_SOLVED = 5.0;          // info[0]: solved by the double-double normal equations
__device__ __forceinline__ bool ar_needs_dd(const double* info, int e, double cond_threshold) {
  const double status = info[IRA_AR_INFO_DOUBLES * e + 0];
  if (status == 1.0) return true;                                      // the float64 factorisation broke down
  if (status != 0.0) return false;                                     // not finite / already handled
  return !(info[IRA_AR_INFO_DOUBLES * e + 3] <= cond_threshold);        // estimate above the threshold (or NaN)
}

__global__ __launch_bounds__(LAG_THREADS) void ar_lag_dd_kernel(const float* __restrict__ x, const double* __restrict__ x64,
                                                                const int64_t* __restrict__ xoff,
                                                                const int32_t* __restrict__ nlen,
                                                                const double* __restrict__ divisor, int p, int nchunks_max,
                                                                const double* __restrict__ info, double cond_threshold,
                                                                double* __restrict__ ddpart) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int e = blockIdx.z, chunk = blockIdx.x;
  if (!ar_needs_dd(info, e, cond_threshold)) return;
  const long long N = nlen[e];
  const long long row0 = (long long)p + (long long)chunk * LAG_CHUNK;
  if (row0 >= N) return;
  const long long n_end = (row0 + LAG_CHUNK < N) ? row0 + LAG_CHUNK : N;
  const int rows = (int)(n_end - row0);
  double* lds = reinterpret_cast<double*>(smem_raw);                   // samples row0 - p .. n_end - 1
  const float* xs = x ? x + xoff[e] : nullptr;
  const double* xd = x64 ? x64 + xoff[e] : nullptr;
  const double div = divisor ? divisor[e] : 1.0;
  const int tid = threadIdx.x;
  for (int m = tid; m < p + rows; m += LAG_THREADS) {
    const long long idx = row0 - p + m;                                // >= 0
    lds[m] = (xd ? xd[idx] : (double)xs[idx]) / div;                   // the same rounded values every other AR kernel uses
  }
  __syncthreads();
  double* out = ddpart + (((long long)e * nchunks_max + chunk) * (p + 1)) * 2;
  for (int l = tid; l <= p; l += LAG_THREADS) {
    dd acc = {0.0, 0.0};
    const double* cur = lds + p;                                       // s[n]
    const double* lag = lds + p - l;                                   // s[n - l]
    for (int r = 0; r < rows; ++r) {
      const dd pr = dd_two_prod(cur[r], lag[r]);
      const dd s1 = dd_two_sum(acc.hi, pr.hi);
      acc = dd_fast_two_sum(s1.hi, s1.lo + (acc.lo + pr.lo));
    }
    out[2 * l] = acc.hi;
    out[2 * l + 1] = acc.lo;
  }
}

__global__ __launch_bounds__(SV_THREADS) void ar_solve_dd_kernel(const double* __restrict__ part,
                                                                 const double* __restrict__ ddpart,
                                                                 const int32_t* __restrict__ nlen, int p, double ridge,
                                                                 double* __restrict__ ddscratch,
                                                                 double* __restrict__ coeffs, double* __restrict__ info,
                                                                 int lag_nchunks_max, long long lag_rec_doubles,
                                                                 double cond_threshold) {
  __shared__ dd piv_s, trace_s;
  __shared__ int fail;
  const int e = blockIdx.x, tid = threadIdx.x;
  if (!ar_needs_dd(info, e, cond_threshold)) return;
  const long long N = nlen[e];
  const int nlag = p + 1;
  const double* rec = part + (long long)e * lag_rec_doubles;
  const double* head = rec + (long long)lag_nchunks_max * nlag;
  const double* tail = head + nlag;
  const double* pe = ddpart + ((long long)e * lag_nchunks_max * nlag) * 2;
  const int lchunks = lag_chunks(N, p);
  dd* G = reinterpret_cast<dd*>(ddscratch + (long long)e * 2 * ((long long)p * p + p));   // p x p, then the vector
  dd* vec = G + (long long)p * p;
  for (int d = tid; d < nlag; d += SV_THREADS) {
    dd c = {0.0, 0.0};
    for (int ch = 0; ch < lchunks; ++ch) c = dd_add(c, dd{pe[((long long)ch * nlag + d) * 2], pe[((long long)ch * nlag + d) * 2 + 1]});
    if (d >= 1) vec[d - 1] = dd_neg(c);                                  // r = -Phi[0][d]
    dd run = c;
    for (int a = 1; a + d <= p; ++a) {                                   // Phi[a][a+d], see ar_solve_kernel
      const int m = a - 1;
      run = dd_add(run, dd_two_prod(head[p - 1 - m], head[p - 1 - m - d]));
      run = dd_add(run, dd_neg(dd_two_prod(tail[m], tail[m + d])));
      dd v = run;
      if (d == 0) v = dd_add(v, dd{ridge, 0.0});
      const int r = a + d - 1, cidx = a - 1;
      G[(long long)r * p + cidx] = v;
      G[(long long)cidx * p + r] = v;
    }
  }
  if (tid == 0) fail = 0;
  __syncthreads();
  if (tid == 0) {
    dd tr = {0.0, 0.0};
    for (int k = 0; k < p; ++k) tr = dd_add(tr, G[(long long)k * p + k]);
    trace_s = tr;
  }
  __syncthreads();
  // numpy.linalg.lstsq(rcond=None) drops singular values of A below eps * max(rows, columns) * sigma_max, i.e. eigenvalues of
  // G = A^T A below the SQUARE of that ratio times lambda_max (<= trace): a pivot under that cut means the reference
  // truncates at least one direction, and the element goes to ira_ar_minnorm like every other rank-deficient fit
  // (with 1e-26 alone, singular-value ratios between 1e-13 and eps * rows got a full-rank solution).
  // This is a SUFFICIENT test, not lstsq's criterion itself: a Cholesky pivot bounds lambda_min only from
  // above (pivot_k >= lambda_min) and the trace bounds lambda_max only from above (trace <= p lambda_max), so an element whose
  // lambda_min sits just below the cut while every pivot stays above it keeps the full-rank double-double solution where the
  // reference truncates one direction.  ira_ar_minnorm's eigen-decomposition applies the exact criterion to every element
  // that does trip this test; for the others the two solutions differ by the contribution of a direction with
  // sigma / sigma_max ~ eps * rows, which the double-double arithmetic resolves (cond(G) * 1e-32).
  const double rc = 2.220446049250313e-16 * (double)((N - p) > p ? (N - p) : p);
  const double tiny = fmax(1e-26, rc * rc) * trace_s.hi;
  for (int k = 0; k < p; ++k) {
    if (tid == 0) {
      const dd d = G[(long long)k * p + k];
      if (!(d.hi > tiny)) { fail = 1; piv_s = dd{1.0, 0.0}; }
      else piv_s = dd_sqrt(d);
    }
    __syncthreads();
    const dd d = piv_s;
    for (int i = k + tid; i < p; i += SV_THREADS)
      G[(long long)i * p + k] = (i == k) ? d : dd_div(G[(long long)i * p + k], d);
    __syncthreads();
    const int m = p - k - 1;
    for (int idx = tid; idx < m * m; idx += SV_THREADS) {
      const int ii = idx / m, jj = idx - ii * m;
      if (jj > ii) continue;
      const int i = k + 1 + ii, j = k + 1 + jj;
      G[(long long)i * p + j] = dd_add(G[(long long)i * p + j], dd_neg(dd_mul(G[(long long)i * p + k], G[(long long)j * p + k])));
    }
    __syncthreads();
  }
  if (fail) {
    // Singular to lstsq's rank cut even in double-double arithmetic.  The status is WRITTEN, not left as it was: an
    // element that came here on its condition estimate alone still carries status 0 (its float64 pivots stayed positive by
    // rounding noise) and would otherwise be refined from a meaningless float64 factor and reported as solved.  Status 1
    // hands it to ira_ar_minnorm, which returns lstsq's minimum-norm solution (status 4 + rank).
    if (tid == 0) info[IRA_AR_INFO_DOUBLES * e + 0] = 1.0;
    return;
  }
  for (int k = 0; k < p; ++k) {                         // L y = r
    if (tid == 0) vec[k] = dd_div(vec[k], G[(long long)k * p + k]);
    __syncthreads();
    const dd yk = vec[k];
    for (int i = k + 1 + tid; i < p; i += SV_THREADS) vec[i] = dd_add(vec[i], dd_neg(dd_mul(G[(long long)i * p + k], yk)));
    __syncthreads();
  }
  for (int k = p - 1; k >= 0; --k) {                    // L^T a = y
    if (tid == 0) vec[k] = dd_div(vec[k], G[(long long)k * p + k]);
    __syncthreads();
    const dd ak = vec[k];
    for (int i = tid; i < k; i += SV_THREADS) vec[i] = dd_add(vec[i], dd_neg(dd_mul(G[(long long)k * p + i], ak)));
    __syncthreads();
  }
  double* co = coeffs + (long long)e * (p + 1);
  if (tid == 0) co[0] = 1.0;
  for (int j = tid; j < p; j += SV_THREADS) co[j + 1] = vec[j].hi + vec[j].lo;
  if (tid == 0) info[IRA_AR_INFO_DOUBLES * e + 0] = AR_DD_SOLVED;
}

// ------------------------------------------------------------------------------------------------------------
// Rank-deficient fits: the minimum-norm least-squares solution the reference's lstsq returns (zplane.py:117).
// When a Cholesky pivot of G = A^T A is not positive (ar_solve_kernel status 1: a constant segment, digital silence
// after a few taps, a segment shorter than ~2p ...) the pivot-patched solve means nothing, while numpy.linalg.lstsq
// drops the null directions and returns the shortest solution.  For the flagged elements only (every other workgroup
// exits at once): G and r are re-assembled from the lag record, G = V diag(lambda) V^T by cyclic Jacobi rotations in a
// round-robin order (p/2 disjoint rotations per round, so rows, then columns, of all pairs rotate in parallel), and
//   a = - V diag(1/lambda_k if lambda_k > cut else 0) V^T r,   cut = rel_cut * lambda_max.
// lstsq cuts SINGULAR values at eps*max(M,N)*sigma_max, i.e. lambda at ~1e-20 lambda_max -- below the rounding noise
// of a float64 Gram matrix (~p eps lambda_max): rel_cut (1e-12 by default) sits above that noise, so directions with
// sigma/sigma_max between 1e-10 and 1e-6 are dropped here and kept there.  That band is the cond(A) > 1e6 regime in
// which no normal-equation method holds 1e-4 anyway; exact rank deficiency is reproduced.
// A and V live in global scratch (2 p^2 doubles per element): this is a rare path, not a fast one.
// ------------------------------------------------------------------------------------------------------------
constexpr int MN_MAX_P = 512;
constexpr int MN_MAX_SWEEPS = 40;

__global__ __launch_bounds__(SV_THREADS) void ar_minnorm_kernel(const double* __restrict__ part,
                                                                const int32_t* __restrict__ nlen, int p,
                                                                double* __restrict__ scratch2,
                                                                double* __restrict__ coeffs, double* __restrict__ info,
                                                                int lag_nchunks_max, long long lag_rec_doubles,
                                                                double rel_cut) {
  __shared__ double vec[MN_MAX_P], yv[MN_MAX_P], rot_c[MN_MAX_P / 2], rot_s[MN_MAX_P / 2];
  __shared__ int rot_i[MN_MAX_P / 2], rot_j[MN_MAX_P / 2];
  __shared__ double red[SV_THREADS / 64];
  __shared__ double off_s, diag_s;
  __shared__ int bad_s;
  const int e = blockIdx.x, tid = threadIdx.x;
  if (info[IRA_AR_INFO_DOUBLES * e + 0] != 1.0) return;
  const long long N = nlen[e];
  double* A = scratch2 + (long long)e * 2 * p * p;
  double* V = A + (long long)p * p;
  // ---- G and r from the lag sums and the head/tail samples, exactly as ar_solve_kernel assembles them --------------------
  {
    const int nlag = p + 1;
    const double* rec = part + (long long)e * lag_rec_doubles;
    const double* head = rec + (long long)lag_nchunks_max * nlag;
    const double* tail = head + nlag;
    const int lchunks = lag_chunks(N, p);
    for (int d = tid; d < nlag; d += SV_THREADS) {
      double c = 0.0;
      for (int ch = 0; ch < lchunks; ++ch) c += rec[(long long)ch * nlag + d];
      if (d >= 1) vec[d - 1] = -c;
      double run = c, comp = 0.0;
      for (int a = 1; a + d <= p; ++a) {
        const int m = a - 1;
        const double t1 = head[p - 1 - m] * head[p - 1 - m - d];
        const double t2 = -tail[m] * tail[m + d];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const double term = q == 0 ? t1 : t2;
          const double t = run + term;
          comp += (fabs(run) >= fabs(term)) ? (run - t) + term : (term - t) + run;
          run = t;
        }
        const double v = run + comp;
        const int r = a + d - 1, cidx = a - 1;
        A[r * p + cidx] = v;
        A[cidx * p + r] = v;
      }
    }
  }
  for (int idx = tid; idx < p * p; idx += SV_THREADS) V[idx] = (idx / p == idx % p) ? 1.0 : 0.0;
  if (tid == 0) bad_s = 0;
  __syncthreads();
  {
    int bad = 0;
    for (int idx = tid; idx < p * p; idx += SV_THREADS) { const double v = A[idx]; if (!(v - v == 0.0)) bad = 1; }
    for (int j = tid; j < p; j += SV_THREADS) { const double v = vec[j]; if (!(v - v == 0.0)) bad = 1; }
    if (bad) bad_s = 1;                                             // benign race: every writer stores 1
  }
  __syncthreads();
  double* co = coeffs + (long long)e * (p + 1);
  if (bad_s) {
    // a NaN or infinite sample: the reference's lstsq raises LinAlgError ("SVD did not converge"); status 3, NaN coefficients
    const double qn = __longlong_as_double(0x7ff8000000000000ll);
    for (int j = tid; j < p; j += SV_THREADS) co[j + 1] = qn;
    if (tid == 0) { co[0] = 1.0; info[IRA_AR_INFO_DOUBLES * e + 0] = 3.0; }
    return;
  }
  // ---- cyclic Jacobi, round-robin pairing (circle method: player m-1 fixed, the others rotate) ------------------------------
  const int m = p + (p & 1);                                        // even number of players; index p (if any) is a bye
  const int npairs = m / 2;
  for (int sweep = 0; sweep < MN_MAX_SWEEPS; ++sweep) {
    double offp = 0.0, diagp = 0.0;
    for (int idx = tid; idx < p * p; idx += SV_THREADS) {
      const int r = idx / p, c = idx - r * p;
      const double v = A[idx];
      if (r == c) diagp += v * v; else offp += v * v;
    }
    offp = ira::wave_sum(offp); diagp = ira::wave_sum(diagp);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = offp;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < SV_THREADS / 64; ++w) t += red[w]; off_s = t; }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = diagp;
    __syncthreads();
    if (tid == 0) { double t = 0.0; for (int w = 0; w < SV_THREADS / 64; ++w) t += red[w]; diag_s = t; }
    __syncthreads();
    if (!(off_s > 1e-27 * diag_s) || !(off_s > 0.0)) break;        // off-diagonal norm below ~3e-14 of the diagonal's
    for (int round = 0; round < m - 1; ++round) {
      for (int k = tid; k < npairs; k += SV_THREADS) {
        int i = (k == 0) ? m - 1 : (round + k) % (m - 1);
        int j = (k == 0) ? round : (round - k + (m - 1)) % (m - 1);
        if (i > j) { const int t = i; i = j; j = t; }
        double c = 1.0, sn = 0.0;
        if (j < p) {
          const double aij = A[i * p + j];
          if (aij != 0.0) {
            const double tau = (A[j * p + j] - A[i * p + i]) / (2.0 * aij);
            const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            sn = t * c;
          }
        }
        rot_i[k] = i; rot_j[k] = j; rot_c[k] = c; rot_s[k] = sn;
      }
      __syncthreads();
      // rows: A <- J^T A
      for (int idx = tid; idx < npairs * p; idx += SV_THREADS) {
        const int k = idx / p, col = idx - k * p;
        const int i = rot_i[k], j = rot_j[k];
        if (j >= p || rot_s[k] == 0.0) continue;
        const double c = rot_c[k], sn = rot_s[k];
        const double ai = A[i * p + col], aj = A[j * p + col];
        A[i * p + col] = c * ai - sn * aj;
        A[j * p + col] = sn * ai + c * aj;
      }
      __syncthreads();
      // columns: A <- A J, V <- V J
      for (int idx = tid; idx < npairs * p; idx += SV_THREADS) {
        const int k = idx / p, row = idx - k * p;
        const int i = rot_i[k], j = rot_j[k];
        if (j >= p || rot_s[k] == 0.0) continue;
        const double c = rot_c[k], sn = rot_s[k];
        const double ai = A[row * p + i], aj = A[row * p + j];
        A[row * p + i] = c * ai - sn * aj;
        A[row * p + j] = sn * ai + c * aj;
        const double vi = V[row * p + i], vj = V[row * p + j];
        V[row * p + i] = c * vi - sn * vj;
        V[row * p + j] = sn * vi + c * vj;
      }
      __syncthreads();
    }
  }
  // ---- a = V diag(1/lambda | 0) V^T r -----------------------------------------------------------------------------------------
  double lmax = 0.0;
  for (int k = 0; k < p; ++k) lmax = fmax(lmax, A[k * p + k]);       // every thread: p global reads, rare path
  const double cut = rel_cut * lmax;
  for (int k = tid; k < p; k += SV_THREADS) {
    const double lam = A[k * p + k];
    double dot = 0.0;
    for (int r = 0; r < p; ++r) dot += V[r * p + k] * vec[r];
    yv[k] = (lam > cut && lam > 0.0) ? dot / lam : 0.0;
  }
  __syncthreads();
  for (int r = tid; r < p; r += SV_THREADS) {
    double acc = 0.0;
    for (int k = 0; k < p; ++k) acc += V[r * p + k] * yv[k];
    co[r + 1] = acc;
  }
  if (tid == 0) {
    co[0] = 1.0;
    int rank = 0;
    double lmin = INFINITY;
    for (int k = 0; k < p; ++k) { const double lam = A[k * p + k]; if (lam > cut && lam > 0.0) { ++rank; lmin = fmin(lmin, lam); } }
    info[IRA_AR_INFO_DOUBLES * e + 0] = 4.0;                         // minimum-norm solution over `rank` directions
    info[IRA_AR_INFO_DOUBLES * e + 1] = lmax;
    info[IRA_AR_INFO_DOUBLES * e + 2] = rank ? lmin : 0.0;
    info[IRA_AR_INFO_DOUBLES * e + 3] = (double)rank;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Aberth-Ehrlich: all roots of c[0] z^n + ... + c[n] (descending powers), float64 complex.
// One workgroup per polynomial, one thread per root (n <= 1024).  Jacobi-style sweeps (all corrections from
// the previous iterate) keep it deterministic.  Leading/trailing handling follows the reference: trailing
// |c| < trail_eps coefficients are dropped first (zplane.py:153-155), then numpy.roots semantics (leading
// zeros stripped; exact trailing zeros become roots at 0).
// ------------------------------------------------------------------------------------------------------------
constexpr int RT_MAX_DEG = 1024;
constexpr int RT_MAX_ITERS = 200;

struct cdbl { double re, im; };
__device__ __forceinline__ cdbl c_mul(cdbl a, cdbl b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cdbl c_div(cdbl a, cdbl b) {
  // Smith's algorithm
  if (fabs(b.re) >= fabs(b.im)) {
    const double r = b.im / b.re, d = b.re + b.im * r;
    return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
  }
  const double r = b.re / b.im, d = b.re * r + b.im;
  return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}

// LPR lanes per root (round 4): the sum over the other roots, 1 / (z_k - z_j) -- a float64 reciprocal per pair, four fifths
// of an iteration -- is dealt to LPR neighbouring lanes and combined with two DPP shuffles; the Horner chain (serial by
// nature) is evaluated by every lane of the group.  One lane per root kept a 64-root polynomial on ONE wave for ~20
// iterations of ~14 k cycles each: 0.15-0.18 ms per 256 polynomials at a quarter of the SIMDs.
template <int LPR>
__global__ void poly_roots_kernel(const double* __restrict__ coeffs, int stride, int ncoef, double trail_eps,
                                  double* __restrict__ roots, int32_t* __restrict__ nroots_out) {
  __shared__ double c[RT_MAX_DEG + 1];
  __shared__ cdbl z[2][RT_MAX_DEG];
  __shared__ int sh_lo, sh_hi, sh_changed;
  const int e = blockIdx.x, nt = blockDim.x / LPR;
  const int tid = threadIdx.x / LPR, part = threadIdx.x % LPR;     // root slot and the lane's share of the pair sum
  const double* ce = coeffs + (long long)e * stride;
  double* re = roots + (long long)e * 2 * (ncoef - 1);
  if (threadIdx.x == 0) {
    int hi = ncoef;   // exclusive end after trimming tiny trailing coefficients
    while (hi > 1 && fabs(ce[hi - 1]) < trail_eps) --hi;
    int lo = 0;       // strip leading exact zeros (numpy.roots)
    while (lo < hi && ce[lo] == 0.0) ++lo;
    sh_lo = lo; sh_hi = hi;
  }
  __syncthreads();
  int lo = sh_lo, hi = sh_hi;
  if (hi - lo <= 1 || hi <= 1) {
    if (threadIdx.x == 0) nroots_out[e] = 0;
    return;
  }
  // exact trailing zeros -> roots at the origin
  int tz = 0;
  while (hi - 1 - tz > lo && ce[hi - 1 - tz] == 0.0) ++tz;
  const int n = hi - lo - 1 - tz;   // degree of the deflated polynomial
  const double lead = ce[lo];
  for (int k = threadIdx.x; k <= n; k += blockDim.x) c[k] = ce[lo + k] / lead;   // monic
  __syncthreads();
  if (n >= 1) {
    // initial radius: geometric mean of the root moduli, |c_n|^(1/n), kept in a sane range
    double r0 = pow(fabs(c[n]), 1.0 / (double)n);
    if (!(r0 > 1e-3)) r0 = 1e-3;
    if (r0 > 1e3) r0 = 1e3;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
      double s, co;
      sincos(2.0 * 3.14159265358979323846 * (double)k / (double)n + 0.4, &s, &co);
      z[0][k] = {r0 * co, r0 * s};
    }
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < RT_MAX_ITERS; ++it) {
      if (threadIdx.x == 0) sh_changed = 0;
      __syncthreads();
      int changed = 0;
      // (every lane of a group runs the same trip count: the shuffles below need the whole group)
      for (int k0 = tid; k0 < ((n + nt - 1) / nt) * nt; k0 += nt) {
        const bool live = k0 < n;
        const int k = live ? k0 : n - 1;
        const cdbl zk = z[cur][k];
        // Horner for p and p'
        cdbl pv = {1.0, 0.0}, dv = {0.0, 0.0};
        for (int m = 1; m <= n; ++m) {
          dv = c_mul(dv, zk); dv.re += pv.re; dv.im += pv.im;
          pv = c_mul(pv, zk); pv.re += c[m];
        }
        cdbl w = {0.0, 0.0};
        if (pv.re != 0.0 || pv.im != 0.0) {
          const cdbl ratio = c_div(pv, dv);
          cdbl sum = {0.0, 0.0};
          for (int j = part; j < n; j += LPR) {
            if (j == k) continue;
            const cdbl zj = z[cur][j];
            const double dr = zk.re - zj.re, di = zk.im - zj.im;
            const double inv = 1.0 / (dr * dr + di * di);   // 1/(zk - zj) = conj(d)/|d|^2
            sum.re += dr * inv; sum.im -= di * inv;
          }
          if (LPR > 1) {
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) {                // the group's lanes are neighbours: xor 1, xor 2
              sum.re += __shfl_xor(sum.re, o, 64);
              sum.im += __shfl_xor(sum.im, o, 64);
            }
          }
          const cdbl rs = c_mul(ratio, sum);
          w = c_div(ratio, cdbl{1.0 - rs.re, -rs.im});
        }
        if (live && part == 0) z[cur ^ 1][k] = {zk.re - w.re, zk.im - w.im};
        const double wm = fabs(w.re) + fabs(w.im), zm = fabs(zk.re) + fabs(zk.im);
        if (live && wm > 2e-13 * zm) changed = 1;
      }
      if (changed) sh_changed = 1;   // benign race: every writer stores 1
      __syncthreads();
      cur ^= 1;
      const int any = sh_changed;
      __syncthreads();
      if (!any) break;
    }
    for (int k = threadIdx.x; k < n; k += blockDim.x) { re[2 * k] = z[cur][k].re; re[2 * k + 1] = z[cur][k].im; }
  }
  for (int k = n + threadIdx.x; k < n + tz; k += blockDim.x) { re[2 * k] = 0.0; re[2 * k + 1] = 0.0; }
  if (threadIdx.x == 0) nroots_out[e] = n + tz;
}

// b[n] = sum_{k=0..p} a[k] h[n-k], 0 <= n-k < N; h = x / divisor (float64)
__global__ void fir_numerator_kernel(const double* __restrict__ coeffs, int p, const float* __restrict__ x,
                                     const int64_t* __restrict__ xoff, const int32_t* __restrict__ nlen,
                                     const double* __restrict__ divisor, int q, double* __restrict__ b) {
  const int e = blockIdx.x;
  const double* a = coeffs + (long long)e * (p + 1);
  const float* xs = x + xoff[e];
  const double div = divisor ? divisor[e] : 1.0;
  const long long N = nlen[e];
  for (int n = threadIdx.x; n <= q; n += blockDim.x) {
    double acc = 0.0;
    for (int k = 0; k <= p; ++k) {
      const long long m = (long long)n - k;
      if (m < 0 || m >= N) continue;
      acc += a[k] * ((double)xs[m] / div);
    }
    b[(long long)e * (q + 1) + n] = acc;
  }
}

}  // namespace

extern "C" int64_t ira_ar_partial_doubles(int32_t p, int32_t max_len) {
  if (p < 1 || p > GR_MAX_P || max_len <= p) return 0;
  const int64_t nchunks = ((int64_t)max_len - p + GR_CHUNK - 1) / GR_CHUNK;
  const int64_t dense = nchunks * groups_total(p) * (int64_t)GR_PART;       // MFMA Gram partials (IRA_AR_DENSE)
  const int64_t lag = lag_record_doubles(max_len, p);                        // lag sums + head/tail samples
  return dense > lag ? dense : lag;
}

// flags & IRA_AR_DENSE_GRAM selects the dense MFMA Gram (cross-check / A-B); gram, solve and refine of one fit must be given
// the same flags: they agree on the layout of the partial record through it.
static inline bool ar_dense(int32_t flags) { return (flags & IRA_AR_DENSE_GRAM) != 0; }
// one wave per element (ar_solve_wave_kernel): lag-sum record, order <= 64, unless the caller asks for the workgroup kernel
static inline bool ar_wave_solve(int32_t flags, int order) {
  return !ar_dense(flags) && order <= 64 && (flags & IRA_AR_WORKGROUP_SOLVE) == 0;
}

static int32_t ar_check(int32_t nb, int32_t max_len, int32_t order) {
  if (nb < 0) return IRA_E_SIZE;
  if (order < 1 || order > GR_MAX_P || max_len <= order) return IRA_E_SIZE;
  if (nb > 65535 || groups_total(order) > 65535) return IRA_E_SIZE;
  return IRA_OK;
}

extern "C" int32_t ira_ar_gram(const float* x_dev, const double* x64_dev, const int64_t* xoff_dev,
                               const int32_t* len_dev, const double* divisor_dev, int32_t nb, int32_t max_len,
                               int32_t order, double* partial_dev, int32_t flags, void* stream) {
  if (x_dev == nullptr && x64_dev == nullptr) return IRA_E_NULL;
  IRA_CHECK_PTR(xoff_dev); IRA_CHECK_PTR(len_dev); IRA_CHECK_PTR(partial_dev);
  const int32_t rc = ar_check(nb, max_len, order);
  if (rc != IRA_OK || nb == 0) return rc;
  if (!ar_dense(flags)) {
    const int lchunks = lag_chunks(max_len, order);
    const int lpt = lag_per_thread(order + 1);
    const int ngroups = (order + 1 + lpt - 1) / lpt;
    const size_t lds = sizeof(double) * ((size_t)lpt * ngroups + LAG_CHUNK);
    auto launch = [&](auto kernel) -> int32_t {
      if (lds > 64 * 1024) {
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (er != hipSuccess) return ira_hip_status(er);
      }
      kernel<<<dim3(lchunks, 1, nb), LAG_THREADS, lds, (hipStream_t)stream>>>(
          x64_dev ? nullptr : x_dev, x64_dev, xoff_dev, len_dev, divisor_dev, order, lchunks,
          lag_record_doubles(max_len, order), partial_dev);
      return IRA_OK;
    };
    int32_t lrc;
    switch (lpt) {
      case 16: lrc = launch(&ar_lag_kernel<16>); break;
      case 13: lrc = launch(&ar_lag_kernel<13>); break;
      case 12: lrc = launch(&ar_lag_kernel<12>); break;
      case 10: lrc = launch(&ar_lag_kernel<10>); break;
      default: lrc = launch(&ar_lag_kernel<8>); break;
    }
    if (lrc != IRA_OK) return lrc;
    IRA_RETURN_LAUNCH();
  }
  const int nchunks = (int)(((int64_t)max_len - order + GR_CHUNK - 1) / GR_CHUNK);
  const int gs = groups_side(order);
  ar_gram_kernel<true><<<dim3(nchunks, gs, nb), 64, 0, (hipStream_t)stream>>>(
      x64_dev ? nullptr : x_dev, x64_dev, xoff_dev, len_dev, divisor_dev, order, nchunks, partial_dev);
  if (gs > 1)
    ar_gram_kernel<false><<<dim3(nchunks, gs * (gs - 1) / 2, nb), 64, 0, (hipStream_t)stream>>>(
        x64_dev ? nullptr : x_dev, x64_dev, xoff_dev, len_dev, divisor_dev, order, nchunks, partial_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_ar_solve(const double* partial_dev, const int32_t* len_dev, int32_t nb, int32_t max_len,
                                int32_t order, double ridge, double* gscratch_dev, double* coeffs_dev,
                                double* info_dev, int32_t flags, void* stream) {
  IRA_CHECK_PTR(partial_dev); IRA_CHECK_PTR(len_dev); IRA_CHECK_PTR(coeffs_dev);
  const int32_t rc = ar_check(nb, max_len, order);
  if (rc != IRA_OK || nb == 0) return rc;
  if (order > SV_LDS_P && gscratch_dev == nullptr) return IRA_E_NULL;
  const int nchunks = (int)(((int64_t)max_len - order + GR_CHUNK - 1) / GR_CHUNK);
  size_t lds = sizeof(double) * (size_t)order;
  if (order <= SV_LDS_P) lds += sizeof(double) * (size_t)order * order;
  else lds += sizeof(double) * sv_blocked_lds_doubles(order);        // panel + diagonal block of the blocked factorisation
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ar_solve_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return ira_hip_status(e);
  }
  if (ar_wave_solve(flags, order)) {
    ar_solve_wave_kernel<<<nb, 64, 0, (hipStream_t)stream>>>(partial_dev, len_dev, order, ridge, coeffs_dev, info_dev,
                                                              lag_chunks(max_len, order), lag_record_doubles(max_len, order),
                                                              nullptr, 0.0);
    IRA_RETURN_LAUNCH();
  }
  ar_solve_kernel<<<nb, SV_THREADS, lds, (hipStream_t)stream>>>(partial_dev, len_dev, order, nchunks, ridge,
                                                                 gscratch_dev, coeffs_dev, info_dev,
                                                                 ar_dense(flags) ? 0 : 1, lag_chunks(max_len, order),
                                                                 lag_record_doubles(max_len, order), nullptr, 0.0);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_ar_minnorm(const double* partial_dev, const int32_t* len_dev, int32_t nb, int32_t max_len,
                                  int32_t order, double* scratch2_dev, double* coeffs_dev, double* info_dev,
                                  double rel_cut, void* stream) {
  IRA_CHECK_PTR(partial_dev); IRA_CHECK_PTR(len_dev); IRA_CHECK_PTR(scratch2_dev); IRA_CHECK_PTR(coeffs_dev);
  IRA_CHECK_PTR(info_dev);
  const int32_t rc = ar_check(nb, max_len, order);
  if (rc != IRA_OK || nb == 0) return rc;
  if (order > MN_MAX_P || !(rel_cut > 0.0) || !(rel_cut < 1.0)) return IRA_E_SIZE;
  ar_minnorm_kernel<<<nb, SV_THREADS, 0, (hipStream_t)stream>>>(partial_dev, len_dev, order, scratch2_dev, coeffs_dev,
                                                                 info_dev, lag_chunks(max_len, order),
                                                                 lag_record_doubles(max_len, order), rel_cut);
  IRA_RETURN_LAUNCH();
}


extern "C" int64_t ira_ar_exact_doubles(int32_t order, int32_t max_len, int32_t which) {
  if (order < 1 || order > GR_MAX_P || max_len <= order) return 0;
  if (which == 0) return 2ll * lag_chunks(max_len, order) * (order + 1);          // double-double lag-sum partials
  return 2ll * ((int64_t)order * order + order);                                    // Gram matrix + vector
}

extern "C" int32_t ira_ar_exact(const float* x_dev, const double* x64_dev, const int64_t* xoff_dev, const int32_t* len_dev,
                                const double* divisor_dev, int32_t nb, int32_t max_len, int32_t order, double ridge,
                                const double* partial_dev, double* ddpartial_dev, double* ddscratch_dev,
                                double* coeffs_dev, double* info_dev, double cond_threshold, void* stream) {
  if (x_dev == nullptr && x64_dev == nullptr) return IRA_E_NULL;
  IRA_CHECK_PTR(xoff_dev); IRA_CHECK_PTR(len_dev); IRA_CHECK_PTR(partial_dev); IRA_CHECK_PTR(ddpartial_dev);
  IRA_CHECK_PTR(ddscratch_dev); IRA_CHECK_PTR(coeffs_dev); IRA_CHECK_PTR(info_dev);
  const int32_t rc = ar_check(nb, max_len, order);
  if (rc != IRA_OK || nb == 0) return rc;
  if (!(cond_threshold >= 1.0) || !(ridge >= 0.0)) return IRA_E_SIZE;
  const int lchunks = lag_chunks(max_len, order);
  const size_t lds = sizeof(double) * ((size_t)order + LAG_CHUNK);
  if (lds > 64 * 1024) return IRA_E_SIZE;
  hipStream_t st = (hipStream_t)stream;
  ar_lag_dd_kernel<<<dim3(lchunks, 1, nb), LAG_THREADS, lds, st>>>(x64_dev ? nullptr : x_dev, x64_dev, xoff_dev, len_dev,
                                                                   divisor_dev, order, lchunks, info_dev, cond_threshold,
                                                                   ddpartial_dev);
  ar_solve_dd_kernel<<<nb, SV_THREADS, 0, st>>>(partial_dev, ddpartial_dev, len_dev, order, ridge, ddscratch_dev, coeffs_dev,
                                                info_dev, lchunks, lag_record_doubles(max_len, order), cond_threshold);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_ar_refine(const float* x_dev, const double* x64_dev, const int64_t* xoff_dev,
                                 const int32_t* len_dev, const double* divisor_dev, int32_t nb, int32_t max_len,
                                 int32_t order, const double* partial_dev, double* gscratch_dev, double* coeffs_dev,
                                 double* info_dev, double* grad_dev, double cond_threshold, int32_t steps,
                                 int32_t flags, void* stream) {
  if (x_dev == nullptr && x64_dev == nullptr) return IRA_E_NULL;
  IRA_CHECK_PTR(xoff_dev); IRA_CHECK_PTR(len_dev); IRA_CHECK_PTR(partial_dev); IRA_CHECK_PTR(coeffs_dev);
  IRA_CHECK_PTR(info_dev); IRA_CHECK_PTR(grad_dev);
  const int32_t rc = ar_check(nb, max_len, order);
  if (rc != IRA_OK || nb == 0 || steps <= 0) return rc;
  if (steps > 4 || !(cond_threshold >= 1.0)) return IRA_E_SIZE;
  if (order > SV_LDS_P && gscratch_dev == nullptr) return IRA_E_NULL;
  const int lchunks = lag_chunks(max_len, order);
  const int nlag = order + 1, ngroups = (nlag + 3) / 4;
  const int nsub = ngroups >= LAG_THREADS ? 1 : LAG_THREADS / ngroups;
  const size_t lds_g = sizeof(double) * ((size_t)(order + 3) + 2 * LAG_CHUNK + nlag + (size_t)nsub * 4 * ngroups);
  if (lds_g > 64 * 1024) {
    hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(&ar_grad_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g);
    if (er != hipSuccess) return ira_hip_status(er);
  }
  const int nchunks = (int)(((int64_t)max_len - order + GR_CHUNK - 1) / GR_CHUNK);
  size_t lds_s = sizeof(double) * (size_t)order;
  if (order <= SV_LDS_P) lds_s += sizeof(double) * (size_t)order * order;
  else lds_s += sizeof(double) * sv_blocked_lds_doubles(order);
  if (lds_s > 64 * 1024) {
    hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(&ar_solve_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
    if (er != hipSuccess) return ira_hip_status(er);
  }
  hipStream_t st = (hipStream_t)stream;
  for (int it = 0; it < steps; ++it) {
    ar_grad_kernel<<<dim3(lchunks, 1, nb), LAG_THREADS, lds_g, st>>>(x64_dev ? nullptr : x_dev, x64_dev, xoff_dev, len_dev,
                                                                      divisor_dev, order, lchunks, coeffs_dev, info_dev,
                                                                      cond_threshold, grad_dev);
    if (ar_wave_solve(flags, order))
      ar_solve_wave_kernel<<<nb, 64, 0, st>>>(partial_dev, len_dev, order, 0.0, coeffs_dev, info_dev, lchunks,
                                              lag_record_doubles(max_len, order), grad_dev, cond_threshold);
    else
      ar_solve_kernel<<<nb, SV_THREADS, lds_s, st>>>(partial_dev, len_dev, order, nchunks, 0.0, gscratch_dev, coeffs_dev,
                                                      info_dev, ar_dense(flags) ? 0 : 1, lchunks,
                                                      lag_record_doubles(max_len, order), grad_dev, cond_threshold);
  }
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_ar_fit(const float* x_dev, const double* x64_dev, const int64_t* xoff_dev,
                              const int32_t* len_dev, const double* divisor_dev, int32_t nb, int32_t max_len,
                              int32_t order, double ridge, double* partial_dev, double* gscratch_dev,
                              double* coeffs_dev, double* info_dev, int32_t flags, void* stream) {
  int32_t rc = ira_ar_gram(x_dev, x64_dev, xoff_dev, len_dev, divisor_dev, nb, max_len, order, partial_dev, flags, stream);
  if (rc != IRA_OK) return rc;
  return ira_ar_solve(partial_dev, len_dev, nb, max_len, order, ridge, gscratch_dev, coeffs_dev, info_dev, flags, stream);
}

extern "C" int32_t ira_poly_roots(const double* coeffs_dev, int32_t npoly, int32_t ncoef, double trail_eps,
                                  double* roots_dev, int32_t* nroots_dev, void* stream) {
  IRA_CHECK_PTR(coeffs_dev); IRA_CHECK_PTR(roots_dev); IRA_CHECK_PTR(nroots_dev);
  if (npoly <= 0) return npoly == 0 ? IRA_OK : IRA_E_SIZE;
  if (ncoef < 2 || ncoef - 1 > RT_MAX_DEG) return IRA_E_SIZE;
  int threads = 64;
  while (threads < ncoef - 1 && threads < 1024) threads <<= 1;
  if (threads <= 256) {                                      // four lanes per root while a workgroup can hold them
    poly_roots_kernel<4><<<npoly, 4 * threads, 0, (hipStream_t)stream>>>(coeffs_dev, ncoef, ncoef, trail_eps, roots_dev,
                                                                        nroots_dev);
    IRA_RETURN_LAUNCH();
  }
  poly_roots_kernel<1><<<npoly, threads, 0, (hipStream_t)stream>>>(coeffs_dev, ncoef, ncoef, trail_eps, roots_dev,
                                                               nroots_dev);
  IRA_RETURN_LAUNCH();
}

extern "C" int32_t ira_fir_numerator(const double* coeffs_dev, int32_t order, const float* x_dev,
                                     const int64_t* xoff_dev, const int32_t* len_dev, const double* divisor_dev,
                                     int32_t nb, int32_t zero_order, double* b_dev, void* stream) {
  IRA_CHECK_PTR(coeffs_dev); IRA_CHECK_PTR(x_dev); IRA_CHECK_PTR(xoff_dev); IRA_CHECK_PTR(len_dev);
  IRA_CHECK_PTR(b_dev);
  if (nb <= 0) return nb == 0 ? IRA_OK : IRA_E_SIZE;
  if (order < 0 || zero_order < 0) return IRA_E_SIZE;
  fir_numerator_kernel<<<nb, 128, 0, (hipStream_t)stream>>>(coeffs_dev, order, x_dev, xoff_dev, len_dev,
                                                            divisor_dev, zero_order, b_dev);
  IRA_RETURN_LAUNCH();
}
