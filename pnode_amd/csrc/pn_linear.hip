// pnode_amd -- the parameter sensitivities of a Linear layer, fused: dW and db of `out = x W^T + b` from the cotangent G at the
// layer's output, accumulated over the stages and time steps of a reverse sweep (row a-9 of the hot path:
// RHSJacPShell.multTranspose + the VecAXPY on mu inside TSAdjointStep_RK, reference pnode/petsc_adjoint.py:341-363).
//
//   PW[s][m][n] += sum_{k in K-range s} (alpha G[k][m]) X[k][n]        s = 0..7   (G: rows x out, X: rows x in, row-major)
//   PB[s][j][m] += sum_{k in the slabs of K-range s that tile column j adds up}  alpha G[k][m]
//
// and, once per reverse sweep,  mu_W[m][n] += sum_s PW[s][m][n],  mu_b[m] += sum_{s,j} PB[s][j][m]  (then PW = PB = 0).
//
// Why a kernel of its own: the product is K-deep (K = rows = 4096 at BASELINE's target configuration, 512 x 512 out) -- the
// one GEMM shape of the time step the BLAS library serves badly (fp32: 25.6 us against 19.5 for the forward- and dX-shaped
// products of the same size; fp64: 129 us against 40) -- and everything around it was extra passes: the column sum for db, a copy
// of the last layer's cotangent, the accumulation into mu.  Here:
//   * K is split eight ways and the split index is blockIdx % 8: workgroups are dealt to the 8 XCDs round-robin, so each XCD
//     works on ONE K range and the G and X rows it needs stay in its own 4 MB L2;
//   * 64 x 64 output tiles per workgroup of eight waves; each wave owns a 32 x 32 tile (fp32: one v_mfma_f32_32x32x2_f32
//     accumulator, exact fp32, a k-ordered fmaf chain; fp64: 2 x 2 accumulators of v_mfma_f64_16x16x4_f64) and HALF of every K
//     slab -- waves 0-3 the first half of its rows, waves 4-7 the second, added through LDS at the end;
//   * slabs of 32 rows per operand (8 KB fp32, 16 KB fp64) through TWO LDS buffers, one barrier per slab: while slab s is
//     multiplied, slab s+1 goes from registers to the other buffer and the global loads of slab s+2 are in flight
//     (round 5 had one buffer and two barriers per slab: 41 500 cycles per tile against 39 500, tools/mb_wgrad6.hip);
//   * the partial tile in PW is read under the K loop and added at its end: the sum over stages and time steps costs no pass;
//   * the workgroups of a tile row share the column sums of their G slabs between them (slab s: tile column s % ntn): db for free;
//   * GROUPED launches (pn_linear_wgrad_group): the pairs of all Linear layers of one stage VJP go through ONE launch --
//     workgroups of the next pair start while the last ones of the previous pair drain, and there is one launch boundary per
//     stage instead of one per layer: 20.1 us per 4096 x 512 x 512 pair against 22.6 alone (and 25 inside the sweep in round 5).
// Bit-reproducible (fixed split, fixed order, no atomics); independent of how pairs are grouped into launches.
// rows >= 256 (any number: a ragged last slab is zero-filled), out % 64 == 0, in % 64 == 0, out * in <= 2^22; everything else takes the general path (torch GEMM +
// pn_colsum_accum_multi).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <string>
#include <type_traits>

#include "pn_internal.h"
#include "pnode_amd.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int kSplit = 8;        // K ranges = XCDs
constexpr int BM = 64, BN = 64;
constexpr int kThreads = 512;    // eight waves: 2 x 2 tiles of 32 x 32, times the two halves of a K slab
constexpr int kMaxPairs = PN_WGRAD_MAX_PAIRS;
// Above 2048 x 2048 weights a layer has tiles enough to fill the chip without a K split and the partial buffers (8 x the weight)
// stop being small change: such layers are left to the library GEMM.
constexpr int64_t kMaxWeights = (int64_t)1 << 22;

template <typename T>
struct Shape {                   // one slab = 32 rows of G and of X in either precision (fp32: 8 KB per operand, fp64: 16 KB)
  static constexpr int BK = 32;
  static constexpr int VEC = 16 / (int)sizeof(T);
  static constexpr int VPT = BK * BM / VEC / kThreads;          // 16-byte vectors per thread per operand: 1 (fp32), 2 (fp64)
  static constexpr int LD = BM;                                 // unpadded rows
  // fp32: a half-wave of ds_read_b32 reads 32 consecutive floats of one row: conflict-free as it stands.  fp64: a half-wave of
  // ds_read_b64 reads 16 doubles of each of TWO consecutive rows, which start 128 dwords = 0 banks apart: odd rows are stored with
  // their two halves of 16 doubles swapped pairwise (column ^ 16), so that the two rows sit 32 banks apart
  static __device__ __forceinline__ int col(int row, int c) { return std::is_same<T, float>::value ? c : (c ^ ((row & 1) << 4)); }
};

struct GroupArgs {
  const void *g[kMaxPairs], *x[kMaxPairs];
  void *pw[kMaxPairs];
  double *pb[kMaxPairs];
  double alpha[kMaxPairs];
  int M[kMaxPairs], N[kMaxPairs];
  int first[kMaxPairs + 1];      // first workgroup of pair p; first[npairs] = grid size
  int npairs, K;
};

// RAGGED: rows is not a multiple of 8 slabs -- rows that do not exist are loaded as zeros (an instantiation of its own: the test in
// front of every load costs the aligned case 2-17 %)
template <typename T, bool RAGGED>
__device__ __forceinline__ void wgrad_body(const GroupArgs &ga_) {
  using S = Shape<T>;
  constexpr bool F32 = std::is_same<T, float>::value;
  constexpr int BK = S::BK, LD = S::LD, VEC = S::VEC;
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  constexpr int SLAB = BK * LD;                                  // elements per operand per buffer
  __shared__ T smem[2 * 2 * SLAB];                               // [buffer][G | X][BK][LD]; staging for the two reductions at the end
  int p = 0;
#pragma unroll
  for (int q = 1; q < kMaxPairs; ++q) p += (q < ga_.npairs && (int)blockIdx.x >= ga_.first[q]) ? 1 : 0;
  const T *__restrict__ G = static_cast<const T *>(ga_.g[p]);
  const T *__restrict__ X = static_cast<const T *>(ga_.x[p]);
  T *__restrict__ PW = static_cast<T *>(ga_.pw[p]);
  double *__restrict__ PB = ga_.pb[p];
  const int M = ga_.M[p], N = ga_.N[p], K = ga_.K;
  const T alpha = (T)ga_.alpha[p];
  const int bid = (int)blockIdx.x - ga_.first[p];
  const int split = bid % kSplit, tile = bid / kSplit;
  const int ntn = N / BN;
  const int tm = tile / ntn, tn = tile % ntn;
  // rows per K range: whole slabs; rows that do not exist (K is not a multiple of 8 slabs) are loaded as zeros
  const int kper = RAGGED ? (K + kSplit * BK - 1) / (kSplit * BK) * BK : K / kSplit, k0 = split * kper, nslab = kper / BK;
  constexpr bool ragged = RAGGED;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int kh = w >> 2, wm = w & 1, wn = (w >> 1) & 1;          // K half, tile row, tile column of this wave
  constexpr int GROW = BM / VEC, VPT = S::VPT;                   // 16-byte vectors per slab row; VPT G and VPT X vectors per thread
  static_assert(VPT * kThreads * VEC == BK * BM && kThreads % GROW == 0, "whole vectors per thread, every thread in one column group");
  const int lrow = t / GROW, lc = t % GROW;                      // vector v of this thread: row lrow + v * (kThreads / GROW), column group lc
  constexpr int RSTEP = kThreads / GROW;
  const bool bias = PB != nullptr;
  vec_t gv[VPT], xv[VPT];
  // this thread's columns of the G slabs that are this workgroup's to add up: K / 8 / BK / ntn values each, summed in the state's
  // precision (packed adds; four double adds per thread per slab in front of the barrier cost 1 us per 4096 x 512 x 512 pair), the
  // threads of a column group then in double
  T colsum[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) colsum[e] = (T)0;

  // element e of this lane's accumulator registers -> (row, column) of the wave's 32 x 32 tile
  auto tile_row = [&](int e) { return F32 ? (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) : ((e >> 3) * 16 + (lane >> 4) + 4 * (e & 3)); };
  auto tile_col = [&](int e) { return F32 ? (lane & 31) : (((e >> 2) & 1) * 16 + (lane & 15)); };
  // (fp64: e = 8 i + 4 j + r for accumulator (i, j), register r: C/D of v_mfma_f64_16x16x4_f64 is col = lane & 15, row = (lane >> 4) + 4 r)

  T *pw = PW + (size_t)split * M * N;
  T acc[16], old[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = (T)0;
  double *pbp = bias && t < BM ? PB + ((size_t)split * ntn + tn) * M + tm * BM + t : nullptr;

  auto gload = [&](int slab) {
    const int kb = k0 + slab * BK;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int row = kb + lrow + v * RSTEP;
      if (!ragged || row < K) {
        gv[v] = *reinterpret_cast<const vec_t *>(G + (size_t)row * M + tm * BM + lc * VEC);
        xv[v] = *reinterpret_cast<const vec_t *>(X + (size_t)row * N + tn * BN + lc * VEC);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) gv[v][e] = (T)0, xv[v][e] = (T)0;
      }
    }
  };
  // The workgroups of a tile row (tn = 0..ntn-1) see the same G slabs: slab s is added up by the one with tn == s % ntn,
  // so that no workgroup carries the column sums alone (the launch ends with its slowest workgroup).
  auto lstore = [&](int slab, int buf) {
    T *Gs = smem + buf * 2 * SLAB, *Xs = Gs + SLAB;
    if constexpr (VPT == 1) {                    // (fp32: one vector per thread, no swizzle)
      const vec_t g = alpha * gv[0];             // here, not at the load: the product would wait for the load in front of the MFMAs
      *reinterpret_cast<vec_t *>(&Gs[lrow * LD + lc * VEC]) = g;
      if (bias && slab % ntn == tn) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) colsum[e] += g[e];
      }
      *reinterpret_cast<vec_t *>(&Xs[lrow * LD + lc * VEC]) = xv[0];
    } else {
#pragma unroll
      for (int v = 0; v < VPT; ++v) {
        const int row = lrow + v * RSTEP;
        const vec_t g = alpha * gv[v];
        *reinterpret_cast<vec_t *>(&Gs[row * LD + S::col(row, lc * VEC)]) = g;
        if (bias && slab % ntn == tn) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) colsum[e] += g[e];
        }
        *reinterpret_cast<vec_t *>(&Xs[row * LD + S::col(row, lc * VEC)]) = xv[v];
      }
    }
  };
  auto compute = [&](int buf) {
    const T *Gs = smem + buf * 2 * SLAB, *Xs = Gs + SLAB;
    if constexpr (F32) {
      const int lr = lane & 31, lh = lane >> 5;
      f32x16 c;
#pragma unroll
      for (int e = 0; e < 16; ++e) c[e] = acc[e];
#pragma unroll
      for (int kk = 0; kk < BK / 2; kk += 2) {
        const float a = Gs[(kh * (BK / 2) + kk + lh) * LD + wm * 32 + lr];
        const float b = Xs[(kh * (BK / 2) + kk + lh) * LD + wn * 32 + lr];
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = c[e];
    } else {
      const int lr = lane & 15, lq = lane >> 4;
      f64x4 c[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) c[i][j][r] = acc[8 * i + 4 * j + r];
#pragma unroll
      for (int kk = 0; kk < BK / 2; kk += 4) {
        double a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = Gs[(kh * (BK / 2) + kk + lq) * LD + S::col(kk + lq, wm * 32 + i * 16 + lr)];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Xs[(kh * (BK / 2) + kk + lq) * LD + S::col(kk + lq, wn * 32 + j * 16 + lr)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], c[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[8 * i + 4 * j + r] = c[i][j][r];
    }
  };

  gload(0);
  // what the earlier stages / time steps left in PW (and PB): needed after the K loop, loaded now -- BEHIND the first slab's loads:
  // loads return in order, and the first LDS store must not wait for a tile that comes from HBM
#pragma unroll
  for (int e = 0; e < 16; ++e)
    old[e] = kh == 0 ? pw[(size_t)(tm * BM + wm * 32 + tile_row(e)) * N + tn * BN + wn * 32 + tile_col(e)] : (T)0;
  const double pbold = pbp ? *pbp : 0.0;
  lstore(0, 0);
  if (nslab > 1) gload(1);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    compute(s & 1);
    if (s + 1 < nslab) lstore(s + 1, (s + 1) & 1);      // the other buffer: its last readers passed the barrier of slab s-1
    if (s + 2 < nslab) gload(s + 2);                    // in flight while slab s+1 is multiplied
    __syncthreads();
  }
  // ---- the end, ONE barrier (the slab buffers are free now): the second K half hands its tile to the first through
  // [4 waves x 16 registers][64 lanes]; behind it [kThreads / GROW][BM] doubles: the threads of a column group, added in thread order
  T(*red)[64] = reinterpret_cast<T(*)[64]>(&smem[0]);
  double(*cs)[BM] = reinterpret_cast<double(*)[BM]>(&smem[4 * 16 * 64]);
  static_assert(sizeof(smem) >= 4 * 16 * 64 * sizeof(T) + (kThreads / GROW) * BM * sizeof(double), "the reductions are staged in the slab buffers");
  if (kh == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(w & 3) * 16 + e][lane] = acc[e];
  }
  if (bias) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) cs[lrow][lc * VEC + e] = (double)colsum[e];
  }
  __syncthreads();
  if (kh == 0) {                                      // PW = PW + (first + second)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      pw[(size_t)(tm * BM + wm * 32 + tile_row(e)) * N + tn * BN + wn * 32 + tile_col(e)] = old[e] + (acc[e] + red[(w & 3) * 16 + e][lane]);
  }
  if (pbp) {
    double sum = cs[0][t];
#pragma unroll
    for (int j = 1; j < kThreads / GROW; ++j) sum += cs[j][t];
    *pbp = pbold + sum;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The fp32 product on the BF16 matrix cores, by exact operand splitting (the default for fp32 states).
// Every fp32 operand a is split exactly into three bf16 terms a = a_hi + a_mid + a_lo (the top 16 bits of an fp32 pattern ARE a bf16:
// three truncations cover the 24-bit significand, every remainder is exact in fp32); a*b is then the sum of nine bf16 x bf16
// products, each of them exact in fp32.  The six largest -- hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi; what is dropped is below
// 2^-23 |a b| -- go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 x 32 cycles of matrix pipe per 32 x 32 x 16 block
// against 8 x 64 for v_mfma_f32_32x32x2_f32 (the bf16 pipe is 16 times as fast per product).  Measured against float64 on
// cotangent-like operands (tools/mb_wgrad_bf16x3.hip, error / sum |g x|): max 5.7e-8, rms 1.1e-8 -- BELOW the plain fp32 fmaf chain
// (1.0e-7 / 1.8e-8: fewer roundings), and 16.4 us per 4096 x 512 x 512 pair against 20.5.  What bounds it now is no longer the matrix
// pipe: one MFMA instead of six takes 9.9 us (the split's VALU work, 1.5 x the LDS store volume, HBM: G + X + the partial tiles).
// LDS image per (operand, part): [32 rows k][64 bf16], 128-byte rows with the two 64-byte halves swapped on rows with (k >> 1) & 1,
// so that the four rows of a transposed read fall on disjoint banks; operand fragments by ds_read_b64_tr_b16 (4 rows x 16 columns
// per 16-lane group, delivered column-major: the k-contiguous fragment the MFMA wants from a [k][m] image).
// Inf operands give NaN (inf - inf in the split); everything finite, subnormals included, is exact up to the dropped terms.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <bool RAGGED>
__device__ __forceinline__ void wgrad_body_x3(const GroupArgs &ga_) {
  constexpr int BK = 32, ROWB = 128, PART = BK * ROWB, BUF = 2 * 3 * PART;      // bytes: row, (operand, part), buffer
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];                   // 48 KB; staging for the two reductions at the end
  int p = 0;
#pragma unroll
  for (int q = 1; q < kMaxPairs; ++q) p += (q < ga_.npairs && (int)blockIdx.x >= ga_.first[q]) ? 1 : 0;
  const float *__restrict__ G = static_cast<const float *>(ga_.g[p]);
  const float *__restrict__ X = static_cast<const float *>(ga_.x[p]);
  float *__restrict__ PW = static_cast<float *>(ga_.pw[p]);
  double *__restrict__ PB = ga_.pb[p];
  const int M = ga_.M[p], N = ga_.N[p], K = ga_.K;
  const float alpha = (float)ga_.alpha[p];
  const int bid = (int)blockIdx.x - ga_.first[p];
  const int split = bid % kSplit, tile = bid / kSplit;
  const int ntn = N / BN;
  const int tm = tile / ntn, tn = tile % ntn;
  // rows per K range: whole slabs; rows that do not exist (K is not a multiple of 8 slabs) are loaded as zeros
  const int kper = RAGGED ? (K + kSplit * BK - 1) / (kSplit * BK) * BK : K / kSplit, k0 = split * kper, nslab = kper / BK;
  constexpr bool ragged = RAGGED;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int kh = w >> 2, wm = w & 1, wn = (w >> 1) & 1;          // K half, tile row, tile column of this wave
  const int lrow = t >> 4, lc = t & 15;                          // this thread's vector of a slab: row lrow, columns 4 lc .. 4 lc + 3
  const bool bias = PB != nullptr;
  f32x4 gv, xv;
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  float *pw = PW + (size_t)split * M * N;
  f32x16 acc, old;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  double *pbp = bias && t < BM ? PB + ((size_t)split * ntn + tn) * M + tm * BM + t : nullptr;
  auto swz = [](int k, int c) { return c ^ (((k >> 1) & 1) << 5); };

  auto gload = [&](int slab) {
    const int row = k0 + slab * BK + lrow;
    if (!ragged || row < K) {
      gv = *reinterpret_cast<const f32x4 *>(G + (size_t)row * M + tm * BM + lc * 4);
      xv = *reinterpret_cast<const f32x4 *>(X + (size_t)row * N + tn * BN + lc * 4);
    } else {
      gv = f32x4{0.f, 0.f, 0.f, 0.f}, xv = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto split_store = [&](const f32x4 v, char *base) {
    unsigned a[4], r1[4], r2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = __float_as_uint(v[e]);
      const float f1 = v[e] - __uint_as_float(a[e] & 0xFFFF0000u);               // exact
      r1[e] = __float_as_uint(f1);
      r2[e] = __float_as_uint(f1 - __uint_as_float(r1[e] & 0xFFFF0000u));       // exact; at most 8 significant bits are left
    }
    // v_perm_b32 with this selector: the top halves of two patterns side by side = two bf16, truncated
    const u32x2 hi = {__builtin_amdgcn_perm(a[1], a[0], 0x07060302u), __builtin_amdgcn_perm(a[3], a[2], 0x07060302u)};
    const u32x2 mid = {__builtin_amdgcn_perm(r1[1], r1[0], 0x07060302u), __builtin_amdgcn_perm(r1[3], r1[2], 0x07060302u)};
    const u32x2 lo = {__builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u), __builtin_amdgcn_perm(r2[3], r2[2], 0x07060302u)};
    const int off = lrow * ROWB + swz(lrow, lc * 4) * 2;
    *reinterpret_cast<u32x2 *>(base + off) = hi;
    *reinterpret_cast<u32x2 *>(base + PART + off) = mid;
    *reinterpret_cast<u32x2 *>(base + 2 * PART + off) = lo;
  };
  // The workgroups of a tile row (tn = 0..ntn-1) see the same G slabs: slab s is added up by the one with tn == s % ntn.
  int mine = tn;                                                 // the next slab whose column sums are this workgroup's
  auto lstore = [&](int slab, int buf) {
    char *b = smem + buf * BUF;
    const f32x4 g = alpha * gv;
    split_store(g, b);
    if (slab == mine) {
      mine += ntn;
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) colsum[e] += g[e];
      }
    }
    split_store(xv, b + 3 * PART);
  };
  // the operand fragment of v_mfma_f32_32x32x16_bf16 for this wave's 32 columns and its half of the slab's rows: lane l holds
  // element [k = 8 (l >> 5) + j][col0 + (l & 31)], j = 0..7 -- two transposed reads: lane 4q + p of a 16-lane group supplies the
  // address of row q, columns 4p .. 4p + 3 of the group's 4 x 16 block and receives column (lane & 15), rows 0..3
  const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  auto frag = [&](const char *part, int col0) -> s16x8 {
    s16x4 r[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = kh * 16 + 8 * (g4 >> 1) + 4 * u + q4;
      const int c = col0 + (g4 & 1) * 16 + 4 * p4;
      r[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(part + k * ROWB + swz(k, c) * 2));
    }
    const s16x8 f = {r[0][0], r[0][1], r[0][2], r[0][3], r[1][0], r[1][1], r[1][2], r[1][3]};
    return f;
  };
  auto compute = [&](int buf) {
    const char *b = smem + buf * BUF;
    const s16x8 ah = frag(b, wm * 32), am = frag(b + PART, wm * 32), al = frag(b + 2 * PART, wm * 32);
    const s16x8 bh = frag(b + 3 * PART, wn * 32), bm = frag(b + 4 * PART, wn * 32), bl = frag(b + 5 * PART, wn * 32);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);      // the small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  };
  auto tile_row = [&](int e) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); };      // (C/D map of the 32 x 32 MFMAs: dtype-independent)

  gload(0);
  // what the earlier stages / time steps left in PW (and PB): BEHIND the first slab's loads (loads return in order)
#pragma unroll
  for (int e = 0; e < 16; ++e) old[e] = kh == 0 ? pw[(size_t)(tm * BM + wm * 32 + tile_row(e)) * N + tn * BN + wn * 32 + (lane & 31)] : 0.f;
  const double pbold = pbp ? *pbp : 0.0;
  lstore(0, 0);
  if (nslab > 1) gload(1);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    // the split and the stores of the next slab FIRST (its loads were issued a slab ago; the other buffer's last readers passed the
    // barrier before), the MFMAs behind them: they are still in the pipe while the wave goes on to the barrier (-4 % in
    // tools/mb_wgrad_bf16x3.hip, nothing measurable here)
    if (s + 1 < nslab) lstore(s + 1, (s + 1) & 1);
    compute(s & 1);
    if (s + 2 < nslab) gload(s + 2);
    __syncthreads();
  }
  float(*red)[64] = reinterpret_cast<float(*)[64]>(smem);
  double(*cs)[BM] = reinterpret_cast<double(*)[BM]>(smem + 4 * 16 * 64 * sizeof(float));
  static_assert(sizeof(smem) >= 4 * 16 * 64 * sizeof(float) + 8 * BM * sizeof(double), "the reductions are staged in the slab buffers");
  if (kh == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(w & 3) * 16 + e][lane] = acc[e];
  }
  if (bias) {
    // the four rows of a wave (lanes 16 apart hold the same columns) are added by shuffles, the eight waves through LDS: the last
    // thread's serial sum is 8 long, not 32 (it sits in the tail of every workgroup)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = colsum[e];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 16) cs[w][lc * 4 + e] = (double)v;
    }
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      pw[(size_t)(tm * BM + wm * 32 + tile_row(e)) * N + tn * BN + wn * 32 + (lane & 31)] = old[e] + (acc[e] + red[(w & 3) * 16 + e][lane]);
  }
  if (pbp) {
    double sum = cs[0][t];
#pragma unroll
    for (int j = 1; j < 8; ++j) sum += cs[j][t];
    *pbp = pbold + sum;
  }
}

// The same product on workgroup tiles of 128 x 128 by SIXTEEN waves (1024 threads): 2 K halves x (2 x 4) wave tiles of 64 x 32 --
// two accumulators per wave, the three X fragments feed both.  What bounds the 64 x 64 form is not the matrix pipe but what feeds it:
// per MFMA 1 KB of fragments read from LDS, 0.5 KB of split parts stored, and the split's VALU work, done once per workgroup that
// shares a slab (8 of them at 512 columns).  Here: 0.75 KB read, 0.25 KB stored, half the split work per MFMA -- 14.3 against 17.2 us
// per 4096 x 512 x 512 pair, four pairs per launch (tools/mb_wgrad_bf16x3.hip, profiles/r06_microbench.txt).  12 parts of 4 KB per
// buffer, two buffers = 96 KB: ONE workgroup per CU, so the launcher takes this form only when the launch fills whole rounds of
// the chip.  The arithmetic of every element of dW and of db is that of the 64 x 64 form, operation for operation (same K ranges,
// same slabs, same order of the six terms, the K halves added last; db: per (row of a slab, column) float sums over the slabs the
// 64-wide tile column would have taken, rows paired as its shuffles pair them, eight double adds): the SAME BITS, so a result
// does not depend on which form a launch took -- tests/test_gpu_kernels.py compares them.
template <bool RAGGED>
__device__ __forceinline__ void wgrad_body_x3_wide(const GroupArgs &ga_) {
  constexpr int BK = 32, ROWB = 128, PART = BK * ROWB, BUF = 12 * PART, TM = 128, TN = 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];                   // 96 KB
  int p = 0;
#pragma unroll
  for (int q = 1; q < kMaxPairs; ++q) p += (q < ga_.npairs && (int)blockIdx.x >= ga_.first[q]) ? 1 : 0;
  const float *__restrict__ G = static_cast<const float *>(ga_.g[p]);
  const float *__restrict__ X = static_cast<const float *>(ga_.x[p]);
  float *__restrict__ PW = static_cast<float *>(ga_.pw[p]);
  double *__restrict__ PB = ga_.pb[p];
  const int M = ga_.M[p], N = ga_.N[p], K = ga_.K;
  const float alpha = (float)ga_.alpha[p];
  const int bid = (int)blockIdx.x - ga_.first[p];
  const int split = bid % kSplit, tile = bid / kSplit;
  const int ntn = N / TN, ntn64 = N / BN;
  const int tm = tile / ntn, tn = tile % ntn;
  const int kper = RAGGED ? (K + kSplit * BK - 1) / (kSplit * BK) * BK : K / kSplit, k0 = split * kper, nslab = kper / BK;
  constexpr bool ragged = RAGGED;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int kh = w >> 3, wm = w & 1, wn = (w >> 1) & 3;          // K half; rows 64 wm .. +63, columns 32 wn .. +31 of the tile
  const int lrow = t >> 5, lc5 = t & 31, sub = lc5 >> 4, lc = lc5 & 15;       // this thread's vector: row lrow, columns 4 lc5 .. +3 (64-column sub-tile `sub`)
  const bool bias = PB != nullptr;
  f32x4 gv, xv;
  float colsum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float *pw = PW + (size_t)split * M * N;
  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc0[e] = 0.f, acc1[e] = 0.f;
  // db: this workgroup stands for the 64-wide tile columns 2 tn and 2 tn + 1 of its tile row -- their slots of PB, their slabs
  double *pbp = bias && t < 2 * TM ? PB + ((size_t)split * ntn64 + 2 * tn + (t >> 7)) * M + tm * TM + (t & 127) : nullptr;
  auto swz = [](int k, int c) { return c ^ (((k >> 1) & 1) << 5); };

  auto gload = [&](int slab) {
    const int row = k0 + slab * BK + lrow;
    if (!ragged || row < K) {
      gv = *reinterpret_cast<const f32x4 *>(G + (size_t)row * M + tm * TM + lc5 * 4);
      xv = *reinterpret_cast<const f32x4 *>(X + (size_t)row * N + tn * TN + lc5 * 4);
    } else {
      gv = f32x4{0.f, 0.f, 0.f, 0.f}, xv = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto split_store = [&](const f32x4 v, char *base) {
    unsigned a[4], r1[4], r2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = __float_as_uint(v[e]);
      const float f1 = v[e] - __uint_as_float(a[e] & 0xFFFF0000u);
      r1[e] = __float_as_uint(f1);
      r2[e] = __float_as_uint(f1 - __uint_as_float(r1[e] & 0xFFFF0000u));
    }
    const u32x2 hi = {__builtin_amdgcn_perm(a[1], a[0], 0x07060302u), __builtin_amdgcn_perm(a[3], a[2], 0x07060302u)};
    const u32x2 mid = {__builtin_amdgcn_perm(r1[1], r1[0], 0x07060302u), __builtin_amdgcn_perm(r1[3], r1[2], 0x07060302u)};
    const u32x2 lo = {__builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u), __builtin_amdgcn_perm(r2[3], r2[2], 0x07060302u)};
    const int off = lrow * ROWB + swz(lrow, lc * 4) * 2;
    *reinterpret_cast<u32x2 *>(base + off) = hi;
    *reinterpret_cast<u32x2 *>(base + PART + off) = mid;
    *reinterpret_cast<u32x2 *>(base + 2 * PART + off) = lo;
  };
  int mine0 = 2 * tn, mine1 = 2 * tn + 1;                        // the next slabs of the two 64-wide tile columns
  auto lstore = [&](int slab, int buf) {
    char *b = smem + buf * BUF;
    const f32x4 g = alpha * gv;
    split_store(g, b + sub * 3 * PART);                          // G: two sub-tiles of 64 columns x (hi, mid, lo)
    if (slab == mine0) {
      mine0 += ntn64;
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) colsum[0][e] += g[e];
      }
    }
    if (slab == mine1) {
      mine1 += ntn64;
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) colsum[1][e] += g[e];
      }
    }
    split_store(xv, b + (6 + sub * 3) * PART);                   // X: the same
  };
  const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  auto frag = [&](const char *part, int col0) -> s16x8 {
    s16x4 r[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = kh * 16 + 8 * (g4 >> 1) + 4 * u + q4;
      const int c = col0 + (g4 & 1) * 16 + 4 * p4;
      r[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(part + k * ROWB + swz(k, c) * 2));
    }
    const s16x8 f = {r[0][0], r[0][1], r[0][2], r[0][3], r[1][0], r[1][1], r[1][2], r[1][3]};
    return f;
  };
  auto six = [&](f32x16 &acc, const s16x8 ah, const s16x8 am, const s16x8 al, const s16x8 bh, const s16x8 bm, const s16x8 bl) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);      // the order of the 64 x 64 form
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  };
  auto compute = [&](int buf) {
    const char *b = smem + buf * BUF;
    const char *ga = b + wm * 3 * PART;                          // this wave's 64 rows of dW = G sub-tile wm
    const char *xb = b + (6 + (wn >> 1) * 3) * PART;
    const int bc = (wn & 1) * 32;
    const s16x8 bh = frag(xb, bc), bm = frag(xb + PART, bc), bl = frag(xb + 2 * PART, bc);
    {
      const s16x8 ah = frag(ga, 0), am = frag(ga + PART, 0), al = frag(ga + 2 * PART, 0);
      six(acc0, ah, am, al, bh, bm, bl);
    }
    {
      const s16x8 ah = frag(ga, 32), am = frag(ga + PART, 32), al = frag(ga + 2 * PART, 32);
      six(acc1, ah, am, al, bh, bm, bl);
    }
  };
  auto tile_row = [&](int e) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); };

  gload(0);
  const double pbold = pbp ? *pbp : 0.0;
  lstore(0, 0);
  if (nslab > 1) gload(1);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    if (s + 1 < nslab) lstore(s + 1, (s + 1) & 1);
    compute(s & 1);
    if (s + 2 < nslab) gload(s + 2);
    __syncthreads();
  }
  // the partial tile left by the earlier stages / time steps: loaded HERE (64 registers that the loop does not carry)
  f32x16 old0, old1;
  float *tp = pw + (size_t)(tm * TM + wm * 64) * N + tn * TN + wn * 32 + (lane & 31);
  if (kh == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) old0[e] = tp[(size_t)tile_row(e) * N], old1[e] = tp[(size_t)(32 + tile_row(e)) * N];
  }
  float(*red)[64] = reinterpret_cast<float(*)[64]>(smem);                                  // 8 waves x 32 values x 64 lanes = 64 KB
  float(*csf)[16][TM] = reinterpret_cast<float(*)[16][TM]>(smem + 8 * 32 * 64 * sizeof(float));      // + 16 KB
  static_assert(sizeof(smem) >= 8 * 32 * 64 * sizeof(float) + 2 * 16 * TM * sizeof(float), "the reductions are staged in the slab buffers");
  if (kh == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(w & 7) * 32 + e][lane] = acc0[e], red[(w & 7) * 32 + 16 + e][lane] = acc1[e];
  }
  if (bias) {
    // a wave holds two rows of the slab (lanes 32 apart: the same columns); the 64 x 64 form adds rows 4j .. 4j + 3 by shuffles in
    // float ((r0 + r1) + (r2 + r3)) and its eight waves in double: waves 2j and 2j + 1 here are its wave j
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = colsum[h][e];
        v += __shfl_xor(v, 32);
        if (lane < 32) csf[h][w][lc5 * 4 + e] = v;
      }
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      tp[(size_t)tile_row(e) * N] = old0[e] + (acc0[e] + red[(w & 7) * 32 + e][lane]);
      tp[(size_t)(32 + tile_row(e)) * N] = old1[e] + (acc1[e] + red[(w & 7) * 32 + 16 + e][lane]);
    }
  }
  if (pbp) {
    const int h = t >> 7, m = t & 127;
    double sum = (double)(csf[h][0][m] + csf[h][1][m]);
#pragma unroll
    for (int j = 1; j < 8; ++j) sum += (double)(csf[h][2 * j][m] + csf[h][2 * j + 1][m]);
    *pbp = pbold + sum;
  }
}

// (fp64: 128 VGPRs as it compiles, two workgroups per CU; its ragged form 131: one.)  fp32: at most 80 VGPRs, so that THREE workgroups fit a CU (six waves per SIMD; LDS 3 x 32 KB): while one is in its prologue or
// its tail the other two keep the matrix pipes busy.  fp64 needs 118 VGPRs (two per CU).
__global__ __launch_bounds__(kThreads, 6) void pn_linear_wgrad_kernel_f32(GroupArgs a) { wgrad_body<float, false>(a); }
__global__ __launch_bounds__(kThreads) void pn_linear_wgrad_kernel_f64(GroupArgs a) { wgrad_body<double, false>(a); }
__global__ __launch_bounds__(kThreads, 4) void pn_linear_wgrad_kernel_f32x3(GroupArgs a) { wgrad_body_x3<false>(a); }
__global__ __launch_bounds__(kThreads, 6) void pn_linear_wgrad_kernel_f32_ragged(GroupArgs a) { wgrad_body<float, true>(a); }
__global__ __launch_bounds__(kThreads) void pn_linear_wgrad_kernel_f64_ragged(GroupArgs a) { wgrad_body<double, true>(a); }
__global__ __launch_bounds__(kThreads, 4) void pn_linear_wgrad_kernel_f32x3_ragged(GroupArgs a) { wgrad_body_x3<true>(a); }
__global__ __launch_bounds__(1024, 4) void pn_linear_wgrad_kernel_f32x3w(GroupArgs a) { wgrad_body_x3_wide<false>(a); }
__global__ __launch_bounds__(1024, 4) void pn_linear_wgrad_kernel_f32x3w_ragged(GroupArgs a) { wgrad_body_x3_wide<true>(a); }

// mu_W += sum_s PW[s] (s = 0..7, in that order); PW = 0
template <typename T>
__global__ __launch_bounds__(256) void pn_linear_wgrad_finish_kernel(T *__restrict__ PW, size_t mn, T *__restrict__ mu) {
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
  if (i >= mn) return;
  vec_t v[kSplit];
#pragma unroll
  for (int s = 0; s < kSplit; ++s) v[s] = *reinterpret_cast<const vec_t *>(PW + (size_t)s * mn + i);
  vec_t m = *reinterpret_cast<const vec_t *>(mu + i);
  vec_t sum = v[0];
#pragma unroll
  for (int s = 1; s < kSplit; ++s) sum += v[s];
  m += sum;
  *reinterpret_cast<vec_t *>(mu + i) = m;
  vec_t z;
#pragma unroll
  for (int e = 0; e < VEC; ++e) z[e] = (T)0;
#pragma unroll
  for (int s = 0; s < kSplit; ++s) *reinterpret_cast<vec_t *>(PW + (size_t)s * mn + i) = z;
}

template <typename T>
__global__ __launch_bounds__(256) void pn_linear_bgrad_finish_kernel(double *__restrict__ PB, int M, int parts, T *__restrict__ mu) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  double s = 0.0;
  int k = 0;
  for (; k + 8 <= parts; k += 8) {               // (parts = 8 * tile columns) eight loads in flight, added in index order
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = PB[(size_t)(k + j) * M + m];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; k < parts; ++k) s += PB[(size_t)k * M + m];
  mu[m] += (T)s;
  for (k = 0; k < parts; ++k) PB[(size_t)k * M + m] = 0.0;
}

int cu_count() {
  static int n = 0;              // (the calling thread's current device; every device of a node is the same part)
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;
  }
  return n;
}

template <typename T>
int finish_t(hipStream_t st, int64_t out_f, int64_t in_f, void *pw, void *pb, void *mu_w, void *mu_b) {
  const size_t mn = (size_t)out_f * (size_t)in_f;
  constexpr int VEC = 16 / (int)sizeof(T);
  hipLaunchKernelGGL(pn_linear_wgrad_finish_kernel<T>, dim3((unsigned)((mn / VEC + 255) / 256)), dim3(256), 0, st, (T *)pw, mn, (T *)mu_w);
  if (pb != nullptr && mu_b != nullptr)
    hipLaunchKernelGGL(pn_linear_bgrad_finish_kernel<T>, dim3((unsigned)((out_f + 255) / 256)), dim3(256), 0, st, (double *)pb, (int)out_f,
                       (int)(kSplit * (in_f / BN)), (T *)mu_b);
  return 0;
}

}  // namespace

extern "C" {

int pn_linear_wgrad_supported(int dtype, int64_t rows, int64_t out_f, int64_t in_f) {
  if (dtype != PN_F32 && dtype != PN_F64) return 0;
  // any number of rows from 256 on (a K range is rounded up to whole slabs of 32 rows, the missing rows are zeros); fewer rows than
  // that leave most of the eight K ranges empty: the general path then
  return (rows >= 256 && rows < (int64_t)1 << 30 && out_f > 0 && out_f % BM == 0 && in_f > 0 && in_f % BN == 0 && out_f * in_f <= kMaxWeights) ? 1 : 0;
}

int64_t pn_linear_wgrad_work_bytes(int dtype, int64_t out_f, int64_t in_f, int64_t *bias_bytes) {
  if (bias_bytes) *bias_bytes = (int64_t)kSplit * (in_f / BN) * out_f * (int64_t)sizeof(double);
  return (int64_t)kSplit * out_f * in_f * (int64_t)(dtype == PN_F64 ? sizeof(double) : sizeof(float));
}

int pn_linear_wgrad_group(void *stream, int dtype, int64_t rows, int npairs, const pn_wgrad_pair *pairs, int flags) {
  if (npairs < 1 || npairs > kMaxPairs) return pn::fail("pn_linear_wgrad_group: 1 <= npairs <= PN_WGRAD_MAX_PAIRS");
  GroupArgs a;
  a.npairs = npairs;
  a.K = (int)rows;
  int64_t blocks = 0, wblocks = 0;
  double flops = 0;
  bool wide = dtype == PN_F32 && !(flags & (PN_WGRAD_EXACT_FP32 | PN_WGRAD_TILE_64));
  for (int p = 0; p < kMaxPairs; ++p) {
    const pn_wgrad_pair &q = pairs[p < npairs ? p : 0];
    if (p < npairs) {
      if (!pn_linear_wgrad_supported(dtype, rows, q.out_f, q.in_f))
        return pn::fail("pn_linear_wgrad: unsupported dtype or shape (see pn_linear_wgrad_supported)");
      if ((((uintptr_t)q.g) | ((uintptr_t)q.x) | ((uintptr_t)q.pw)) & 15) return pn::fail("pn_linear_wgrad: operands must be 16-byte aligned");
      if (q.g == nullptr || q.x == nullptr || q.pw == nullptr) return pn::fail("pn_linear_wgrad: null operand");
    }
    a.g[p] = q.g, a.x[p] = q.x, a.pw[p] = q.pw, a.pb[p] = (double *)q.pb, a.alpha[p] = q.alpha;
    a.M[p] = (int)q.out_f, a.N[p] = (int)q.in_f;
    if (p < npairs) {
      flops += 2.0 * (double)rows * (double)q.out_f * (double)q.in_f;
      wblocks += (q.out_f / 128) * (q.in_f / 128) * kSplit;
      if (q.out_f % 128 || q.in_f % 128) wide = false;
    }
  }
  // The 128 x 128 form runs ONE workgroup per CU: taken when the launch fills whole rounds of the chip (at most a fifth of the last
  // round idle) -- the four 512 x 512 layers of a stage VJP are two rounds of 256 -- and not by a single small layer, which the
  // 64 x 64 form spreads over four times as many workgroups.  Same bits either way (wgrad_body_x3_wide).
  if (wide) {
    const int64_t ncu = cu_count(), rounds = (wblocks + ncu - 1) / ncu;
    wide = wblocks >= ncu && rounds * ncu * 5 <= wblocks * 6;
  }
  for (int p = 0; p < kMaxPairs; ++p) {
    a.first[p] = (int)blocks;
    if (p < npairs) blocks += wide ? (pairs[p].out_f / 128) * (pairs[p].in_f / 128) * kSplit : (pairs[p].out_f / BM) * (pairs[p].in_f / BN) * kSplit;
  }
  a.first[kMaxPairs] = (int)blocks;
  for (int p = npairs; p < kMaxPairs; ++p) a.first[p] = (int)blocks;
  if (blocks > (int64_t)1 << 30) return pn::fail("pn_linear_wgrad_group: too many workgroups");
  void *v0 = nullptr, *v1 = nullptr;
  const int prof = pn::prof_events(PN_K_LINEAR_WGRAD, flops, &v0, &v1);
  hipEvent_t e0 = (hipEvent_t)v0, e1 = (hipEvent_t)v1;
  if (prof < 0) return 1;
  hipStream_t st = (hipStream_t)stream;
  const bool ragged = rows % (kSplit * 32) != 0;
  if (wide) {
    auto kw = ragged ? pn_linear_wgrad_kernel_f32x3w_ragged : pn_linear_wgrad_kernel_f32x3w;
    if (prof)
      hipExtLaunchKernelGGL(kw, dim3((unsigned)blocks), dim3(1024), 0, st, e0, e1, 0, a);
    else
      hipLaunchKernelGGL(kw, dim3((unsigned)blocks), dim3(1024), 0, st, a);
    hipError_t werr = hipGetLastError();
    if (werr != hipSuccess) return pn::fail(std::string("pn_linear_wgrad: ") + hipGetErrorString(werr));
    return 0;
  }
  auto kern = dtype == PN_F64 ? (ragged ? pn_linear_wgrad_kernel_f64_ragged : pn_linear_wgrad_kernel_f64)
                              : ((flags & PN_WGRAD_EXACT_FP32) ? (ragged ? pn_linear_wgrad_kernel_f32_ragged : pn_linear_wgrad_kernel_f32)
                                                               : (ragged ? pn_linear_wgrad_kernel_f32x3_ragged : pn_linear_wgrad_kernel_f32x3));
  if (prof)
    hipExtLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kThreads), 0, st, e0, e1, 0, a);
  else
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kThreads), 0, st, a);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pn::fail(std::string("pn_linear_wgrad: ") + hipGetErrorString(err));
  return 0;
}

int pn_linear_wgrad(void *stream, int dtype, int64_t rows, int64_t out_f, int64_t in_f, const void *g, const void *x, double alpha,
                    void *pw, void *pb) {
  pn_wgrad_pair q;
  q.g = g, q.x = x, q.pw = pw, q.pb = pb, q.alpha = alpha, q.out_f = out_f, q.in_f = in_f;
  return pn_linear_wgrad_group(stream, dtype, rows, 1, &q, 0);
}

int pn_linear_wgrad_finish(void *stream, int dtype, int64_t out_f, int64_t in_f, void *pw, void *pb, void *mu_w, void *mu_b) {
  if (dtype != PN_F32 && dtype != PN_F64) return pn::fail("pn_linear_wgrad_finish: fp32 or fp64");
  if ((((uintptr_t)pw) | ((uintptr_t)mu_w)) & 15) return pn::fail("pn_linear_wgrad_finish: operands must be 16-byte aligned");
  if (dtype == PN_F32)
    finish_t<float>((hipStream_t)stream, out_f, in_f, pw, pb, mu_w, mu_b);
  else
    finish_t<double>((hipStream_t)stream, out_f, in_f, pw, pb, mu_w, mu_b);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pn::fail(std::string("pn_linear_wgrad_finish: ") + hipGetErrorString(err));
  return 0;
}

}  // extern "C"
