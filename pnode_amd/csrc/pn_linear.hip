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
// one GEMM shape of the time step the BLAS library serves at 84 TFLOP/s (25.6 us; the forward- and dX-shaped products of the
// same size run at 110) -- and everything around it was extra passes: the column sum for db, a copy of the last layer's
// cotangent, the accumulation into mu.  Here:
//   * K is split eight ways and the split index is blockIdx % 8: workgroups are dealt to the 8 XCDs round-robin, so each XCD
//     works on ONE K range and the 2 MB of G and X rows it needs stay in its own 4 MB L2;
//   * 64 x 64 output tiles per workgroup of eight waves: four 32 x 32 tiles of v_mfma_f32_32x32x2_f32 (exact fp32, a k-ordered
//     fmaf chain), each computed twice over -- waves 0-3 take the even half of every K slab of 32, waves 4-7 the odd half, the
//     halves are added through LDS at the end --, slabs through LDS with the next slab's global loads in flight while this one
//     is multiplied; 512 workgroups for a 512 x 512 layer: two per CU, four waves per SIMD;
//   * the partial tile in PW is read under the K loop and added at its end: the sum over stages and time steps costs no pass;
//   * the workgroups of a tile row share the column sums of their G slabs between them (slab s: tile column s % ntn): db for free.
// Bit-reproducible (fixed split, fixed order, no atomics); independent of how the engine groups its other accumulations.
// fp32, rows % 256 == 0, out % 64 == 0, in % 64 == 0, out * in <= 2^22; everything else takes the general path (torch GEMM +
// pn_colsum_accum_multi).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <string>

#include "pn_internal.h"
#include "pnode_amd.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSplit = 8;        // K ranges = XCDs
constexpr int BM = 64, BN = 64, BK = 32, PAD = 4;
// Above 2048 x 2048 weights a layer has tiles enough to fill the chip without a K split and the partial buffers (8 x the weight)
// stop being small change: such layers are left to the library GEMM.
constexpr int64_t kMaxWeights = (int64_t)1 << 22;

// Eight waves per workgroup: waves 0-3 multiply the even half of every K slab (16 of its 32 rows), waves 4-7 the odd half, each
// its own 32 x 32 tile chain; the two halves are added through LDS at the end.  Four waves per SIMD instead of two: with two, the
// MFMA pipe was busy 0.62 of a wave's residence and a third of the wave cycles were spent parked at barriers and waitcnts
// (profiles/r05_pmc_wgrad.txt) -- more independent chains per SIMD cover that.
constexpr int kThreads = 512;

__global__ __launch_bounds__(kThreads) void pn_linear_wgrad_kernel(const float *__restrict__ G, const float *__restrict__ X, int K, int M, int N,
                                                                   float alpha, float *__restrict__ PW, double *__restrict__ PB) {
  __shared__ float smem[2][BK][BM + PAD];       // G slab, X slab (BM == BN); staging for the two reductions at the end
  float (*Gs)[BM + PAD] = smem[0];
  float (*Xs)[BN + PAD] = smem[1];
  static_assert(BM == BN, "one slab shape");
  const int split = blockIdx.x % kSplit, tile = blockIdx.x / kSplit;
  const int ntn = N / BN;
  const int tm = tile / ntn, tn = tile % ntn;
  const int kper = K / kSplit, k0 = split * kper, nslab = kper / BK;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int kh = w >> 2, wm = w & 1, wn = (w >> 1) & 1;          // K half, tile row, tile column of this wave
  const int lr = lane & 31, lh = lane >> 5;
  constexpr int GROW = BM / 4;                                   // 16-byte vectors per slab row; one G and one X vector per thread
  static_assert(BK * BM / 4 == kThreads, "one vector of each slab per thread");
  const int lrow = t / GROW, lc4 = t % GROW;
  const bool bias = PB != nullptr;
  f32x4 ga, xa;
  double colsum[4] = {0.0, 0.0, 0.0, 0.0};      // this thread's four columns of the G slabs that are this workgroup's to add up

  // what the earlier stages / time steps left in PW: loaded now (by the waves that will store the tile), needed after the K loop
  float *pw = PW + (size_t)split * M * N;
  f32x16 acc, old;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = tm * BM + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
    old[e] = kh == 0 ? pw[(size_t)row * N + tn * BN + wn * 32 + lr] : 0.f;
    acc[e] = 0.f;
  }
  // (the bias partial as well: a load in front of the final add would sit in the tail of every workgroup)
  double *pbp = bias && t < BM ? PB + ((size_t)split * ntn + tn) * M + tm * BM + t : nullptr;
  const double pbold = pbp ? *pbp : 0.0;

  auto gload = [&](int slab) {
    const int kb = k0 + slab * BK;
    ga = *reinterpret_cast<const f32x4 *>(G + (size_t)(kb + lrow) * M + tm * BM + lc4 * 4);
    xa = *reinterpret_cast<const f32x4 *>(X + (size_t)(kb + lrow) * N + tn * BN + lc4 * 4);
  };
  // The eight workgroups of a tile row (tn = 0..ntn-1) see the same G slabs: slab s is added up by the one with tn == s % ntn,
  // so that no workgroup carries the column sums alone (the launch ends with its slowest workgroup).
  auto lstore = [&](int slab) {
    const f32x4 v = alpha * ga;                  // here, not at the load: the product would wait for the load in front of the MFMAs
    *reinterpret_cast<f32x4 *>(&Gs[lrow][lc4 * 4]) = v;
    if (bias && slab % ntn == tn) {
#pragma unroll
      for (int e = 0; e < 4; ++e) colsum[e] += (double)v[e];
    }
    *reinterpret_cast<f32x4 *>(&Xs[lrow][lc4 * 4]) = xa;
  };

  gload(0);
  lstore(0);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    if (s + 1 < nslab) gload(s + 1);             // the next slab's global loads fly while this one is multiplied
    // LDS read, MFMA, LDS read, MFMA ...  (Tried and slower by 1 us: all of a slab's fragments read ahead of the MFMA chain, with
    // and without a second LDS buffer -- tools/mb_wgrad_abi.hip.)
#pragma unroll
    for (int kk = 0; kk < BK / 2; kk += 2) {
      const float a = Gs[kh * (BK / 2) + kk + lh][wm * 32 + lr];
      const float b = Xs[kh * (BK / 2) + kk + lh][wn * 32 + lr];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
    if (s + 1 < nslab) {
      lstore(s + 1);
      __syncthreads();
    }
  }
  if (bias) {
    // 32 threads share a column group (t % 16): add them in thread order through LDS (the slab buffers are free now)
    double (*cs)[BM] = reinterpret_cast<double (*)[BM]>(&smem[0][0][0]);
    static_assert(sizeof(smem) >= (kThreads / GROW) * BM * sizeof(double), "the column sums are staged in the slab buffers");
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[lrow][lc4 * 4 + e] = colsum[e];
    __syncthreads();
    if (t < BM) {
      double sum = cs[0][t];
#pragma unroll
      for (int j = 1; j < kThreads / GROW; ++j) sum += cs[j][t];
      *pbp = pbold + sum;
    }
    __syncthreads();
  }
  // the odd K half hands its tile to the even one: PW = PW + (even + odd)
  float (*red)[64] = reinterpret_cast<float (*)[64]>(&smem[0][0][0]);      // [4 waves x 16 registers][64 lanes]
  static_assert(sizeof(smem) >= 4 * 16 * 64 * sizeof(float), "the odd half's tiles are staged in the slab buffers");
  if (kh == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(w & 3) * 16 + e][lane] = acc[e];
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = tm * BM + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      pw[(size_t)row * N + tn * BN + wn * 32 + lr] = old[e] + (acc[e] + red[(w & 3) * 16 + e][lane]);
    }
  }
}

// mu_W += sum_s PW[s] (s = 0..7, in that order); PW = 0
__global__ __launch_bounds__(256) void pn_linear_wgrad_finish_kernel(float *__restrict__ PW, size_t mn, float *__restrict__ mu) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= mn) return;
  f32x4 v[kSplit];
#pragma unroll
  for (int s = 0; s < kSplit; ++s) v[s] = *reinterpret_cast<const f32x4 *>(PW + (size_t)s * mn + i);
  f32x4 m = *reinterpret_cast<const f32x4 *>(mu + i);
  f32x4 sum = v[0];
#pragma unroll
  for (int s = 1; s < kSplit; ++s) sum += v[s];
  m += sum;
  *reinterpret_cast<f32x4 *>(mu + i) = m;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < kSplit; ++s) *reinterpret_cast<f32x4 *>(PW + (size_t)s * mn + i) = z;
}

__global__ __launch_bounds__(256) void pn_linear_bgrad_finish_kernel(double *__restrict__ PB, int M, int parts, float *__restrict__ mu) {
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
  mu[m] += (float)s;
  for (k = 0; k < parts; ++k) PB[(size_t)k * M + m] = 0.0;
}

}  // namespace

extern "C" {

int pn_linear_wgrad_supported(int dtype, int64_t rows, int64_t out_f, int64_t in_f) {
  return (dtype == PN_F32 && rows > 0 && rows % (kSplit * BK) == 0 && out_f > 0 && out_f % BM == 0 && in_f > 0 && in_f % BN == 0 &&
          rows < (int64_t)1 << 30 && out_f * in_f <= kMaxWeights) ? 1 : 0;
}

int64_t pn_linear_wgrad_work_bytes(int64_t out_f, int64_t in_f, int64_t *bias_bytes) {
  if (bias_bytes) *bias_bytes = (int64_t)kSplit * (in_f / BN) * out_f * (int64_t)sizeof(double);
  return (int64_t)kSplit * out_f * in_f * (int64_t)sizeof(float);
}

int pn_linear_wgrad(void *stream, int dtype, int64_t rows, int64_t out_f, int64_t in_f, const void *g, const void *x, double alpha,
                    void *pw, void *pb) {
  if (!pn_linear_wgrad_supported(dtype, rows, out_f, in_f)) return pn::fail("pn_linear_wgrad: unsupported dtype or shape (see pn_linear_wgrad_supported)");
  if ((((uintptr_t)g) | ((uintptr_t)x) | ((uintptr_t)pw)) & 15) return pn::fail("pn_linear_wgrad: operands must be 16-byte aligned");
  const unsigned blocks = (unsigned)((out_f / BM) * (in_f / BN) * kSplit);
  void *v0 = nullptr, *v1 = nullptr;
  const int prof = pn::prof_events(PN_K_LINEAR_WGRAD, 2.0 * (double)rows * (double)out_f * (double)in_f, &v0, &v1);
  hipEvent_t e0 = (hipEvent_t)v0, e1 = (hipEvent_t)v1;
  if (prof < 0) return 1;
  if (prof)
    hipExtLaunchKernelGGL(pn_linear_wgrad_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, e0, e1, 0, (const float *)g, (const float *)x,
                          (int)rows, (int)out_f, (int)in_f, (float)alpha, (float *)pw, (double *)pb);
  else
    hipLaunchKernelGGL(pn_linear_wgrad_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, (const float *)g, (const float *)x, (int)rows,
                       (int)out_f, (int)in_f, (float)alpha, (float *)pw, (double *)pb);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pn::fail(std::string("pn_linear_wgrad: ") + hipGetErrorString(err));
  return 0;
}

int pn_linear_wgrad_finish(void *stream, int dtype, int64_t out_f, int64_t in_f, void *pw, void *pb, void *mu_w, void *mu_b) {
  if (dtype != PN_F32) return pn::fail("pn_linear_wgrad_finish: fp32 only");
  if ((((uintptr_t)pw) | ((uintptr_t)mu_w)) & 15) return pn::fail("pn_linear_wgrad_finish: operands must be 16-byte aligned");
  const size_t mn = (size_t)out_f * (size_t)in_f;
  hipLaunchKernelGGL(pn_linear_wgrad_finish_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float *)pw, mn,
                     (float *)mu_w);
  if (pb != nullptr && mu_b != nullptr)
    hipLaunchKernelGGL(pn_linear_bgrad_finish_kernel, dim3((unsigned)((out_f + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (double *)pb,
                       (int)out_f, (int)(kSplit * (in_f / BN)), (float *)mu_b);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return pn::fail(std::string("pn_linear_wgrad_finish: ") + hipGetErrorString(err));
  return 0;
}

}  // extern "C"
