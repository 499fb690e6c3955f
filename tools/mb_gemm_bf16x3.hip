// Microbenchmark (round 6, exploratory): the forward product of a Linear layer, Y = X W^T + b (X: M x K, W: N x K, both row-major:
// K-contiguous operands), in fp32 on the BF16 matrix cores by exact three-way operand splitting -- the arithmetic of
// pn_linear_wgrad_kernel_f32x3 (csrc/pn_linear.hip) with K-contiguous operands: fragments by ds_read_b128 from a [row][32 k] image
// (64-byte rows, the 16-byte chunk index XORed with (row >> 2) & 3), no transposed reads, no K split, no partial tiles.
// hipBLASLt serves this shape (4096 x 512 x 512, bias epilogue) in 19.7-20.4 us inside the C3a sweep.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mb_gemm_bf16x3 tools/mb_gemm_bf16x3.hip && tools/mb_gemm_bf16x3
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      std::exit(1);                                                                       \
    }                                                                                     \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 512, BK = 32;

template <int BM, int BN>
struct Cfg {
  static constexpr int ROWB = 64;                                 // bytes per LDS row (32 bf16)
  static constexpr int PA = BM * ROWB, PB = BN * ROWB;            // bytes per part of the X tile / of the W tile
  static constexpr int BUF = 3 * (PA + PB);
};

// BM x BN tile per workgroup of 8 waves: (BM/32) x (BN/32) tiles of 32 x 32 spread over WM x WN waves x 2 K halves
template <int BM, int BN, int TERMS>
__global__ __launch_bounds__(kThreads) void gemm_nt_x3(const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
                                                       float *__restrict__ Y, int M, int N, int K) {
  using C = Cfg<BM, BN>;
  constexpr int TM = BM / 64, TN = BN / 64;                       // 32 x 32 tiles per wave in each direction (waves: 2 x 2 x 2 K halves)
  __shared__ __attribute__((aligned(16))) char smem[2 * C::BUF];
  const int ntn = N / BN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int kh = w >> 2, wm = w & 1, wn = (w >> 1) & 1;
  const int nslab = K / BK;
  constexpr int XV = BM * BK / 4 / kThreads, WV = BN * BK / 4 / kThreads;       // float4 per thread per slab
  f32x4 xv[XV], wv[WV];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto swz = [](int r, int c16) { return c16 ^ ((r >> 2) & 3); };                // 16-byte chunk (0..3) of row r
  auto gload = [&](int slab) {
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int idx = t + kThreads * v, r = idx >> 3, kq = idx & 7;
      xv[v] = *reinterpret_cast<const f32x4 *>(X + (size_t)(tm * BM + r) * K + slab * BK + kq * 4);
    }
#pragma unroll
    for (int v = 0; v < WV; ++v) {
      const int idx = t + kThreads * v, r = idx >> 3, kq = idx & 7;
      wv[v] = *reinterpret_cast<const f32x4 *>(W + (size_t)(tn * BN + r) * K + slab * BK + kq * 4);
    }
  };
  auto split_store = [&](const f32x4 v, char *base, int part_bytes, int r, int kq) {
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
    const int off = r * C::ROWB + swz(r, kq >> 1) * 16 + (kq & 1) * 8;
    *reinterpret_cast<u32x2 *>(base + off) = hi;
    *reinterpret_cast<u32x2 *>(base + part_bytes + off) = mid;
    *reinterpret_cast<u32x2 *>(base + 2 * part_bytes + off) = lo;
  };
  auto lstore = [&](int buf) {
    char *b = smem + buf * C::BUF;
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int idx = t + kThreads * v;
      split_store(xv[v], b, C::PA, idx >> 3, idx & 7);
    }
#pragma unroll
    for (int v = 0; v < WV; ++v) {
      const int idx = t + kThreads * v;
      split_store(wv[v], b + 3 * C::PA, C::PB, idx >> 3, idx & 7);
    }
  };
  // lane l: row (l & 31) of the 32-row block, k = kh * 16 + 8 (l >> 5) + j, j = 0..7: one 16-byte chunk
  auto frag = [&](const char *part, int row0) -> s16x8 {
    const int r = row0 + (lane & 31), c16 = kh * 2 + (lane >> 5);
    return *reinterpret_cast<const s16x8 *>(part + r * C::ROWB + swz(r, c16) * 16);
  };
  auto compute = [&](int buf) {
    const char *b = smem + buf * C::BUF;
    s16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = (wm * TM + i) * 32;
      ah[i] = frag(b, row0), am[i] = frag(b + C::PA, row0), al[i] = frag(b + 2 * C::PA, row0);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row0 = (wn * TN + j) * 32;
      bh[j] = frag(b + 3 * C::PA, row0), bm[j] = frag(b + 3 * C::PA + C::PB, row0), bl[j] = frag(b + 3 * C::PA + 2 * C::PB, row0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x16 c = acc[i][j];
        if (TERMS >= 6) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], c, 0, 0, 0);
        acc[i][j] = c;
      }
  };

  gload(0);
  lstore(0);
  if (nslab > 1) gload(1);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    compute(s & 1);
    if (s + 1 < nslab) lstore((s + 1) & 1);
    if (s + 2 < nslab) gload(s + 2);
    __syncthreads();
  }
  // the second K half hands its tiles to the first; bias; store (col = lane & 31: 128 contiguous bytes per row per register)
  float(*red)[64] = reinterpret_cast<float(*)[64]>(smem);
  static_assert(sizeof(smem) >= (size_t)4 * TM * TN * 16 * 64 * 4, "staging of the second K half");
  if (kh == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(((w & 3) * TM + i) * TN + j) * 16 + e][lane] = acc[i][j][e];
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = tn * BN + (wn * TN + j) * 32 + (lane & 31);
        const float bj = bias ? bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = tm * BM + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          Y[(size_t)row * N + col] = (acc[i][j][e] + red[(((w & 3) * TM + i) * TN + j) * 16 + e][lane]) + bj;
        }
      }
  }
}

__global__ void naive_kernel(const float *X, const float *W, const float *bias, int M, int N, int K, double *out, double *mag) {
  const int m = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0, a = 0;
  for (int k = 0; k < K; ++k) {
    const double pr = (double)X[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
    s += pr;
    a += fabs(pr);
  }
  out[(size_t)m * N + n] = s + (double)bias[n];
  mag[(size_t)m * N + n] = a + fabs((double)bias[n]);
}

template <int BM, int BN, int TERMS>
void run(const char *name, int M, int N, int K, std::vector<float *> &Xd, float *W, float *bias, std::vector<float *> &Yd,
         const std::vector<double> &ref, const std::vector<double> &mag) {
  const int blocks = (M / BM) * (N / BN);
  gemm_nt_x3<BM, BN, TERMS><<<blocks, kThreads>>>(Xd[0], W, bias, Yd[0], M, N, K);
  CHECK(hipDeviceSynchronize());
  std::vector<float> h((size_t)M * N);
  CHECK(hipMemcpy(h.data(), Yd[0], h.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, rms = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    const double e = std::fabs((double)h[i] - ref[i]) / mag[i];
    worst = std::fmax(worst, e);
    rms += e * e;
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 300, np = (int)Xd.size();
  std::vector<double> us;
  for (int round = 0; round < 3; ++round) {
    for (int r = 0; r < 20; ++r) gemm_nt_x3<BM, BN, TERMS><<<blocks, kThreads>>>(Xd[r % np], W, bias, Yd[r % np], M, N, K);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) gemm_nt_x3<BM, BN, TERMS><<<blocks, kThreads>>>(Xd[r % np], W, bias, Yd[r % np], M, N, K);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    us.push_back(ms / reps * 1e3);
  }
  std::sort(us.begin(), us.end());
  std::printf("%-44s blocks %4d  error / sum|x w|: max %.2e rms %.2e | %6.2f us = %6.1f fp32-equivalent TFLOP/s\n", name, blocks, worst,
              std::sqrt(rms / h.size()), us[1], 2.0 * M * N * K / (us[1] * 1e-6) / 1e12);
  std::fflush(stdout);
}

int main() {
  const int M = 4096, N = 512, K = 512, NP = 8;
  std::vector<float *> Xd(NP), Yd(NP);
  std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N);
  srand(7);
  for (auto &v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.08f;
  for (auto &v : hb) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  float *W, *bias;
  CHECK(hipMalloc(&W, hw.size() * 4));
  CHECK(hipMalloc(&bias, hb.size() * 4));
  CHECK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  for (int p = NP - 1; p >= 0; --p) {
    srand(p + 1);
    for (auto &v : hx) v = (float)rand() / RAND_MAX - 0.37f;
    CHECK(hipMalloc(&Xd[p], hx.size() * 4));
    CHECK(hipMalloc(&Yd[p], (size_t)M * N * 4));
    CHECK(hipMemcpy(Xd[p], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  }
  double *refd, *magd;
  CHECK(hipMalloc(&refd, (size_t)M * N * 8));
  CHECK(hipMalloc(&magd, (size_t)M * N * 8));
  naive_kernel<<<dim3(N / 256, M), 256>>>(Xd[0], W, bias, M, N, K, refd, magd);
  std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
  CHECK(hipMemcpy(ref.data(), refd, ref.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(mag.data(), magd, mag.size() * 8, hipMemcpyDeviceToHost));
  for (int pass = 0; pass < 2; ++pass) {
    std::printf("--- pass %d\n", pass);
    run<64, 64, 6>("64 x 64 tiles (512 workgroups), 6 terms", M, N, K, Xd, W, bias, Yd, ref, mag);
    run<64, 64, 1>("64 x 64 tiles, 1 term (plain bf16)", M, N, K, Xd, W, bias, Yd, ref, mag);
    run<128, 64, 6>("128 x 64 tiles (256 workgroups), 6 terms", M, N, K, Xd, W, bias, Yd, ref, mag);
    run<128, 64, 1>("128 x 64 tiles, 1 term", M, N, K, Xd, W, bias, Yd, ref, mag);
    run<128, 128, 6>("128 x 128 tiles (128 workgroups), 6 terms", M, N, K, Xd, W, bias, Yd, ref, mag);
    run<64, 128, 6>("64 x 128 tiles (256 workgroups), 6 terms", M, N, K, Xd, W, bias, Yd, ref, mag);
  }
  return 0;
}
