// Microbenchmark (round 5): a hand-written fp32 MFMA kernel for the weight sensitivity of a Linear layer,
//   dW[m][n] = sum_k G[k][m] * X[k][n]   (G: K x M, X: K x N row-major; K = 4096 deep, M = N = 512)
// -- the K-deep "TN" product hipBLASLt serves at 84 TFLOP/s (25.6 us) inside the C3a time step.  Design: split K eight
// ways with split = blockIdx % 8, i.e. one K range per XCD (workgroups are dealt to the 8 XCDs round-robin), so that the
// 2 MB of G and X rows an XCD needs stay in its L2; BM x BN tiles per workgroup of 4 waves, v_mfma_f32_32x32x2_f32, K slabs
// of 32 through LDS with register prefetch of the next slab; partial tiles to a work buffer, reduced by a second kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mb_wgrad tools/mb_wgrad.hip && tools/mb_wgrad
#include <hip/hip_runtime.h>

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


// WM x WN = MFMA tiles (32 x 32) per wave; waves are laid out WAVES_M x WAVES_N over the BM x BN block tile
template <int BM, int BN, int WAVES_M, int WAVES_N, int PAD, int BK>
__global__ __launch_bounds__(256) void wgrad_kernel(const float *__restrict__ G, const float *__restrict__ X, int K, int M, int N,
                                                    int S, float *__restrict__ P) {
  constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);     // MFMA tiles per wave
  static_assert(WAVES_M * WAVES_N == 4, "four waves");
  __shared__ float Gs[BK][BM + PAD];
  __shared__ float Xs[BK][BN + PAD];
  const int split = blockIdx.x % S, tile = blockIdx.x / S;
  const int ntn = N / BN;
  const int tm = tile / ntn, tn = tile % ntn;
  const int kper = K / S, k0 = split * kper, nslab = kper / BK;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w % WAVES_M, wn = w / WAVES_M;
  const int lr = lane & 31, lh = lane >> 5;
  constexpr int GV = BK * BM / 4 / 256, XV = BK * BN / 4 / 256;         // float4 loads per thread per slab
  constexpr int GROW = BM / 4, XROW = BN / 4;                           // float4 per row
  f32x4 ga[GV], xa[XV];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto gload = [&](int slab) {
    const int kb = k0 + slab * BK;
#pragma unroll
    for (int i = 0; i < GV; ++i) {
      const int idx = t + 256 * i, r = idx / GROW, c4 = idx % GROW;
      ga[i] = *reinterpret_cast<const f32x4 *>(G + (size_t)(kb + r) * M + tm * BM + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = t + 256 * i, r = idx / XROW, c4 = idx % XROW;
      xa[i] = *reinterpret_cast<const f32x4 *>(X + (size_t)(kb + r) * N + tn * BN + c4 * 4);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < GV; ++i) {
      const int idx = t + 256 * i, r = idx / GROW, c4 = idx % GROW;
      *reinterpret_cast<f32x4 *>(&Gs[r][c4 * 4]) = ga[i];
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = t + 256 * i, r = idx / XROW, c4 = idx % XROW;
      *reinterpret_cast<f32x4 *>(&Xs[r][c4 * 4]) = xa[i];
    }
  };

  gload(0);
  lstore();
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    if (s + 1 < nslab) gload(s + 1);             // the next slab's global loads fly while this one is multiplied
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Gs[kk + lh][(wm * TM + i) * 32 + lr];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Xs[kk + lh][(wn * TN + j) * 32 + lr];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (s + 1 < nslab) {
      lstore();
      __syncthreads();
    }
  }
  float *out = P + (size_t)split * M * N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = tm * BM + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int col = tn * BN + (wn * TN + j) * 32 + lr;
        out[(size_t)row * N + col] = acc[i][j][e];
      }
}

__global__ void reduce_kernel(const float *P, int S, size_t mn, float *mu, float alpha) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= mn) return;
  f32x4 s = *reinterpret_cast<const f32x4 *>(P + i);
  for (int k = 1; k < S; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(P + (size_t)k * mn + i);
    s += v;
  }
  f32x4 m = *reinterpret_cast<f32x4 *>(mu + i);
  m += alpha * s;
  *reinterpret_cast<f32x4 *>(mu + i) = m;
}

__global__ void naive_kernel(const float *G, const float *X, int K, int M, int N, double *out) {
  const int m = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)G[(size_t)k * M + m] * (double)X[(size_t)k * N + n];
  out[(size_t)m * N + n] = s;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int PAD, int BK = 32>
void run(const char *name, int K, int M, int N, int S, std::vector<float *> &Gd, std::vector<float *> &Xd, float *P, float *mu,
         const std::vector<double> &ref) {
  const int tiles = (M / BM) * (N / BN), blocks = tiles * S;
  CHECK(hipMemset(mu, 0, (size_t)M * N * 4));
  wgrad_kernel<BM, BN, WAVES_M, WAVES_N, PAD, BK><<<blocks, 256>>>(Gd[0], Xd[0], K, M, N, S, P);
  reduce_kernel<<<(M * N / 4 + 255) / 256, 256>>>(P, S, (size_t)M * N, mu, 1.0f);
  CHECK(hipDeviceSynchronize());
  std::vector<float> h((size_t)M * N);
  CHECK(hipMemcpy(h.data(), mu, h.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    worst = std::fmax(worst, std::fabs(h[i] - ref[i]));
    scale = std::fmax(scale, std::fabs(ref[i]));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 200, np = (int)Gd.size();
  float ms_g = 0, ms_gr = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int r = 0; r < 10; ++r) wgrad_kernel<BM, BN, WAVES_M, WAVES_N, PAD, BK><<<blocks, 256>>>(Gd[r % np], Xd[r % np], K, M, N, S, P);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) {
      wgrad_kernel<BM, BN, WAVES_M, WAVES_N, PAD, BK><<<blocks, 256>>>(Gd[r % np], Xd[r % np], K, M, N, S, P);
      if (pass) reduce_kernel<<<(M * N / 4 + 255) / 256, 256>>>(P, S, (size_t)M * N, mu, 0.5f);
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(pass ? &ms_gr : &ms_g, e0, e1));
  }
  const double flop = 2.0 * K * M * N;
  std::printf("%-34s blocks %4d  max|err| %.2e (scale %.1f)  GEMM %6.2f us = %5.1f TFLOP/s   GEMM+reduce %6.2f us\n", name, blocks, worst, scale,
              ms_g / reps * 1e3, flop / (ms_g / reps * 1e-3) / 1e12, ms_gr / reps * 1e3);
}

int main() {
  const int K = 4096, M = 512, N = 512, NP = 8;
  std::vector<float *> Gd(NP), Xd(NP);
  std::vector<float> hg((size_t)K * M), hx((size_t)K * N);
  for (int p = 0; p < NP; ++p) {
    srand(p + 1);
    for (auto &v : hg) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto &v : hx) v = (float)rand() / RAND_MAX - 0.37f;        // asymmetric data
    CHECK(hipMalloc(&Gd[p], hg.size() * 4));
    CHECK(hipMalloc(&Xd[p], hx.size() * 4));
    CHECK(hipMemcpy(Gd[p], hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(Xd[p], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  }
  // (the last pair uploaded is pair NP-1; the reference is taken on pair 0: upload it again into slot 0's buffers is not needed, they hold it)
  double *refd;
  CHECK(hipMalloc(&refd, (size_t)M * N * 8));
  naive_kernel<<<dim3(N / 256, M), 256>>>(Gd[0], Xd[0], K, M, N, refd);
  std::vector<double> ref((size_t)M * N);
  CHECK(hipMemcpy(ref.data(), refd, ref.size() * 8, hipMemcpyDeviceToHost));
  float *P, *mu;
  CHECK(hipMalloc(&P, (size_t)16 * M * N * 4));
  CHECK(hipMalloc(&mu, (size_t)M * N * 4));
  run<128, 64, 2, 2, 4>("128x64 S=8 2x2 pad4", K, M, N, 8, Gd, Xd, P, mu, ref);
  run<64, 64, 2, 2, 0>("64x64 S=8", K, M, N, 8, Gd, Xd, P, mu, ref);
  run<64, 64, 2, 2, 4>("64x64 S=8 pad4", K, M, N, 8, Gd, Xd, P, mu, ref);
  run<64, 64, 2, 2, 0>("64x64 S=16 (1024 blocks)", K, M, N, 16, Gd, Xd, P, mu, ref);
  run<64, 64, 2, 2, 0, 64>("64x64 S=8 BK=64", K, M, N, 8, Gd, Xd, P, mu, ref);
  run<64, 64, 2, 2, 0, 16>("64x64 S=8 BK=16", K, M, N, 8, Gd, Xd, P, mu, ref);
  run<64, 128, 2, 2, 0>("64x128 S=8", K, M, N, 8, Gd, Xd, P, mu, ref);
  run<64, 128, 2, 2, 0>("64x128 S=16 (512 blocks)", K, M, N, 16, Gd, Xd, P, mu, ref);
  run<128, 64, 2, 2, 0>("128x64 S=16 (512 blocks)", K, M, N, 16, Gd, Xd, P, mu, ref);
  run<128, 128, 2, 2, 0>("128x128 S=16", K, M, N, 16, Gd, Xd, P, mu, ref);
  run<128, 128, 2, 2, 0, 16>("128x128 S=16 BK=16", K, M, N, 16, Gd, Xd, P, mu, ref);
  return 0;
}
