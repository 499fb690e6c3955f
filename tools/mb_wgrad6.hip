// Microbenchmark (round 6): structures for the K-deep fp32 MFMA product behind pn_linear_wgrad,
//   PW[split][m][n] += sum_{k in range(split)} (alpha G[k][m]) X[k][n]      (G: K x M, X: K x N row-major; K = 4096, M = N = 512)
// One templated kernel: BM x BN workgroup tile, WGM x WGN x KH waves (KH: the K halves of a slab on separate wave groups), TM x TN
// MFMA tiles (32 x 32, v_mfma_f32_32x32x2_f32) per wave, slabs of BK rows through LDS, register-staged; DBUF: two LDS buffers and
// ONE barrier per slab (else one buffer, two barriers: the round-5 kernel); PF2: global loads two slabs ahead (two register sets).
// The accumulators start from the partial tile (kh == 0 waves).  Optional in-kernel stamps (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mb_wgrad6 tools/mb_wgrad6.hip && tools/mb_wgrad6
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

struct Group {
  const float *G[8], *X[8];
  float *P[8];
  int per_pair;        // workgroups per pair
};

template <int BM, int BN, int WGM, int WGN, int KH, int BK, int SPLIT, int DBUF, int STAMP, int GLDS = 0>
__global__ __launch_bounds__(64 * WGM * WGN * KH) void wg_kernel(Group grp, int K, int M, int N, float alpha, unsigned long long *stamps) {
  const int pair = blockIdx.x / grp.per_pair, bid = blockIdx.x % grp.per_pair;
  const float *__restrict__ G = grp.G[pair];
  const float *__restrict__ X = grp.X[pair];
  float *__restrict__ PW = grp.P[pair];
  constexpr int T = 64 * WGM * WGN * KH;
  constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
  constexpr int NBUF = GLDS == 2 || DBUF == 3 ? 3 : (DBUF ? 2 : 1);      // DBUF == 4: two buffers, stores spread between the MFMAs
  constexpr int SLAB = BK * (BM + BN);                                  // floats per buffer
  constexpr int RED = (KH == 2) ? WGM * WGN * TM * TN * 16 * 64 : 0;    // staging of the odd K half's tiles
  constexpr int LDSF = (NBUF * SLAB > RED) ? NBUF * SLAB : RED;
  __shared__ float smem[LDSF];
  const int split = bid % SPLIT, tile = bid / SPLIT;
  const int ntn = N / BN;
  const int tm = tile / ntn, tn = tile % ntn;
  const int kper = K / SPLIT, k0 = split * kper, nslab = kper / BK;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int kh = w / (WGM * WGN), wm = w % WGM, wn = (w / WGM) % WGN;
  const int lr = lane & 31, lh = lane >> 5;
  constexpr int GV = BK * BM / 4 / T, XV = BK * BN / 4 / T;
  static_assert(GV * T * 4 == BK * BM && XV * T * 4 == BK * BN, "whole vectors per thread");
  constexpr int GROW = BM / 4, XROW = BN / 4;
  unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
  if (STAMP && t == 0) {
    st0 = __builtin_amdgcn_s_memtime();
    rt0 = __builtin_amdgcn_s_memrealtime();
  }
  f32x4 ga[GV], xa[XV];
  f32x16 acc[TM][TN];
  float *pw = PW + (size_t)split * M * N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = tm * BM + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int col = tn * BN + (wn * TN + j) * 32 + lr;
        acc[i][j][e] = (kh == 0 && !GLDS) ? pw[(size_t)row * N + col] : 0.f;
      }
  auto gload = [&](int slab) {
    const int kb = k0 + slab * BK;
#pragma unroll
    for (int i = 0; i < GV; ++i) {
      const int idx = t + T * i, r = idx / GROW, c4 = idx % GROW;
      ga[i] = *reinterpret_cast<const f32x4 *>(G + (size_t)(kb + r) * M + tm * BM + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = t + T * i, r = idx / XROW, c4 = idx % XROW;
      xa[i] = *reinterpret_cast<const f32x4 *>(X + (size_t)(kb + r) * N + tn * BN + c4 * 4);
    }
  };
  auto lstore = [&](int buf) {
    float *Gs = smem + buf * SLAB, *Xs = Gs + BK * BM;
#pragma unroll
    for (int i = 0; i < GV; ++i) {
      const int idx = t + T * i, r = idx / GROW, c4 = idx % GROW;
      *reinterpret_cast<f32x4 *>(&Gs[r * BM + c4 * 4]) = alpha * ga[i];
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int idx = t + T * i, r = idx / XROW, c4 = idx % XROW;
      *reinterpret_cast<f32x4 *>(&Xs[r * BN + c4 * 4]) = xa[i];
    }
  };
  auto compute = [&](int buf) {
    const float *Gs = smem + buf * SLAB, *Xs = Gs + BK * BM;
    constexpr int KW = BK / KH;
#pragma unroll
    for (int kk = 0; kk < KW; kk += 2) {
      float a[TM], b[TN];
      const int k = kh * KW + kk + lh;
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Gs[k * BM + (wm * TM + i) * 32 + lr];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Xs[k * BN + (wn * TN + j) * 32 + lr];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  // GLDS: the slab goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write); one wave-instruction fills
  // 1 KiB of LDS = 256 floats in lane order, so the LDS image is unpadded rows; alpha is applied to the finished tile
  auto gdma = [&](int slab, int buf) {
    const int kb = k0 + slab * BK;
    float *Gs = smem + buf * SLAB, *Xs = Gs + BK * BM;
    constexpr int NW = T / 64, GC = BK * BM / 256, XC = BK * BN / 256;
    static_assert(GC % NW == 0 && XC % NW == 0, "whole wave-instructions per wave");
#pragma unroll
    for (int i = 0; i < GC / NW; ++i) {
      const int c = w * (GC / NW) + i, f = c * 256 + lane * 4, r = f / BM, col = f % BM;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(G + (size_t)(kb + r) * M + tm * BM + col),
                                       (__attribute__((address_space(3))) void *)(Gs + c * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < XC / NW; ++i) {
      const int c = w * (XC / NW) + i, f = c * 256 + lane * 4, r = f / BN, col = f % BN;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(X + (size_t)(kb + r) * N + tn * BN + col),
                                       (__attribute__((address_space(3))) void *)(Xs + c * 256), 16, 0, 0);
    }
  };
  if (GLDS == 2) {
    // three LDS buffers: the DMA of slab s+2 is issued at the top of iteration s and has TWO compute phases to land; the wait at
    // the end of iteration s is counted (leaves the newest slab's DMAs in flight) and the barrier is a raw s_barrier
    constexpr int PER = (BK * BM / 256 + BK * BN / 256) / (T / 64);       // DMA instructions per wave per slab
    gdma(0, 0);
    if (nslab > 1) gdma(1, 1);
    if (nslab > 1) {
      if (PER == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      if (PER == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      if (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (STAMP && t == 0) st1 = __builtin_amdgcn_s_memtime();
    int cur = 0;
    for (int s = 0; s < nslab; ++s) {
      int nb = cur + 2;
      if (nb >= 3) nb -= 3;
      if (s + 2 < nslab) gdma(s + 2, nb);
      compute(cur);
      if (s + 2 < nslab) {
        if (PER == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (PER == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 == 3 ? 0 : cur + 1;
    }
  } else if (GLDS) {
    gdma(0, 0);
    __syncthreads();
    if (STAMP && t == 0) st1 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < nslab; ++s) {
      if (s + 1 < nslab) gdma(s + 1, (s + 1) & 1);
      compute(s & 1);
      __syncthreads();
    }
  } else {
  gload(0);
  lstore(0);
  if (DBUF && nslab > 1) gload(1);
  __syncthreads();
  if (STAMP && t == 0) st1 = __builtin_amdgcn_s_memtime();
  }
  if (GLDS) {
  } else if (DBUF == 4) {
    // two LDS buffers; the registers hold slab s+1 during iteration s; each of its vectors is stored to the other buffer BETWEEN two
    // MFMAs of this slab (spread evenly: no burst of ds_writes in front of the barrier, tools/mb_mfma_ladder.hip rows E/H) and its
    // register is refilled at once with the vector of slab s+2
    constexpr int KW = BK / KH, NK = KW / 2, NP = GV + XV;
    for (int s = 0; s < nslab; ++s) {
      const int buf = s & 1;
      const float *Gs = smem + buf * SLAB, *Xs = Gs + BK * BM;
      float *Gn = smem + (buf ^ 1) * SLAB, *Xn = Gn + BK * BM;
      const bool st = s + 1 < nslab, ld = s + 2 < nslab;
      const int kb2 = k0 + (s + 2) * BK;
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        float a[TM], b[TN];
        const int k = kh * KW + 2 * q + lh;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = Gs[k * BM + (wm * TM + i) * 32 + lr];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Xs[k * BN + (wn * TN + j) * 32 + lr];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
          if (q == ((pc + 1) * NK) / (NP + 1)) {
            if (pc < GV) {
              const int idx = t + T * pc, r = idx / GROW, c4 = idx % GROW;
              if (st) *reinterpret_cast<f32x4 *>(&Gn[r * BM + c4 * 4]) = alpha * ga[pc];
              if (ld) ga[pc] = *reinterpret_cast<const f32x4 *>(G + (size_t)(kb2 + r) * M + tm * BM + c4 * 4);
            } else {
              const int idx = t + T * (pc - GV), r = idx / XROW, c4 = idx % XROW;
              if (st) *reinterpret_cast<f32x4 *>(&Xn[r * BN + c4 * 4]) = xa[pc - GV];
              if (ld) xa[pc - GV] = *reinterpret_cast<const f32x4 *>(X + (size_t)(kb2 + r) * N + tn * BN + c4 * 4);
            }
          }
        }
      }
      __syncthreads();
    }
  } else if (DBUF == 3) {
    // three LDS buffers, register staged: slab s+2 is written to LDS at the TOP of iteration s (its loads were issued an iteration
    // ago), so the ds_writes have the whole compute phase to complete before the barrier
    if (nslab > 1) {
      lstore(1);
      if (nslab > 2) gload(2);
    }
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nslab; ++s) {
      int nb = cur + 2;
      if (nb >= 3) nb -= 3;
      if (s + 2 < nslab) {
        lstore(nb);
        if (s + 3 < nslab) gload(s + 3);
      }
      compute(cur);
      __syncthreads();
      cur = cur + 1 == 3 ? 0 : cur + 1;
    }
  } else if (DBUF) {
    for (int s = 0; s < nslab; ++s) {
      compute(s & 1);
      if (s + 1 < nslab) lstore((s + 1) & 1);
      if (s + 2 < nslab) gload(s + 2);
      __syncthreads();
    }
  } else {
    for (int s = 0; s < nslab; ++s) {
      if (s + 1 < nslab) gload(s + 1);
      compute(0);
      __syncthreads();
      if (s + 1 < nslab) {
        lstore(0);
        __syncthreads();
      }
    }
  }
  if (STAMP && t == 0) st2 = __builtin_amdgcn_s_memtime();
  if (KH == 2) {
    float(*red)[64] = reinterpret_cast<float(*)[64]>(smem);
    const int wq = w % (WGM * WGN);
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[((wq * TM + i) * TN + j) * 16 + e][lane] = acc[i][j][e];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += red[((wq * TM + i) * TN + j) * 16 + e][lane];
    }
  }
  if (kh == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = tm * BM + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
          const int col = tn * BN + (wn * TN + j) * 32 + lr;
          if (GLDS)
            pw[(size_t)row * N + col] += alpha * acc[i][j][e];
          else
            pw[(size_t)row * N + col] = acc[i][j][e];
        }
  }
  if (STAMP && t == 0) {
    unsigned long long *o = stamps + (size_t)blockIdx.x * 6;
    o[0] = st0;
    o[1] = st1;
    o[2] = st2;
    o[3] = __builtin_amdgcn_s_memtime();
    o[4] = rt0;
    o[5] = __builtin_amdgcn_s_memrealtime();
  }
}

__global__ void naive_kernel(const float *G, const float *X, int K, int M, int N, double *out) {
  const int m = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)G[(size_t)k * M + m] * (double)X[(size_t)k * N + n];
  out[(size_t)m * N + n] = s;
}

struct Ctx {
  int K, M, N;
  std::vector<float *> Gd, Xd, P;
  std::vector<double> ref;
  unsigned long long *stamps;
};

static double median(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

template <int BM, int BN, int WGM, int WGN, int KH, int BK, int SPLIT, int DBUF, int GLDS = 0>
void run(const char *name, Ctx &c) {
  const int K = c.K, M = c.M, N = c.N;
  constexpr int T = 64 * WGM * WGN * KH;
  const int blocks = (M / BM) * (N / BN) * SPLIT;
  auto kern = wg_kernel<BM, BN, WGM, WGN, KH, BK, SPLIT, DBUF, 0, GLDS>;
  auto kern_s = wg_kernel<BM, BN, WGM, WGN, KH, BK, SPLIT, DBUF, 1, GLDS>;
  // correctness: one launch on pair 0 into a zeroed partial buffer, summed over the splits on the host
  CHECK(hipMemset(c.P[0], 0, (size_t)16 * M * N * 4));
  auto grp = [&](int r, int n) {
    Group g;
    for (int i = 0; i < 8; ++i) {
      g.G[i] = c.Gd[(r * n + i) % c.Gd.size()];
      g.X[i] = c.Xd[(r * n + i) % c.Xd.size()];
      g.P[i] = c.P[i % c.P.size()];
    }
    g.per_pair = blocks;
    return g;
  };
  kern<<<blocks, T>>>(grp(0, 1), K, M, N, 1.0f, nullptr);
  CHECK(hipDeviceSynchronize());
  std::vector<float> h((size_t)SPLIT * M * N);
  CHECK(hipMemcpy(h.data(), c.P[0], h.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (size_t i = 0; i < (size_t)M * N; ++i) {
    double s = 0;
    for (int k = 0; k < SPLIT; ++k) s += h[(size_t)k * M * N + i];
    worst = std::fmax(worst, std::fabs(s - c.ref[i]));
    scale = std::fmax(scale, std::fabs(c.ref[i]));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 300, np = (int)c.Gd.size(), nl = (int)c.P.size();
  std::vector<double> us, us4;
  (void)np, (void)nl;
  for (int round = 0; round < 3; ++round) {
    for (int npair = 1; npair <= 4; npair += 3) {
      for (int r = 0; r < 20; ++r) kern<<<blocks * npair, T>>>(grp(r, npair), K, M, N, 0.5f, nullptr);
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < reps / npair; ++r) kern<<<blocks * npair, T>>>(grp(r, npair), K, M, N, 0.5f, nullptr);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      (npair == 1 ? us : us4).push_back(ms / (reps / npair * npair) * 1e3);
    }
  }
  // stamps: one launch in the middle of a train of launches
  for (int r = 0; r < 20; ++r) kern<<<blocks, T>>>(grp(r, 1), K, M, N, 0.5f, nullptr);
  kern_s<<<blocks, T>>>(grp(4, 1), K, M, N, 0.5f, c.stamps);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> hs((size_t)blocks * 6);
  CHECK(hipMemcpy(hs.data(), c.stamps, hs.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> pro, loop, epi, life;
  unsigned long long rmin = ~0ull, rmax = 0, rstart_max = 0;
  for (int b = 0; b < blocks; ++b) {
    const unsigned long long *o = &hs[(size_t)b * 6];
    pro.push_back((double)(o[1] - o[0]));
    loop.push_back((double)(o[2] - o[1]));
    epi.push_back((double)(o[3] - o[2]));
    life.push_back((double)(o[3] - o[0]));
    rmin = std::min(rmin, o[4]);
    rstart_max = std::max(rstart_max, o[4]);
    rmax = std::max(rmax, o[5]);
  }
  const double clk = median(life) / (median([&] {
                       std::vector<double> v;
                       for (int b = 0; b < blocks; ++b) v.push_back((double)(hs[(size_t)b * 6 + 5] - hs[(size_t)b * 6 + 4]));
                       return v;
                     }()) * 10.0);      // cycles per ns: s_memrealtime ticks at 100 MHz
  const double flop = 2.0 * K * M * N;
  std::sort(us.begin(), us.end());
  std::sort(us4.begin(), us4.end());
  std::printf("%-44s thr %3d blocks %4d err %.1e | 4 pairs per launch: %6.2f us per pair = %5.1f TF | one: %6.2f us (min %6.2f) = %5.1f TF | cycles: prologue %6.0f loop %6.0f epilogue %6.0f life %6.0f | "
              "first start -> last end %5.2f us, starts spread %5.2f us, clock %.2f GHz\n",
              name, T, blocks, worst / scale, us4[1], flop / (us4[1] * 1e-6) / 1e12, us[1], us[0], flop / (us[1] * 1e-6) / 1e12, median(pro), median(loop), median(epi), median(life),
              (rmax - rmin) / 100.0, (rstart_max - rmin) / 100.0, clk);
  std::fflush(stdout);
}

int main() {
  Ctx c;
  c.K = 4096, c.M = 512, c.N = 512;
  const int K = c.K, M = c.M, N = c.N, NP = 8, L = 4;
  c.Gd.resize(NP), c.Xd.resize(NP), c.P.resize(L);
  std::vector<float> hg((size_t)K * M), hx((size_t)K * N);
  for (int p = NP - 1; p >= 0; --p) {
    srand(p + 1);
    for (auto &v : hg) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto &v : hx) v = (float)rand() / RAND_MAX - 0.37f;        // asymmetric data
    CHECK(hipMalloc(&c.Gd[p], hg.size() * 4));
    CHECK(hipMalloc(&c.Xd[p], hx.size() * 4));
    CHECK(hipMemcpy(c.Gd[p], hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(c.Xd[p], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  }
  double *refd;
  CHECK(hipMalloc(&refd, (size_t)M * N * 8));
  naive_kernel<<<dim3(N / 256, M), 256>>>(c.Gd[0], c.Xd[0], K, M, N, refd);
  c.ref.resize((size_t)M * N);
  CHECK(hipMemcpy(c.ref.data(), refd, c.ref.size() * 8, hipMemcpyDeviceToHost));
  for (int l = 0; l < L; ++l) {
    CHECK(hipMalloc(&c.P[l], (size_t)16 * M * N * 4));
    CHECK(hipMemset(c.P[l], 0, (size_t)16 * M * N * 4));
  }
  CHECK(hipMalloc(&c.stamps, (size_t)4096 * 6 * 8));
  for (int pass = 0; pass < 2; ++pass) {
    std::printf("--- pass %d\n", pass);
    //   BM   BN  WGM WGN KH BK SPLIT DBUF GLDS
    run<64, 64, 2, 2, 2, 32, 8, 0>("64x64 8w(kh2) bk32 s8 1buf  [round-5 form]", c);
    run<64, 64, 2, 2, 2, 32, 8, 1>("64x64 8w(kh2) bk32 s8 dbuf", c);
    run<64, 64, 2, 2, 2, 32, 8, 4>("64x64 8w(kh2) bk32 s8 dbuf spread", c);
    run<64, 64, 2, 2, 2, 64, 8, 4>("64x64 8w(kh2) bk64 s8 dbuf spread", c);
    run<64, 64, 2, 2, 1, 32, 8, 4>("64x64 4w bk32 s8 dbuf spread", c);
    run<64, 64, 2, 2, 1, 64, 8, 4>("64x64 4w bk64 s8 dbuf spread", c);
    run<128, 64, 2, 2, 2, 32, 8, 4>("128x64 8w(64x32,kh2) bk32 s8 dbuf spread", c);
    run<128, 64, 4, 2, 1, 32, 8, 4>("128x64 8w(32x32) bk32 s8 dbuf spread", c);
    run<128, 128, 2, 2, 1, 32, 16, 4>("128x128 4w(64x64) bk32 s16 dbuf spread", c);
    run<128, 128, 2, 4, 1, 32, 16, 4>("128x128 8w(64x32) bk32 s16 dbuf spread", c);
    run<128, 128, 2, 2, 2, 32, 16, 4>("128x128 8w(64x64,kh2) bk32 s16 dbuf spread", c);
    run<64, 64, 2, 2, 2, 32, 4, 4>("64x64 8w(kh2) bk32 s4 dbuf spread (256 blocks)", c);
    run<64, 64, 2, 2, 2, 32, 16, 4>("64x64 8w(kh2) bk32 s16 dbuf spread (1024 blocks)", c);
  }
  return 0;
}
