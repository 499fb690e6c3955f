// Microbenchmark (round 6): what each ingredient of an LDS-staged fp32 MFMA loop costs on gfx950, one ingredient at a time.
//   A  dependent chain of v_mfma_f32_32x32x2_f32, operands in registers
//   B  + both operands read from LDS in front of every MFMA (2 x ds_read_b32), no barrier
//   C  + a workgroup barrier every 8 MFMAs
//   D  + two ds_write_b128 per thread every 8 MFMAs (register data), second barrier as in the one-buffer loop
//   E  as D with one barrier (two LDS buffers)
//   F  E + the global loads (2 x dwordx4 per thread per slab from a 2 MB window: L2-resident)
// 512 workgroups x 512 threads (2 per CU, 4 waves per SIMD) and 256-thread variants; 128 slabs of 8 MFMAs per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mb_mfma_ladder tools/mb_mfma_ladder.hip && tools/mb_mfma_ladder
#include <hip/hip_runtime.h>

#include <algorithm>
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

template <int MODE, int T, int NACC>
__global__ __launch_bounds__(T) void ladder(const float *__restrict__ src, float *__restrict__ out, int nslab) {
  __shared__ float smem[2][2][32][64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = w & 1, wn = (w >> 1) & 1, kh = (w >> 2) & 1;
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  // fill LDS once
  for (int i = t; i < 2 * 2 * 32 * 64; i += T) (&smem[0][0][0][0])[i] = (float)((i * 7) % 13) * 0.01f;
  __syncthreads();
  float a = 0.5f + lane * 0.001f, b = 0.25f - lane * 0.002f;
  f32x4 ga = {1.f, 2.f, 3.f, 4.f}, xa = {0.5f, 0.25f, 0.125f, 0.0625f};
  const int lrow = (t / 16) % 32, lc4 = t % 16;
  const float *gp = src + (size_t)(blockIdx.x % 8) * (512 * 1024 / 4) + (size_t)lrow * 512 + (blockIdx.x / 8 % 8) * 64 + lc4 * 4;
  for (int s = 0; s < nslab; ++s) {
    const int buf = (MODE >= 4) ? (s & 1) : 0;
    if (MODE == 5 || MODE == 9) {
      ga = *reinterpret_cast<const f32x4 *>(gp + (size_t)(s % 16) * 32 * 512);
      xa = *reinterpret_cast<const f32x4 *>(gp + 262144 + (size_t)(s % 16) * 32 * 512);
    }
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      if (MODE >= 1) {
        a = smem[buf][0][kh * 16 + kk + lh][wm * 32 + lr];
        b = smem[buf][1][kh * 16 + kk + lh][wn * 32 + lr];
      }
      acc[(kk / 2) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(kk / 2) % NACC], 0, 0, 0);
      if (MODE == 7 || MODE == 9 || MODE == 10) {          // the two LDS stores spread between the MFMAs instead of a burst in front of the barrier
        if (kk == 4) {
          *reinterpret_cast<f32x4 *>(&smem[buf ^ 1][0][lrow][lc4 * 4]) = ga;
          if (MODE == 10) ga = *reinterpret_cast<const f32x4 *>(gp + (size_t)((s + 2) % 16) * 32 * 512);      // refilled at once: a whole slab to land
        }
        if (kk == 10) {
          *reinterpret_cast<f32x4 *>(&smem[buf ^ 1][1][lrow][lc4 * 4]) = xa;
          if (MODE == 10) xa = *reinterpret_cast<const f32x4 *>(gp + 262144 + (size_t)((s + 2) % 16) * 32 * 512);
        }
      }
    }
    if (MODE == 7 || MODE == 9 || MODE == 10) __syncthreads();
    if (MODE == 11) {                        // burst of stores in front of the barrier (E), loads issued after the stores for the slab after next
      *reinterpret_cast<f32x4 *>(&smem[buf ^ 1][0][lrow][lc4 * 4]) = ga;
      *reinterpret_cast<f32x4 *>(&smem[buf ^ 1][1][lrow][lc4 * 4]) = xa;
      ga = *reinterpret_cast<const f32x4 *>(gp + (size_t)((s + 2) % 16) * 32 * 512);
      xa = *reinterpret_cast<const f32x4 *>(gp + 262144 + (size_t)((s + 2) % 16) * 32 * 512);
      __syncthreads();
    }
    if (MODE == 8) {                         // half the store volume, burst
      *reinterpret_cast<f32x4 *>(&smem[buf ^ 1][0][lrow][lc4 * 4]) = ga;
      __syncthreads();
    }
    if (MODE == 6) {                         // LDS-DMA instead of ds_write (L2-resident source), wait + barrier
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp + (size_t)(s % 16) * 32 * 512),
                                       (__attribute__((address_space(3))) void *)(&smem[buf ^ 1][0][0][0] + w * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp + 262144 + (size_t)(s % 16) * 32 * 512),
                                       (__attribute__((address_space(3))) void *)(&smem[buf ^ 1][1][0][0] + w * 256), 16, 0, 0);
      __syncthreads();
    }
    if (MODE == 2) __syncthreads();
    if (MODE == 3) {
      __syncthreads();
      *reinterpret_cast<f32x4 *>(&smem[0][0][lrow][lc4 * 4]) = ga;
      *reinterpret_cast<f32x4 *>(&smem[0][1][lrow][lc4 * 4]) = xa;
      __syncthreads();
    }
    if (MODE == 4 || MODE == 5) {
      *reinterpret_cast<f32x4 *>(&smem[buf ^ 1][0][lrow][lc4 * 4]) = ga;
      *reinterpret_cast<f32x4 *>(&smem[buf ^ 1][1][lrow][lc4 * 4]) = xa;
      __syncthreads();
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += acc[i][e];
  if (sum == 123.456f) out[blockIdx.x * T + t] = sum;
}

template <int MODE, int T, int NACC>
void run(const char *name, const float *src, float *out, int blocks) {
  const int nslab = 128;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<double> us;
  for (int round = 0; round < 3; ++round) {
    for (int r = 0; r < 5; ++r) ladder<MODE, T, NACC><<<blocks, T>>>(src, out, nslab);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 50; ++r) ladder<MODE, T, NACC><<<blocks, T>>>(src, out, nslab);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    us.push_back(ms / 50 * 1e3);
  }
  std::sort(us.begin(), us.end());
  const double flop = (double)blocks * (T / 64) * nslab * 8 * 2.0 * 32 * 32 * 2;
  std::printf("%-64s blocks %4d x %3d thr: %8.2f us = %6.1f TFLOP/s = %.3f of 157.3\n", name, blocks, T, us[1], flop / (us[1] * 1e-6) / 1e12,
              flop / (us[1] * 1e-6) / 1e12 / 157.3);
  std::fflush(stdout);
}

// Producer / consumer waves: 8 compute waves (LDS reads + MFMA only) and NL loader waves that move the next slab global -> LDS
// (DMA == 0: registers + ds_write_b128; DMA == 1: global_load_lds_dwordx4), two LDS buffers, one barrier per slab for all waves.
template <int NL, int DMA>
__global__ __launch_bounds__(512 + 64 * NL) void ladder_split(const float *__restrict__ src, float *__restrict__ out, int nslab) {
  __shared__ float smem[2][2][32][64];
  constexpr int T = 512 + 64 * NL;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int wm = w & 1, wn = (w >> 1) & 1, kh = (w >> 2) & 1;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int i = t; i < 2 * 2 * 32 * 64; i += T) (&smem[0][0][0][0])[i] = (float)((i * 7) % 13) * 0.01f;
  __syncthreads();
  const float *base = src + (size_t)(blockIdx.x % 8) * (512 * 1024 / 4) + (blockIdx.x / 8 % 8) * 64;
  if (w < 8) {
    for (int s = 0; s < nslab; ++s) {
      const int buf = s & 1;
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        const float a = smem[buf][0][kh * 16 + kk + lh][wm * 32 + lr];
        const float b = smem[buf][1][kh * 16 + kk + lh][wn * 32 + lr];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      __syncthreads();
    }
  } else {
    const int lt = t - 512;                       // 0 .. 64 NL - 1
    constexpr int PER = 1024 / (64 * NL);         // 16-byte vectors per loader thread per slab (G 512 + X 512)
    for (int s = 0; s < nslab; ++s) {
      const int buf = s & 1;
      if (DMA) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int c = (lt >> 6) * PER + i;      // 1 KiB chunk of the slab pair: 4 rows of one operand
          const int op = c / 8, r = (c % 8) * 4 + (lane >> 4), c4 = lane & 15;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + op * 262144 + (size_t)((s % 16) * 32 + r) * 512 + c4 * 4),
                                           (__attribute__((address_space(3))) void *)(&smem[buf ^ 1][op][(c % 8) * 4][0]), 16, 0, 0);
        }
      } else {
        f32x4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int idx = lt + 64 * NL * i, op = idx / 512, r = (idx % 512) / 16, c4 = idx % 16;
          v[i] = *reinterpret_cast<const f32x4 *>(base + op * 262144 + (size_t)((s % 16) * 32 + r) * 512 + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int idx = lt + 64 * NL * i, op = idx / 512, r = (idx % 512) / 16, c4 = idx % 16;
          *reinterpret_cast<f32x4 *>(&smem[buf ^ 1][op][r][c4 * 4]) = v[i];
        }
      }
      __syncthreads();
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) sum += acc[e];
  if (sum == 123.456f) out[blockIdx.x * T + t] = sum;
}

template <int NL, int DMA>
void run_split(const char *name, const float *src, float *out, int blocks) {
  const int nslab = 128;
  constexpr int T = 512 + 64 * NL;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::vector<double> us;
  for (int round = 0; round < 3; ++round) {
    for (int r = 0; r < 5; ++r) ladder_split<NL, DMA><<<blocks, T>>>(src, out, nslab);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 50; ++r) ladder_split<NL, DMA><<<blocks, T>>>(src, out, nslab);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    us.push_back(ms / 50 * 1e3);
  }
  std::sort(us.begin(), us.end());
  const double flop = (double)blocks * 8 * nslab * 8 * 2.0 * 32 * 32 * 2;
  std::printf("%-64s blocks %4d x %3d thr: %8.2f us = %6.1f TFLOP/s = %.3f of 157.3\n", name, blocks, T, us[1], flop / (us[1] * 1e-6) / 1e12,
              flop / (us[1] * 1e-6) / 1e12 / 157.3);
  std::fflush(stdout);
}

int main() {
  float *src, *out;
  CHECK(hipMalloc(&src, 8u << 20));
  CHECK(hipMemset(src, 0, 8u << 20));
  CHECK(hipMalloc(&out, 4u << 20));
  for (int pass = 0; pass < 2; ++pass) {
    std::printf("--- pass %d\n", pass);
    run<0, 512, 1>("A  MFMA chain, register operands", src, out, 512);
    run<0, 512, 2>("A2 two accumulators", src, out, 512);
    run<0, 256, 1>("A  (256 threads, 1024 blocks: 4 per CU)", src, out, 1024);
    run<0, 256, 1>("A  (256 threads, 256 blocks: one wave per SIMD)", src, out, 256);
    run<0, 256, 4>("A4 (256 threads, 256 blocks: one wave per SIMD, 4 accumulators)", src, out, 256);
    run<1, 512, 1>("B  + operands from LDS", src, out, 512);
    run<1, 512, 2>("B2 + operands from LDS, two accumulators", src, out, 512);
    run<2, 512, 1>("C  + barrier per 8 MFMAs", src, out, 512);
    run<3, 512, 1>("D  + 2 ds_write_b128, two barriers (one-buffer loop)", src, out, 512);
    run<4, 512, 1>("E  + 2 ds_write_b128, one barrier (two buffers)", src, out, 512);
    run<5, 512, 1>("F  E + global loads (L2)", src, out, 512);
    run<6, 512, 1>("G  LDS-DMA (global_load_lds) instead of the ds_writes, one barrier", src, out, 512);
    run<7, 512, 1>("H  E with the two ds_writes spread between the MFMAs", src, out, 512);
    run<8, 512, 1>("I  E with half the store volume (one ds_write_b128)", src, out, 512);
    run<9, 512, 1>("J  H + global loads at the top of the slab (consumed in the same slab)", src, out, 512);
    run<10, 512, 1>("K  H + global loads issued right after each store (a slab to land)", src, out, 512);
    run<11, 512, 1>("L  E + global loads issued after the stores (the shipped loop)", src, out, 512);
    run<10, 512, 1>("K  768 blocks", src, out, 768);
    run<9, 256, 1>("J  256 threads, 1024 blocks", src, out, 1024);
    run<7, 256, 1>("H  256 threads, 256 blocks (one wave per SIMD)", src, out, 256);
    run<4, 256, 1>("E  256 threads, 256 blocks (one wave per SIMD)", src, out, 256);
    run<2, 256, 1>("C  256 threads, 256 blocks (one wave per SIMD)", src, out, 256);
    run_split<2, 0>("M  8 compute waves + 2 loader waves (registers + ds_write)", src, out, 512);
    run_split<1, 0>("M  8 compute waves + 1 loader wave  (registers + ds_write)", src, out, 512);
    run_split<4, 0>("M  8 compute waves + 4 loader waves (registers + ds_write)", src, out, 512);
    run_split<2, 1>("N  8 compute waves + 2 loader waves (LDS-DMA)", src, out, 512);
    run_split<1, 1>("N  8 compute waves + 1 loader wave  (LDS-DMA)", src, out, 512);
    run_split<4, 1>("N  8 compute waves + 4 loader waves (LDS-DMA)", src, out, 512);
    run_split<2, 1>("N  2 loader waves (LDS-DMA), 768 blocks", src, out, 768);
    run<5, 512, 1>("F  768 blocks (3 per CU)", src, out, 768);
    run<5, 512, 1>("F  1024 blocks (4 per CU)", src, out, 1024);
    run<5, 256, 1>("F  256 threads, 1024 blocks", src, out, 1024);
  }
  return 0;
}
