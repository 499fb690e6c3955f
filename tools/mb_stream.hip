// tools/mb_stream.hip -- standalone microbenchmark behind the launch/store-policy choices of
// pnode_amd/csrc/pn_kernels.hip.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb_stream.hip -o tools/mb_stream
//   ./tools/mb_stream            (on the GPU box)
// Measures, with start/stop events bound to each dispatch, the duration of a 3-vector
// (y = u + c*k) and a 7-vector (lambda update) streaming kernel on 8 MiB fp32 vectors for
// several store/load policies, in three cache regimes:
//   hot   : the same buffers every launch (everything resident in the 256 MiB Infinity Cache)
//   prod  : k freshly written by a producer kernel, u from the previous launch, y into a fresh
//           slot of a 3 GiB slab (what the forward sweep does)
//   cold  : every operand cycles through 3 GiB slabs (HBM)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

enum Policy { PLAIN = 0, NT_STORE, WT_STORE, NT_BOTH, NT_LOAD };
static const char *pname[] = {"plain", "nt-store", "wt-store(sc0 sc1)", "nt-load+nt-store", "nt-load"};

template <int POL>
__device__ __forceinline__ f4 ld(const f4 *p) {
  if (POL == NT_BOTH || POL == NT_LOAD) return __builtin_nontemporal_load(p);
  return *p;
}
template <int POL>
__device__ __forceinline__ void st(f4 *p, f4 v) {
  if (POL == NT_STORE || POL == NT_BOTH) {
    __builtin_nontemporal_store(v, p);
  } else if (POL == WT_STORE) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  } else {
    *p = v;
  }
}

template <int NIN, int VPT, int POL, int BLOCK>
__global__ __launch_bounds__(BLOCK) void lin(const f4 *const *xs_unused, const f4 *x0, const f4 *x1, const f4 *x2,
                                             const f4 *x3, const f4 *x4, const f4 *x5, f4 *out, float c,
                                             long nvec) {
  const f4 *x[6] = {x0, x1, x2, x3, x4, x5};
  long base = (long)blockIdx.x * (BLOCK * VPT) + threadIdx.x;
  f4 r[VPT][NIN];
#pragma unroll
  for (int p = 0; p < VPT; ++p) {
    long i = base + (long)p * BLOCK;
    if (i < nvec) {
#pragma unroll
      for (int j = 0; j < NIN; ++j) r[p][j] = ld<POL>(x[j] + i);
    }
  }
#pragma unroll
  for (int p = 0; p < VPT; ++p) {
    long i = base + (long)p * BLOCK;
    if (i < nvec) {
      f4 acc = r[p][0];
#pragma unroll
      for (int j = 1; j < NIN; ++j) acc += c * r[p][j];
      st<POL>(out + i, acc);
    }
  }
  if (POL == WT_STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void fill(f4 *p, float v, long nvec) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nvec) p[i] = f4{v, v, v, v};
}
__global__ void empty_kernel() {}

struct Timer {
  std::vector<hipEvent_t> e0, e1;
  void add(hipEvent_t a, hipEvent_t b) { e0.push_back(a); e1.push_back(b); }
  void report(const char *label) {
    std::vector<float> us;
    for (size_t i = 0; i < e0.size(); ++i) {
      float ms;
      CK(hipEventSynchronize(e1[i]));
      CK(hipEventElapsedTime(&ms, e0[i], e1[i]));
      us.push_back(ms * 1e3f);
      CK(hipEventDestroy(e0[i]));
      CK(hipEventDestroy(e1[i]));
    }
    std::sort(us.begin(), us.end());
    double mean = 0;
    for (float v : us) mean += v;
    mean /= us.size();
    printf("%-64s median %6.2f us  mean %6.2f  min %6.2f  p90 %6.2f\n", label, us[us.size() / 2], mean, us[0],
           us[us.size() * 9 / 10]);
    e0.clear(); e1.clear();
  }
};

constexpr long N = 4096L * 512;       // elements per vector
constexpr long NVEC = N / 4;
constexpr long SLOTS = 384;           // 384 * 8 MiB = 3 GiB slab

template <int NIN, int VPT, int POL, int BLOCK>
void run(const char *regime, float *slabA, float *slabB, hipStream_t st, int reps) {
  Timer T;
  long per = (long)BLOCK * VPT;
  dim3 grid((unsigned)((NVEC + per - 1) / per));
  auto slot = [&](float *slab, long s) { return (f4 *)(slab + (s % SLOTS) * N); };
  for (int r = 0; r < reps + 5; ++r) {
    const f4 *x[6];
    f4 *out;
    if (regime[0] == 'h') {                 // hot
      for (int j = 0; j < 6; ++j) x[j] = slot(slabA, j);
      out = slot(slabA, 7);
    } else if (regime[0] == 'p') {          // producer-fresh inputs, fresh output slot
      x[0] = slot(slabB, r);                // u: written by the previous launch's output
      for (int j = 1; j < 6; ++j) x[j] = slot(slabA, j);
      hipLaunchKernelGGL(fill, dim3((NVEC + 255) / 256), dim3(256), 0, st, (f4 *)x[1], 0.5f, NVEC);
      out = slot(slabB, r + 1);
    } else {                                // cold
      for (int j = 0; j < 6; ++j) x[j] = slot(slabA, (long)r * 7 + j);
      out = slot(slabB, r);
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipExtLaunchKernelGGL((lin<NIN, VPT, POL, BLOCK>), grid, dim3(BLOCK), 0, st, a, b, 0, (const f4 *const *)nullptr,
                          x[0], x[1], x[2], x[3], x[4], x[5], out, 0.25f, NVEC);
    if (r >= 5) T.add(a, b);
    else { CK(hipEventSynchronize(b)); CK(hipEventDestroy(a)); CK(hipEventDestroy(b)); }
  }
  char label[160];
  snprintf(label, sizeof label, "%-5s nin=%d (%2d MiB moved) vpt=%d block=%4d %-18s", regime, NIN, (NIN + 1) * 8, VPT,
           BLOCK, pname[POL]);
  T.report(label);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *A, *B;
  CK(hipMalloc(&A, SLOTS * N * sizeof(float)));
  CK(hipMalloc(&B, SLOTS * N * sizeof(float)));
  CK(hipMemsetAsync(A, 0, SLOTS * N * sizeof(float), st));
  CK(hipMemsetAsync(B, 0, SLOTS * N * sizeof(float), st));
  CK(hipStreamSynchronize(st));
  {
    Timer T;
    for (int r = 0; r < 50; ++r) {
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      hipExtLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, st, a, b, 0);
      T.add(a, b);
    }
    T.report("empty kernel, 1024 x 256");
  }
  const int reps = 60;
  const char *regimes[] = {"hot", "prod", "cold"};
  for (const char *rg : regimes) {
    run<2, 2, PLAIN, 256>(rg, A, B, st, reps);
    run<2, 2, NT_STORE, 256>(rg, A, B, st, reps);
    run<2, 2, WT_STORE, 256>(rg, A, B, st, reps);
    run<2, 2, NT_BOTH, 256>(rg, A, B, st, reps);
    run<2, 2, NT_LOAD, 256>(rg, A, B, st, reps);
    run<2, 1, PLAIN, 256>(rg, A, B, st, reps);
    run<2, 4, PLAIN, 256>(rg, A, B, st, reps);
    run<2, 1, PLAIN, 1024>(rg, A, B, st, reps);
    run<2, 4, NT_STORE, 256>(rg, A, B, st, reps);
    run<6, 2, PLAIN, 256>(rg, A, B, st, reps);
    run<6, 2, NT_STORE, 256>(rg, A, B, st, reps);
    run<6, 2, WT_STORE, 256>(rg, A, B, st, reps);
    run<6, 2, NT_BOTH, 256>(rg, A, B, st, reps);
    run<6, 1, PLAIN, 256>(rg, A, B, st, reps);
    run<1, 2, PLAIN, 256>(rg, A, B, st, reps);
    run<1, 2, NT_STORE, 256>(rg, A, B, st, reps);
    printf("\n");
  }
  return 0;
}
