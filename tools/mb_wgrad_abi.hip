// Times pn_linear_wgrad through the C ABI from a C++ loop (no Python between two launches), operands as tools/mb_wgrad.hip
// makes them.   hipcc --offload-arch=gfx950 -O3 -Iinclude -o tools/mb_wgrad_abi tools/mb_wgrad_abi.hip -Lpnode_amd/lib -lpnode_amd
//               LD_LIBRARY_PATH=pnode_amd/lib tools/mb_wgrad_abi
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pnode_amd.h"

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      std::exit(1);                                                                           \
    }                                                                                         \
  } while (0)

template <typename T>
int run_all(int DT) {
  const int K = 4096, M = 512, N = 512, NP = 8, L = 4;
  const double peak_scale = 1.0;
  (void)peak_scale;
  std::vector<T *> Gd(NP), Xd(NP);
  std::vector<T> hg((size_t)K * M), hx((size_t)K * N);
  for (int p = 0; p < NP; ++p) {
    srand(p + 1);
    for (auto &v : hg) v = (T)((float)rand() / RAND_MAX - 0.5f);
    for (auto &v : hx) v = (T)((float)rand() / RAND_MAX - 0.37f);
    CHECK(hipMalloc(&Gd[p], hg.size() * sizeof(T)));
    CHECK(hipMalloc(&Xd[p], hx.size() * sizeof(T)));
    CHECK(hipMemcpy(Gd[p], hg.data(), hg.size() * sizeof(T), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(Xd[p], hx.data(), hx.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  int64_t nb = 0;
  const int64_t nw = pn_linear_wgrad_work_bytes(DT, M, N, &nb);
  std::vector<void *> pw(L), pb(L);
  for (int l = 0; l < L; ++l) {
    CHECK(hipMalloc(&pw[l], nw));
    CHECK(hipMalloc(&pb[l], nb));
    CHECK(hipMemset(pw[l], 0, nw));
    CHECK(hipMemset(pb[l], 0, nb));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int bias = 0; bias < 2; ++bias)
    for (int layers = 1; layers <= L; layers *= 4) {
      const int reps = 400;
      for (int r = 0; r < 20; ++r) pn_linear_wgrad(nullptr, DT, K, M, N, Gd[r % NP], Xd[r % NP], 1.0, pw[r % layers], bias ? pb[r % layers] : nullptr);
      CHECK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r)
        if (pn_linear_wgrad(nullptr, DT, K, M, N, Gd[r % NP], Xd[r % NP], 1.0, pw[r % layers], bias ? pb[r % layers] : nullptr)) return 1;
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("%s pn_linear_wgrad %s, %d partial buffer(s): %6.2f us = %5.1f TFLOP/s\n", DT == PN_F32 ? "fp32" : "fp64", bias ? "dW + db" : "dW only", layers, ms / reps * 1e3,
                  2.0 * K * M * N / (ms / reps * 1e-3) / 1e12);
    }
  // grouped launches (round 6): the four layers of a stage VJP in ONE launch
  for (int bias = 0; bias < 2; ++bias) {
    const int reps = 100;
    auto group = [&](int r) {
      pn_wgrad_pair q[4];
      for (int l = 0; l < L; ++l) {
        q[l].g = Gd[(4 * r + l) % NP], q[l].x = Xd[(4 * r + l + 1) % NP], q[l].pw = pw[l], q[l].pb = bias ? pb[l] : nullptr;
        q[l].alpha = 1.0, q[l].out_f = M, q[l].in_f = N;
      }
      return pn_linear_wgrad_group(nullptr, DT, K, L, q, (std::getenv("MB_EXACT") ? PN_WGRAD_EXACT_FP32 : 0) | (std::getenv("MB_TILE64") ? PN_WGRAD_TILE_64 : 0));
    };
    for (int r = 0; r < 10; ++r) group(r);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r)
      if (group(r)) return 1;
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%s pn_linear_wgrad_group %s, 4 pairs per launch: %6.2f us per pair = %5.1f TFLOP/s\n", DT == PN_F32 ? "fp32" : "fp64", bias ? "dW + db" : "dW only", ms / reps / L * 1e3,
                2.0 * K * M * N / (ms / reps / L * 1e-3) / 1e12);
  }
  return 0;
}

int main() {
  if (std::getenv("MB_F64")) return run_all<double>(PN_F64);
  return run_all<float>(PN_F32);
}
