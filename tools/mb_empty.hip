// tools/mb_empty.hip -- dispatch floor of a kernel launch vs grid size (start/stop events bound to the dispatch).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){fprintf(stderr,"%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(e_)); exit(1);} } while(0)
__global__ void empty_kernel() {}
__global__ void touch_kernel(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1.f; }
typedef float f4 __attribute__((ext_vector_type(4)));
// grid-stride persistent streaming kernel: y = a + c*b
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_gs(const f4 *a, const f4 *b, f4 *y, float c, long nvec) {
  long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride * UNROLL) {
    f4 ra[UNROLL], rb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { long k = i + u * stride; if (k < nvec) { ra[u] = a[k]; rb[u] = b[k]; } }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { long k = i + u * stride; if (k < nvec) y[k] = ra[u] + c * rb[u]; }
  }
}
static void rep(const char *label, std::vector<float> &us) {
  std::sort(us.begin(), us.end());
  double m = 0; for (float v : us) m += v; m /= us.size();
  printf("%-52s median %6.2f us  mean %6.2f  min %6.2f\n", label, us[us.size()/2], m, us[0]);
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float *buf; const long N = 4096L*512; const long SL = 256;
  CK(hipMalloc(&buf, SL * N * 4 * 3)); CK(hipMemset(buf, 0, SL * N * 4 * 3));
  int grids[] = {1, 32, 256, 512, 1024, 2048, 4096, 8192};
  for (int blk : {64, 256, 1024}) for (int g : grids) {
    std::vector<float> us;
    for (int r = 0; r < 40; ++r) {
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      hipExtLaunchKernelGGL(empty_kernel, dim3(g), dim3(blk), 0, st, a, b, 0);
      CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r >= 5) us.push_back(ms * 1e3f);
      CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    }
    char l[128]; snprintf(l, sizeof l, "empty grid=%5d block=%4d (back-to-back: no)", g, blk); rep(l, us);
  }
  // back-to-back launches (queue kept busy), events on each
  for (int g : {256, 1024, 2048}) {
    std::vector<hipEvent_t> ea, eb; std::vector<float> us;
    for (int r = 0; r < 60; ++r) { hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      hipExtLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), 0, st, a, b, 0); ea.push_back(a); eb.push_back(b); }
    for (size_t i = 0; i < ea.size(); ++i) { CK(hipEventSynchronize(eb[i])); float ms; CK(hipEventElapsedTime(&ms, ea[i], eb[i])); if (i >= 5) us.push_back(ms*1e3f); }
    char l[128]; snprintf(l, sizeof l, "empty grid=%5d block= 256 (back-to-back: yes)", g); rep(l, us);
  }
  // streaming kernel, cold operands, grid sweep (grid-stride) 
  const long nvec = N / 4;
  for (int g : {256, 512, 1024, 2048}) for (int un : {1, 2, 4}) {
    std::vector<hipEvent_t> ea, eb; std::vector<float> us;
    for (int r = 0; r < 60; ++r) {
      f4 *a = (f4*)(buf + ((3L*r) % (3*SL)) * N), *b = (f4*)(buf + ((3L*r+1) % (3*SL)) * N), *y = (f4*)(buf + ((3L*r+2) % (3*SL)) * N);
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      if (un == 1) hipExtLaunchKernelGGL(stream_gs<1>, dim3(g), dim3(256), 0, st, e0, e1, 0, (const f4*)a, (const f4*)b, y, 0.5f, nvec);
      if (un == 2) hipExtLaunchKernelGGL(stream_gs<2>, dim3(g), dim3(256), 0, st, e0, e1, 0, (const f4*)a, (const f4*)b, y, 0.5f, nvec);
      if (un == 4) hipExtLaunchKernelGGL(stream_gs<4>, dim3(g), dim3(256), 0, st, e0, e1, 0, (const f4*)a, (const f4*)b, y, 0.5f, nvec);
      ea.push_back(e0); eb.push_back(e1);
    }
    for (size_t i = 0; i < ea.size(); ++i) { CK(hipEventSynchronize(eb[i])); float ms; CK(hipEventElapsedTime(&ms, ea[i], eb[i])); if (i >= 5) us.push_back(ms*1e3f); }
    char l[128]; snprintf(l, sizeof l, "stream 24MiB cold grid=%5d x256 unroll=%d", g, un); rep(l, us);
  }
  return 0;
}
