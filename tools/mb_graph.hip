// tools/mb_graph.hip -- does a kernel's measured duration depend on how it is launched?
// Launches (a) plain, (b) captured in a hipGraph; run under rocprofv3 --kernel-trace to read durations.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_=(x); if(e_!=hipSuccess){fprintf(stderr,"%s:%d %s\n",__FILE__,__LINE__,hipGetErrorString(e_)); exit(1);} } while(0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void empty_plain() {}
__global__ void empty_graph() {}
__global__ __launch_bounds__(256) void stream_plain(const f4 *a, const f4 *b, f4 *y, float c, long nvec) {
  long i = (long)blockIdx.x * 512 + threadIdx.x; f4 ra[2], rb[2];
  for (int u = 0; u < 2; ++u) { long k = i + u * 256; if (k < nvec) { ra[u] = a[k]; rb[u] = b[k]; } }
  for (int u = 0; u < 2; ++u) { long k = i + u * 256; if (k < nvec) y[k] = ra[u] + c * rb[u]; }
}
__global__ __launch_bounds__(256) void stream_graph(const f4 *a, const f4 *b, f4 *y, float c, long nvec) {
  long i = (long)blockIdx.x * 512 + threadIdx.x; f4 ra[2], rb[2];
  for (int u = 0; u < 2; ++u) { long k = i + u * 256; if (k < nvec) { ra[u] = a[k]; rb[u] = b[k]; } }
  for (int u = 0; u < 2; ++u) { long k = i + u * 256; if (k < nvec) y[k] = ra[u] + c * rb[u]; }
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const long N = 4096L * 512, nvec = N / 4, SL = 200;
  float *buf; CK(hipMalloc(&buf, SL * N * 4)); CK(hipMemset(buf, 0, SL * N * 4));
  auto slot = [&](long s) { return (f4 *)(buf + (s % SL) * N); };
  for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(empty_plain, dim3(1024), dim3(256), 0, st);
  for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(stream_plain, dim3(1024), dim3(256), 0, st, (const f4*)slot(3*r), (const f4*)slot(3*r+1), slot(3*r+2), 0.5f, nvec);
  CK(hipStreamSynchronize(st));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int r = 0; r < 25; ++r) hipLaunchKernelGGL(empty_graph, dim3(1024), dim3(256), 0, st);
  for (int r = 0; r < 25; ++r) hipLaunchKernelGGL(stream_graph, dim3(1024), dim3(256), 0, st, (const f4*)slot(3*r+7), (const f4*)slot(3*r+8), slot(3*r+9), 0.5f, nvec);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  printf("done\n");
  return 0;
}
