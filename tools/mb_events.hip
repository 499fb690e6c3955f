// What do HIP events bound to a dispatch measure?  (round 2)
// RESULT on MI355X / ROCm 7.2 (profiles/r02_mb_events.txt): (a) reads the true duration + 0.69-0.78 us (the start event is a
// marker packet in front of the dispatch); (b) is identically 0, i.e. elapsed() of two dispatch-bound stop events is
// end(b) - end(a): HIP events cannot deliver a dispatch's own start timestamp, so bench.py takes the profiler's
// (rocprofv3 child) for the timed mode and reports the event figures beside it.
//   hipExtLaunchKernelGGL(kernel, ..., startEvent, stopEvent, flags)
// (a) start + stop given: elapsed(start, stop)
// (b) only stop events given, three consecutive launches i, i+1, i+2:
//       m(a,b) = elapsed(stop_a, stop_b);   d = m(i,i+1) + m(i+1,i+2) - m(i,i+2)
//     If elapsed() of two dispatch-bound events is  end(b) - START(a)  this d is exactly the duration of launch i+1
//     (no marker packet anywhere); if it is end(b) - end(a) it is identically 0.
// The kernels spin for a known time on the 100 MHz constant clock (s_memrealtime), so the truth is known.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void spin(long long ticks, int *sink) {
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (sink && threadIdx.x == 1000000) *sink = 1;
}

int main() {
  hipStream_t st;
  hipStreamCreate(&st);
  const int N = 300;
  const double want_us[3] = {10.0, 20.0, 5.0};
  std::vector<hipEvent_t> s(N), e(N), o(N);
  for (int i = 0; i < N; ++i) { hipEventCreate(&s[i]); hipEventCreate(&e[i]); hipEventCreate(&o[i]); }
  // warm-up
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, 1000LL, (int *)nullptr);
  hipStreamSynchronize(st);
  // (a) start+stop pairs
  for (int i = 0; i < N; ++i)
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, s[i], e[i], 0, (long long)(want_us[i % 3] * 100), (int *)nullptr);
  hipStreamSynchronize(st);
  double acc[3] = {0, 0, 0}; int cnt[3] = {0, 0, 0};
  for (int i = 0; i < N; ++i) { float ms; hipEventElapsedTime(&ms, s[i], e[i]); acc[i % 3] += ms * 1e3; cnt[i % 3]++; }
  for (int k = 0; k < 3; ++k) printf("(a) start+stop events : spin %5.1f us -> measured %.2f us\n", want_us[k], acc[k] / cnt[k]);
  // (b) stop events only
  for (int i = 0; i < N; ++i)
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, nullptr, o[i], 0, (long long)(want_us[i % 3] * 100), (int *)nullptr);
  hipStreamSynchronize(st);
  double acc2[3] = {0, 0, 0}, raw[3] = {0, 0, 0}; int cnt2[3] = {0, 0, 0};
  for (int i = 0; i + 2 < N; ++i) {
    float m01, m12, m02;
    hipError_t r1 = hipEventElapsedTime(&m01, o[i], o[i + 1]);
    hipError_t r2 = hipEventElapsedTime(&m12, o[i + 1], o[i + 2]);
    hipError_t r3 = hipEventElapsedTime(&m02, o[i], o[i + 2]);
    if (r1 != hipSuccess || r2 != hipSuccess || r3 != hipSuccess) { printf("elapsed failed: %s\n", hipGetErrorString(r1)); return 1; }
    const int k = (i + 1) % 3;
    acc2[k] += (m01 + m12 - m02) * 1e3; raw[k] += m01 * 1e3; cnt2[k]++;
  }
  for (int k = 0; k < 3; ++k)
    printf("(b) stop events only  : spin %5.1f us -> m(i,i+1)+m(i+1,i+2)-m(i,i+2) = %.2f us   (m(i,i+1) alone %.2f us)\n",
           want_us[k], acc2[k] / cnt2[k], raw[k] / cnt2[k]);
  return 0;
}
