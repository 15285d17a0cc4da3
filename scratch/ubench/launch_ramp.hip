// Does a small scratch allocation slow the wave launch ramp?  4096 single-wave workgroups, 20 KB of LDS each, ~0.3 ms of
// work per wave; start time of every wave (100 MHz wall clock) with and without 64 B of scratch per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ long long wall() { return __builtin_readcyclecounter() * 0 + (long long)wall_clock64(); }
template <bool SCRATCH>
__global__ __launch_bounds__(64, 2) void k(long long* t0, long long* t1, double* sink, int spin) {
  extern __shared__ double lds[];
  const int w = blockIdx.x, lane = threadIdx.x;
  const long long a = wall_clock64();
  double acc = lane;
  volatile double priv[8];   // SCRATCH: dynamically indexed private array -> scratch memory
  if (SCRATCH) for (int i = 0; i < 8; ++i) priv[i] = i + lane;
  lds[lane] = acc;
  for (int i = 0; i < spin; ++i) {
    acc = __builtin_fma(acc, 1.0000001, lds[(lane + i) & 63]);
    if (SCRATCH) acc += priv[(i + lane) & 7];
  }
  if (lane == 0) { t0[w] = a; t1[w] = wall_clock64(); }
  sink[w * 64 + lane] = acc;
}
template <bool S> void run(const char* name) {
  const int W = 4096; long long *t0, *t1; double* sink;
  hipMalloc(&t0, W * 8); hipMalloc(&t1, W * 8); hipMalloc(&sink, W * 64 * 8);
  hipFuncSetAttribute((const void*)k<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 20176);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k<S>, dim3(W), dim3(64), 20176, 0, t0, t1, sink, 20000);
    hipDeviceSynchronize();
  }
  std::vector<long long> a(W), b(W);
  hipMemcpy(a.data(), t0, W * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), t1, W * 8, hipMemcpyDeviceToHost);
  const long long base = *std::min_element(a.begin(), a.end());
  std::vector<double> st(W); for (int i = 0; i < W; ++i) st[i] = (a[i] - base) / 100.0;  // us
  std::sort(st.begin(), st.end());
  printf("%s: wave start times (us): 25%% %.1f  50%% %.1f (wave 2048 = %.1f)  kernel span %.1f us\n", name, st[W / 4], st[W / 2 - 1], st[2047],
         (*std::max_element(b.begin(), b.end()) - base) / 100.0);
}
int main() { run<false>("no scratch"); run<true>("64 B scratch"); run<false>("no scratch"); return 0; }
