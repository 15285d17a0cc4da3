// Microbenchmark (round 4): issue rate of v_mfma_f64_16x16x4_f64 on gfx950 -- alone, and with fp64 VALU FMAs interleaved.
// Prints cycles per MFMA per SIMD for 1, 2 waves per SIMD and 12 independent accumulator tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int VALU_PER_MFMA>
__global__ __launch_bounds__(64) void mfma_loop(int iters, const double* in, double* out, long long* cycles) {
  d4 acc[12];
  for (int t = 0; t < 12; ++t) acc[t] = d4{0, 0, 0, 0};
  double a = in[threadIdx.x], b = in[64 + threadIdx.x];
  double v[8];
  for (int k = 0; k < 8; ++k) v[k] = in[k] + threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < VALU_PER_MFMA; ++k) v[k % 8] = __builtin_fma(v[k % 8], a, b);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int t = 0; t < 12; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int V>
void run(int waves_per_simd, const double* in, double* out, long long* cyc) {
  const int iters = 2000, blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  mfma_loop<V><<<blocks, 64>>>(10, in, out, cyc);
  hipEventRecord(e0);
  mfma_loop<V><<<blocks, 64>>>(iters, in, out, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto c : h) mean += c;
  mean /= blocks;
  const double mfmas = (double)iters * 12;
  printf("VALU/MFMA %2d  waves/SIMD %d: %.3f ms, %.1f counter ticks per MFMA per wave, %.2f TFLOP/s (MFMA only)\n", V, waves_per_simd, ms,
         mean / mfmas, blocks * mfmas * 2048.0 / (ms * 1e-3) / 1e12);
}

int main() {
  double *in, *out;
  long long* cyc;
  hipMalloc(&in, 1024 * 8);
  hipMalloc(&out, 256 * 4 * 4 * 64 * 8);
  hipMalloc(&cyc, 256 * 4 * 4 * 8);
  std::vector<double> h(1024, 1e-3);
  hipMemcpy(in, h.data(), 1024 * 8, hipMemcpyHostToDevice);
  for (int w = 1; w <= 4; w *= 2) {
    run<0>(w, in, out, cyc);
    run<2>(w, in, out, cyc);
    run<5>(w, in, out, cyc);
    run<8>(w, in, out, cyc);
    run<16>(w, in, out, cyc);
  }
  return 0;
}
