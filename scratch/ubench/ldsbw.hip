// LDS throughput vs. number of active lanes (gfx950): does a ds_read_b128 with few active lanes cost fewer LDS cycles?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VEC>
__global__ void k_bw(double* out, int nact, int reps, int uniform) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) lds[i] = i;
  __syncthreads();
  double acc = 0;
  if (lane < nact) {
    const unsigned base = (unsigned)(size_t)(uniform ? 0 : lane * 16);
    for (int it = 0; it < reps; ++it) {
      if (VEC == 2) {
        double2 a, b, c, d, e, f, g, h;
        asm volatile(
            "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
            "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
            "s_waitcnt lgkmcnt(0)"
            : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h)
            : "v"(base));
        acc += a.x;
      } else {
        double a, b, c, d, e, f, g, h;
        asm volatile(
            "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:1024\n ds_read_b64 %2, %8 offset:2048\n ds_read_b64 %3, %8 offset:3072\n"
            "ds_read_b64 %4, %8 offset:4096\n ds_read_b64 %5, %8 offset:5120\n ds_read_b64 %6, %8 offset:6144\n ds_read_b64 %7, %8 offset:7168\n"
            "s_waitcnt lgkmcnt(0)"
            : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h)
            : "v"(base));
        acc += a;
      }
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}
int main() {
  double* out;
  const int blocks = 256 * 8;
  (void)hipMalloc(&out, blocks * 64 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int reps = 2000;
  for (int vec = 1; vec <= 2; ++vec)
    for (int uni = 0; uni < 2; ++uni)
      for (int nact : {64, 32, 16, 8}) {
        float best = 1e9;
        for (int t = 0; t < 3; ++t) {
          (void)hipEventRecord(e0);
          if (vec == 2) hipLaunchKernelGGL(k_bw<2>, dim3(blocks), dim3(64), 20000, 0, out, nact, reps, uni);
          else hipLaunchKernelGGL(k_bw<1>, dim3(blocks), dim3(64), 20000, 0, out, nact, reps, uni);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double instr_per_cu = 8.0 * reps * 8;
        printf("b%-3d %s active %2d : %.3f ms  -> %.1f cycles per wave-instruction per CU @2.3GHz\n", vec * 64, uni ? "uniform " : "per-lane", nact, best,
               best * 1e6 / instr_per_cu * 2.3);
      }
  return 0;
}
