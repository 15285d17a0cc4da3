// What does v_mov_b64_dpp row_newbcast:K write?  (one wave, dst pre-filled with a sentinel)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K>
__global__ void k(double* out) {
  const int lane = threadIdx.x;
  double src = 100.0 + lane, dst = -1.0;
  asm volatile("s_nop 4\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 4" : "+v"(dst) : "v"(src), "n"(K));
  out[lane] = dst;
  double b = __builtin_amdgcn_update_dpp(-2.0, src, 0x150 + K, 0xf, 0xf, false);
  out[64 + lane] = b;
}
int main() {
  double* d; hipMalloc(&d, 128 * 8);
  double h[128];
  k<3><<<1, 64>>>(d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("row_newbcast:3 asm    :"); for (int i = 0; i < 64; ++i) printf(" %g", h[i]); printf("\n");
  printf("row_newbcast:3 builtin:"); for (int i = 0; i < 64; ++i) printf(" %g", h[64 + i]); printf("\n");
  k<11><<<1, 64>>>(d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("row_newbcast:11 asm   :"); for (int i = 0; i < 64; ++i) printf(" %g", h[i]); printf("\n");
  return 0;
}
