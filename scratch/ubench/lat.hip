// Instruction-latency microbenchmarks for gfx950 (single wave): cycles per dependent op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 256
__device__ __forceinline__ double lane_bcast(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
__global__ void k_lat(double* out, long long* cyc, double seed, int* idx) {
  __shared__ double lds[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += 64) lds[i] = (double)((i * 8 + 64) % 2048 * 8);  // pointer chase table (byte offsets)
  __shared__ int ilds[1024];
  for (int i = lane; i < 1024; i += 64) ilds[i] = ((i + 16) % 1024) * 4;
  __syncthreads();
  double x = seed + lane * 1e-3, y = seed * 0.5, acc = 0;
  long long t0, t1;
  int n = 0;
  // 0: dependent fma f64
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_fma(x, 0.999999, 1e-9);
  asm volatile("" : "+v"(x));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = t1 - t0;
  // 1: 4 independent fma chains (throughput)
  double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) {
    a0 = __builtin_fma(a0, 0.999999, 1e-9); a1 = __builtin_fma(a1, 0.999999, 1e-9);
    a2 = __builtin_fma(a2, 0.999999, 1e-9); a3 = __builtin_fma(a3, 0.999999, 1e-9);
  }
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = t1 - t0;
  acc += a0 + a1 + a2 + a3;
  // 2: readlane + fma chain
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_fma(y, lane_bcast(x, i & 7), x);
  asm volatile("" : "+v"(x));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = t1 - t0;
  // 3: shfl_xor f64 chain (ds_bpermute x2)
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) x += __shfl_xor(x, 1 << (i % 6));
  asm volatile("" : "+v"(x));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = (t1 - t0) * 4;
  // 4: LDS pointer chase b64 (read -> address)
  int p = lane * 8;
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) p = (int)*(double*)((char*)lds + p);
  asm volatile("" : "+v"(p));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = (t1 - t0) * 4;
  acc += p;
  // 5: LDS pointer chase b32
  int q = lane * 4;
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) q = *(int*)((char*)ilds + q);
  asm volatile("" : "+v"(q));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = (t1 - t0) * 4;
  acc += q;
  // 6: LDS write -> barrier -> read roundtrip chain
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) {
    lds[lane] = x;
    __syncthreads();
    x = lds[lane ^ 1] + 1.0;
    __syncthreads();
  }
  asm volatile("" : "+v"(x));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = (t1 - t0) * 4;
  // 7: rcp f64 + newton chain
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) { double r = __builtin_amdgcn_rcp(x); x = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r) + 2.0; }
  asm volatile("" : "+v"(x));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = (t1 - t0) * 4;
  // 8: DPP row_shr mov b32 x2 + add f64 chain
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, false);
    x += __hiloint2double(hi, lo);
  }
  asm volatile("" : "+v"(x));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = (t1 - t0) * 4;
  // 9: global load pointer chase (L2 / MALL hit after first pass)
  int g = lane;
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
  for (int i = 0; i < REP / 4; ++i) g = idx[g];
  asm volatile("" : "+v"(g));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = (t1 - t0) * 4;
  acc += g;
  // 10: dependent f64 add
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP; ++i) x = x + 1e-9;
  asm volatile("" : "+v"(x));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = t1 - t0;
  // 11: dependent f32 fma
  float f = (float)x;
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP; ++i) f = __builtin_fmaf(f, 0.99999f, 1e-6f);
  asm volatile("" : "+v"(f));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = t1 - t0;
  // 12: ds_read_b128 dependent chain
  int p2 = (lane & 3) * 16;
  asm volatile("" : "+v"(x), "+v"(y)); t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(x), "+v"(y));
#pragma unroll
  for (int i = 0; i < REP / 4; ++i) { double2 v = *(double2*)((char*)lds + p2); p2 = ((int)v.x) & 0x3ff0; }
  asm volatile("" : "+v"(p2));
  asm volatile("" : "+v"(x)); t1 = __builtin_readcyclecounter(); cyc[n++] = (t1 - t0) * 4;
  acc += p2 + f;
  out[lane] = x + acc;
}
int main() {
  double* out; long long* cyc; int* idx;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 32 * 8); hipMalloc(&idx, 4096 * 4);
  std::vector<int> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (i * 67 + 64) % 4096;
  hipMemcpy(idx, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const char* names[] = {"fma_f64 dep", "fma_f64 4 chains (per op)", "readlane+fma dep", "shfl_xor f64 dep", "lds b64 chase", "lds b32 chase",
                         "lds write-barrier-read-barrier", "rcp+newton+add", "dpp x2 + add f64", "global chase", "add_f64 dep", "fma_f32 dep", "lds b128 chase"};
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, cyc, 1.0, idx);
    hipDeviceSynchronize();
  }
  long long hc[32]; hipMemcpy(hc, cyc, 32 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < 13; ++i) printf("%-32s %8.1f counter ticks/op\n", names[i], (double)hc[i] / REP);
  return 0;
}
