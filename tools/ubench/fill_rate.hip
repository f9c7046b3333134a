// Micro-benchmark: cost of the per-workgroup LDS table fill (kernel = prologue only), current loop vs
// "16 loads up front" variant; 256 workgroups x 1024 lanes, back-to-back launches.
#include "../../fss_amd/csrc/fss_dev.hpp"
#include <cstdio>
using namespace fssamd;

__device__ __forceinline__ void fill_v2(uint32_t* lds, const uint32_t* __restrict__ te0) {
  const int tid = threadIdx.x;
  uint32_t t[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) t[k] = te0[((tid >> 6) + 16 * k) & 255];
  const int odd = (tid >> 5) & 1;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int tbl = ((k >> 4) << 1) | odd;
    const uint32_t v = t[k & 15];
    lds[tid + 1024 * k] = __builtin_amdgcn_alignbit(v, v, (32 - 8 * tbl) & 31);
  }
}

template <int V>
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ te0, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_te[kLdsWords];
  if (V == 0) lds_table_fill(lds_te, te0);
  if (V == 1) fill_v2(lds_te, te0);
  __syncthreads();
  if (V == 2) { if (threadIdx.x == 0) out[blockIdx.x] = 1; return; }
  out[blockIdx.x * 1024 + threadIdx.x] = lds_te[(threadIdx.x * 33 + blockIdx.x) & 32767];
}

template <int V>
void run(const char* name, const uint32_t* te0, uint32_t* out) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<V>, dim3(256), dim3(1024), 0, 0, te0, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<V>, dim3(256), dim3(1024), 0, 0, te0, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.2f us per launch (200 back-to-back)\n", name, ms * 1000 / 200);
}

int main() {
  uint32_t h[256]; for (int i = 0; i < 256; ++i) h[i] = i * 2654435761u;
  uint32_t *te0, *out;
  (void)hipMalloc(&te0, 1024); (void)hipMemcpy(te0, h, 1024, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 256 * 1024 * 4);
  // check v2 == v0
  uint32_t *a, *b; (void)hipMalloc(&a, 256 * 1024 * 4); (void)hipMalloc(&b, 256 * 1024 * 4);
  hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, te0, a);
  hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, te0, b);
  static uint32_t ha[256 * 1024], hb[256 * 1024];
  (void)hipMemcpy(ha, a, sizeof(ha), hipMemcpyDeviceToHost); (void)hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256 * 1024; ++i) bad += ha[i] != hb[i];
  printf("v2 vs current: %d mismatches\n", bad);
  run<2>("empty kernel (128 KiB LDS)", te0, out);
  run<0>("current fill", te0, out);
  run<1>("16 loads up front", te0, out);
  return 0;
}
