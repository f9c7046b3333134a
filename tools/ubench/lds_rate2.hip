// LDS ds_read_b32 rate vs access pattern and batch size (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 2048
// PAT 0: random conflict-free (idx<<8 | lane*4)  1: linear (lane*4 + j*256)  2: same random row for all lanes (idx uniform)  3: random row, 32 replicas (idx<<7 | (lane&31)*4)
template <int PAT, int NB, int BS>
__global__ __launch_bounds__(BS) void k(const uint32_t* __restrict__ tab, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  for (int i = threadIdx.x; i < 256 * 64; i += BS) lds[i] = tab[(i >> 6) & 255] ^ (i & 63);
  __syncthreads();
  const uint8_t* L = (const uint8_t*)lds;
  uint32_t lo = (threadIdx.x & 63) * 4, lo32 = (threadIdx.x & 31) * 4;
  uint32_t mask = PAT == 3 ? 0x7f80 : 0xff00; asm volatile("" : "+v"(mask));
  uint32_t a[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) a[j] = (PAT == 2 ? blockIdx.x : threadIdx.x) * 2654435761u + j * 40503u;
  for (int r = 0; r < REP; ++r) {
    uint32_t v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      uint32_t ad;
      if (PAT == 1) ad = lo + ((j * 256 + (r & 3) * 4096) & 0xffff);
      else ad = __builtin_amdgcn_bitop3_b32(a[j], mask, PAT == 3 ? lo32 : lo, 0xEA);
      v[j] = *(const uint32_t*)(L + ad);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) a[j] = a[j] ^ v[(j + 1) % NB];
  }
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) x ^= a[j];
  out[blockIdx.x * BS + threadIdx.x] = x;
}
template <int PAT, int NB, int BS>
void run(const char* name, const uint32_t* tab, uint32_t* out, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PAT, NB, BS>), dim3(blocks), dim3(BS), 65536, 0, tab, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, NB, BS>), dim3(blocks), dim3(BS), 65536, 0, tab, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  double lookups = (double)blocks * BS * REP * NB;
  printf("%-44s: %.3f ms  %.2f lookups/ns/CU\n", name, ms, lookups / (ms * 1e6) / 256);
}
int main() {
  uint32_t *tab, *out; (void)hipMalloc(&tab, 1024); (void)hipMalloc(&out, 1 << 24);
  uint32_t h[256]; for (int i = 0; i < 256; ++i) h[i] = i * 2654435761u;
  (void)hipMemcpy(tab, h, 1024, hipMemcpyHostToDevice);
  run<0, 16, 1024>("random cf, batch 16, 32 waves", tab, out, 512);
  run<0, 8, 1024>("random cf, batch 8, 32 waves", tab, out, 512);
  run<0, 32, 1024>("random cf, batch 32, 32 waves", tab, out, 512);
  run<1, 16, 1024>("linear, batch 16, 32 waves", tab, out, 512);
  run<1, 32, 1024>("linear, batch 32, 32 waves", tab, out, 512);
  run<2, 16, 1024>("wave-uniform row, batch 16, 32 waves", tab, out, 512);
  run<3, 16, 1024>("random cf 32-replica, batch 16, 32 waves", tab, out, 512);
  run<0, 16, 1024>("random cf, batch 16, 16 waves", tab, out, 256);
  run<1, 16, 1024>("linear, batch 16, 16 waves", tab, out, 256);
  run<0, 32, 1024>("random cf, batch 32, 16 waves", tab, out, 256);
  return 0;
}
