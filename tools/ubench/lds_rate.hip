// Micro-benchmark: lookup rates on gfx950 -- conflict-free ds_read_b32 / b64 from a lane-replicated
// table, and per-lane global_load_dword gathers from a 1 KiB (L1-resident) table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 1024

template <int MODE, int BS>  // 0: ds_read_b32 x16, 1: ds_read_b64 x16, 2: global gather x16, 3: 14 lds + 2 global
__global__ __launch_bounds__(BS) void k(const uint32_t* __restrict__ tab, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  for (int i = threadIdx.x; i < 256 * 64; i += BS) lds[i] = tab[i >> 6] ^ (i & 63);
  __syncthreads();
  const uint8_t* L = (const uint8_t*)lds;
  uint32_t lo = (threadIdx.x & 63) * 4, lo8 = (threadIdx.x & 31) * 8;
  uint32_t mask = 0xff00; asm volatile("" : "+v"(mask));
  uint32_t a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = (threadIdx.x * 2654435761u + j * 40503u);
  for (int r = 0; r < REP; ++r) {
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      uint32_t ad = __builtin_amdgcn_bitop3_b32(a[j], mask, MODE == 1 ? lo8 : lo, 0xEA);  // (a & mask) | lo
      if (MODE == 0) v[j] = *(const uint32_t*)(L + ad);
      else if (MODE == 1) { uint2 t = *(const uint2*)(L + ad); v[j] = t.x ^ t.y; }
      else if (MODE == 2) v[j] = tab[(a[j] >> 8) & 0xff];
      else v[j] = (j < 14) ? *(const uint32_t*)(L + ad) : tab[(a[j] >> 8) & 0xff];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = a[j] ^ v[(j + 1) & 15];
  }
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) x ^= a[j];
  out[blockIdx.x * BS + threadIdx.x] = x;
}

template <int MODE, int BS>
void run(const char* name, const uint32_t* tab, uint32_t* out, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, BS>), dim3(blocks), dim3(BS), 65536, 0, tab, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, BS>), dim3(blocks), dim3(BS), 65536, 0, tab, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  double lookups = (double)blocks * BS * REP * 16;
  printf("%-34s %4d x %4d: %.3f ms  %.2f lookups/ns/CU = %.1f lookups/clk/CU @2.4 (%.1f @2.1)\n", name, blocks, BS, ms,
         lookups / (ms * 1e6) / 256, lookups / (ms * 1e6) / 256 / 2.4, lookups / (ms * 1e6) / 256 / 2.1);
}

int main() {
  uint32_t *tab, *out; (void)hipMalloc(&tab, 1024); (void)hipMalloc(&out, 1 << 24);
  uint32_t h[256]; for (int i = 0; i < 256; ++i) h[i] = i * 2654435761u;
  (void)hipMemcpy(tab, h, 1024, hipMemcpyHostToDevice);
  run<0, 1024>("ds_read_b32 32 waves/CU", tab, out, 512);
  run<0, 1024>("ds_read_b32 16 waves/CU", tab, out, 256);
  run<0, 512>("ds_read_b32  8 waves/CU", tab, out, 256);
  run<1, 1024>("ds_read_b64 32 waves/CU", tab, out, 512);
  run<1, 1024>("ds_read_b64 16 waves/CU", tab, out, 256);
  run<2, 1024>("global gather 1KiB 32 waves/CU", tab, out, 512);
  run<2, 1024>("global gather 1KiB 16 waves/CU", tab, out, 256);
  run<3, 1024>("14 lds + 2 global 32 waves/CU", tab, out, 512);
  run<3, 1024>("14 lds + 2 global 16 waves/CU", tab, out, 256);
  return 0;
}
