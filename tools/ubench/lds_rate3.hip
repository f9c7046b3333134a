// LDS ds_read_b32 rate vs table layout (gfx950), with the in-kernel shader clock
// (s_memtime / s_memrealtime) so rates are per real clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 2048
// LAY: row stride (bytes), lane offset = (lane & LMASK) * LMUL + (lane >> 5) * HI
template <int ROWSHIFT, int LMASK, int LMUL, int HI, int NB, int BS, int B64>
__global__ __launch_bounds__(BS) void k(const uint32_t* __restrict__ tab, uint32_t* out, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  for (int i = threadIdx.x; i < 40960; i += BS) lds[i] = tab[i & 255] * 2654435761u + i;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const uint8_t* L = (const uint8_t*)lds;
  uint32_t lane = threadIdx.x & 63;
  uint32_t lo = (lane & LMASK) * LMUL + (lane >> 5) * HI;
  uint32_t mask = 0xffu << ROWSHIFT; asm volatile("" : "+v"(mask));
  uint32_t a[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) a[j] = threadIdx.x * 2654435761u + j * 40503u;
  for (int r = 0; r < REP; ++r) {
    uint32_t v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      uint32_t ad = __builtin_amdgcn_bitop3_b32(a[j], mask, lo, 0xEA);
      if (B64) { uint2 t = *(const uint2*)(L + ad); v[j] = t.x ^ t.y; }
      else v[j] = *(const uint32_t*)(L + ad);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) a[j] = a[j] ^ v[(j + 1) % NB];
  }
  // time the WHOLE workgroup: the oldest wave of a SIMD wins issue arbitration and finishes far
  // earlier than its younger neighbours, so one wave's own duration overstates the rate
  __syncthreads();
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) x ^= a[j];
  out[blockIdx.x * BS + threadIdx.x] = x;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int ROWSHIFT, int LMASK, int LMUL, int HI, int NB, int BS, int B64>
void run(const char* name, const uint32_t* tab, uint32_t* out, unsigned long long* clk, int blocks) {
  hipLaunchKernelGGL((k<ROWSHIFT, LMASK, LMUL, HI, NB, BS, B64>), dim3(blocks), dim3(BS), 163840, 0, tab, out, clk);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((k<ROWSHIFT, LMASK, LMUL, HI, NB, BS, B64>), dim3(blocks), dim3(BS), 163840, 0, tab, out, clk);
  (void)hipDeviceSynchronize();
  static unsigned long long h[4096];
  (void)hipMemcpy(h, clk, blocks * 16, hipMemcpyDeviceToHost);
  double st = 0, rt = 0; for (int i = 0; i < blocks; ++i) { st += h[2 * i]; rt += h[2 * i + 1]; }
  double cyc = st / blocks, ghz = st / rt * 0.1;
  double waves_per_cu = (double)blocks * BS / 64 / 256;
  double lookups_per_cu = waves_per_cu * 64 * REP * NB;
  printf("%-52s: %.2f GHz  %.2f lookups/clk/CU  (%.1f /ns/CU)\n", name, ghz, lookups_per_cu / cyc, lookups_per_cu / cyc * ghz);
}
int main() {
  uint32_t *tab, *out; unsigned long long* clk;
  (void)hipMalloc(&tab, 1024); (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&clk, 65536);
  uint32_t h[256]; for (int i = 0; i < 256; ++i) h[i] = i * 2654435761u + 12345;
  (void)hipMemcpy(tab, h, 1024, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)k<8, 63, 4, 0, 16, 1024, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
#define RUN(name, ...) { (void)hipFuncSetAttribute((const void*)k<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); run<__VA_ARGS__>(name, tab, out, clk, 256); }
  RUN("A 256B rows, lane*4 (64 copies), 16 waves", 8, 63, 4, 0, 16, 1024, 0);
  RUN("B 128B rows, (lane&31)*4 (32 copies), 16 waves", 7, 31, 4, 0, 16, 1024, 0);
  RUN("C 256B rows, (lane&31)*4, 16 waves", 8, 31, 4, 0, 16, 1024, 0);
  RUN("D 256B rows, (lane&31)*8 b32, 16 waves", 8, 31, 8, 0, 16, 1024, 0);
  RUN("E 128B rows, 2 copies sets +32KiB, 16 waves", 7, 31, 4, 32768, 16, 1024, 0);
  RUN("F 256B rows, (lane&31)*4 + hi*128 (=A), 16 waves", 8, 31, 4, 128, 16, 1024, 0);
  RUN("G b64 256B rows (lane&31)*8, 16 waves", 8, 31, 8, 0, 16, 1024, 1);
  RUN("H b64 512B rows lane*8 (64 copies), 16 waves", 9, 63, 8, 0, 16, 1024, 1);
  RUN("B batch 32", 7, 31, 4, 0, 32, 1024, 0);
  RUN("A batch 32", 8, 63, 4, 0, 32, 1024, 0);
  return 0;
}
