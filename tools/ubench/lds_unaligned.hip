// Does gfx950 serve ds_read_b32 at byte offsets 1..3 (correct data?), and at what rate?
// Table: 256 entries x 32 lane copies; entry = 8 bytes (value doubled) so copy stride 8 B: lane l reads bytes [8*(32*i + (l&31)) + off, +4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 1024
template <int OFF, int BS>
__global__ __launch_bounds__(BS) void k(const uint32_t* __restrict__ tab, uint32_t* out, uint32_t* chk) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  for (int i = threadIdx.x; i < 256 * 32 * 2; i += BS) lds[i] = tab[i >> 6];  // entry (i>>6), doubled, 32 copies
  __syncthreads();
  const uint8_t* L = (const uint8_t*)lds;
  uint32_t lo = (threadIdx.x & 31) * 8 + OFF;
  uint32_t a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = (threadIdx.x * 2654435761u + j * 40503u);
  // correctness probe
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    uint32_t idx = (threadIdx.x * 37u) & 255u;
    uint32_t ad = idx * 256 + lo;
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ad));
    chk[threadIdx.x] = v;
  }
  for (int r = 0; r < REP; ++r) {
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      uint32_t ad = ((a[j] >> 8) & 0xff) * 256 + lo;
      asm volatile("ds_read_b32 %0, %1" : "=v"(v[j]) : "v"(ad));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = a[j] ^ v[(j + 1) & 15];
  }
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) x ^= a[j];
  out[blockIdx.x * BS + threadIdx.x] = x;
}
template <int OFF>
void run(const uint32_t* tab, const uint32_t* htab, uint32_t* out, uint32_t* chk) {
  constexpr int BS = 1024; int blocks = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OFF, BS>), dim3(blocks), dim3(BS), 65536, 0, tab, out, chk);
  (void)hipDeviceSynchronize();
  uint32_t h[64]; (void)hipMemcpy(h, chk, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) { uint32_t e = htab[(t * 37u) & 255u]; uint32_t want = OFF ? (e >> (8 * OFF) | e << (32 - 8 * OFF)) : e; bad += h[t] != want; }
  (void)hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<OFF, BS>), dim3(blocks), dim3(BS), 65536, 0, tab, out, chk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  double lookups = (double)blocks * BS * REP * 16;
  printf("offset %d: rotr-by-%d data %s (%d bad, got %08x for %08x); %.3f ms  %.1f lookups/clk/CU @2.4\n", OFF, 8 * OFF, bad ? "WRONG" : "correct", bad, h[1], htab[37],
         ms, lookups / (ms * 1e6) / 256 / 2.4);
}
int main() {
  uint32_t htab[256]; for (int i = 0; i < 256; ++i) htab[i] = 0x01020304u * (i + 1) ^ (i * 0x9e3779b9u);
  uint32_t *tab, *out, *chk;
  (void)hipMalloc(&tab, sizeof htab); (void)hipMemcpy(tab, htab, sizeof htab, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&chk, 256);
  run<0>(tab, htab, out, chk); run<1>(tab, htab, out, chk); run<2>(tab, htab, out, chk); run<3>(tab, htab, out, chk);
  return 0;
}
