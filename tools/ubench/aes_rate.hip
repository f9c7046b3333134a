// Micro-benchmark: cycles per wave-AES of the engine's own AES primitives in a tight loop
// (no global traffic): aes_mmo<SEL> (one stream) and aes_mmo2<SEL> (two staggered streams).
#include "../../fss_amd/csrc/fss_dev.hpp"
#include <cstdio>
#include <vector>
using namespace fssamd;
#define REP 256

template <int MODE>  // 0: aes_mmo<false>  1: aes_mmo<true>  2: aes_mmo2<false> 3: aes_mmo2<true>
__global__ __launch_bounds__(1024) void k(const DevCtx K, const uint32_t* __restrict__ te0, uint4* out, unsigned long long* clk) {
  FSS_LDS_PROLOGUE(te0);
  const DevCtx* c = &K;
  uint32_t a[4] = {threadIdx.x * 2654435761u, blockIdx.x * 40503u + 1, 3u, 5u}, b[4] = {a[1], a[0], 7u, 11u};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int i = 0; i < REP; ++i) {
    uint32_t oa[4], ob[4];
    uint32_t m = (uint32_t)__builtin_amdgcn_sbfe(a[0], 3, 1);
    if (MODE == 0) { aes_mmo<false>(L, c->rk[0], 0, a, oa); aes_mmo<false>(L, c->rk[1], 0, b, ob); }
    if (MODE == 1) { aes_mmo<true>(L, &c->sel01[0][0], m, a, oa); aes_mmo<true>(L, &c->sel01[0][0], ~m, b, ob); }
    if (MODE == 2) aes_mmo2<false>(L, c->rk[0], c->rk[1], 0, 0, a, b, oa, ob);
    if (MODE == 3) aes_mmo2<true>(L, &c->sel01[0][0], &c->sel01[0][0], m, ~m, a, b, oa, ob);
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = oa[j]; b[j] = ob[j]; }
  }
  __syncthreads();  // whole workgroup: the oldest wave of a SIMD finishes well before the younger ones
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 1024 + threadIdx.x] = make_uint4(a[0] ^ b[0], a[1] ^ b[1], a[2] ^ b[2], a[3] ^ b[3]);
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, const DevCtx& K, const uint32_t* te0, uint4* out, unsigned long long* clk) {
  const int blocks = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, K, te0, out, clk); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, K, te0, out, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  (void)hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  double st = 0, rt = 0; for (int i = 0; i < blocks; ++i) { st += h[2 * i]; rt += h[2 * i + 1]; }
  double cyc = st / blocks, ghz = st / rt * 0.1;
  double wave_aes_per_cu = 16.0 * REP * 2;          // 16 waves, 2 AES per iteration
  double simd_cyc_per_wave_aes = cyc * 4 / wave_aes_per_cu;
  double total_aes = (double)blocks * 1024 * REP * 2;
  printf("%-22s %.2f GHz  %.0f SIMD-cycles per wave-AES -> %.1f G AES/s in-kernel; wall %.3f ms -> %.1f G AES/s (incl. launch + LDS fill)\n",
         name, ghz, simd_cyc_per_wave_aes, 256.0 * ghz / (simd_cyc_per_wave_aes / 256.0), ms, total_aes / (ms * 1e6));
}

int main() {
  DevCtx K{};  // arbitrary key words: timing only
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 44; ++j) K.rk[i][j] = 0x9e3779b9u * (i * 44 + j + 1);
  for (int r = 0; r < 11; ++r) for (int j = 0; j < 8; ++j) { K.sel01[r][j] = 0x85ebca6bu * (r * 8 + j + 1); K.sel02[r][j] = K.sel01[r][j] ^ 5; K.sel13[r][j] = K.sel01[r][j] ^ 9; }
  uint32_t h[256]; for (int i = 0; i < 256; ++i) h[i] = i * 2654435761u;
  uint32_t* te0; uint4* out; unsigned long long* clk;
  (void)hipMalloc(&te0, 1024 + sizeof(DevCtx)); (void)hipMemcpy(te0, h, 1024, hipMemcpyHostToDevice);
  K.self_dev = te0 + 256;
  (void)hipMalloc(&out, 256 * 1024 * 16); (void)hipMalloc(&clk, 256 * 16);
  run<0>("aes_mmo<false> x2", K, te0, out, clk);
  run<1>("aes_mmo<true> x2", K, te0, out, clk);
  run<2>("aes_mmo2<false>", K, te0, out, clk);
  run<3>("aes_mmo2<true>", K, te0, out, clk);
  return 0;
}
