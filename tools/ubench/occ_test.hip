// How many 1024-lane workgroups with 128 KiB / 64 KiB / 0 KiB of LDS run concurrently on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int LDSW>
__global__ __launch_bounds__(1024) void k(unsigned long long* t, unsigned* xcc, int spin) {
  __shared__ unsigned lds[LDSW > 0 ? LDSW : 1];
  if (LDSW > 0) lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u + (LDSW > 0 ? lds[(v >> 8) & 1023] : 0);
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    t[2 * blockIdx.x] = r0; t[2 * blockIdx.x + 1] = r1 + (v == 12345 ? 1 : 0);
    unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    xcc[2 * blockIdx.x] = id; xcc[2 * blockIdx.x + 1] = hw;
  }
}
template <int LDSW>
void run(const char* name, int blocks) {
  unsigned long long* t; unsigned* x;
  (void)hipMalloc(&t, blocks * 16); (void)hipMalloc(&x, blocks * 8);
  hipLaunchKernelGGL(k<LDSW>, dim3(blocks), dim3(1024), 0, 0, t, x, 20000);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k<LDSW>, dim3(blocks), dim3(1024), 0, 0, t, x, 20000);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * blocks); std::vector<unsigned> hx(2 * blocks);
  (void)hipMemcpy(h.data(), t, blocks * 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hx.data(), x, blocks * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0; double dur = 0;
  for (int i = 0; i < blocks; ++i) { t0 = std::min(t0, h[2 * i]); t1 = std::max(t1, h[2 * i + 1]); dur += h[2 * i + 1] - h[2 * i]; }
  // concurrency = sum of durations / span
  int late = 0; for (int i = 0; i < blocks; ++i) if (h[2 * i] - t0 > (t1 - t0) / 4) ++late;
  int xc[8] = {0}; for (int i = 0; i < blocks; ++i) xc[hx[2 * i] & 7]++;
  printf("%-12s blocks %4d: span %.1f us, mean block %.1f us, avg concurrency %.1f, blocks starting late %d, per-XCC %d %d %d %d %d %d %d %d\n", name, blocks,
         (t1 - t0) * 0.01, dur / blocks * 0.01, dur / (double)(t1 - t0), late, xc[0], xc[1], xc[2], xc[3], xc[4], xc[5], xc[6], xc[7]);
  (void)hipFree(t); (void)hipFree(x);
}
int main() {
  run<32768>("lds 128KiB", 256); run<32768>("lds 128KiB", 128); run<32768>("lds 128KiB", 512);
  run<16384>("lds 64KiB", 256); run<16384>("lds 64KiB", 512);
  run<20480>("lds 80KiB", 512);
  run<0>("lds 0", 256); run<0>("lds 0", 512);
  return 0;
}
