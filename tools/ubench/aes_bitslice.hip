// Table-free (bit-sliced) AES-128 on gfx950: what does a block-round cost when the S-box is Boolean algebra on the
// VALU instead of 16 LDS lookups?  (DESIGN.md 4.2 priced it on paper.)
//
//   wide   : 32 blocks per lane, one VGPR per (state byte, bit): 128 state registers          -> 1 or 2 waves / SIMD
//   packed : 16 blocks per lane, two state columns per VGPR (low / high half-word): 64 state registers, ShiftRows pays
//            32 half-word rotates per round                                                   -> up to 4 waves / SIMD
// Rounds are generated (aes_bitslice_gen.py -> aes_bitslice_rounds.inc): Boyar-Peralta S-box + MixColumns +
// AddRoundKey fused into v_bitop3_b32 / 2-input ops, verified in Python against a byte-wise AES and here against a
// table AES on random blocks.  Key planes: KEYED = round-key bit planes read from memory (wave-uniform, scalar loads);
// !KEYED = no AddRoundKey at all, the cheapest conceivable variant (a lower bound, not a cipher).
//
// Output per variant: ns per block, shader clocks per block-round per lane (x 64 = per wave-block-round, the unit of
// DESIGN.md's 143-150 clk for the T-table kernel), VGPRs / scratch from the compiler remarks (see the build line).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage aes_bitslice.hip -o aes_bitslice
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "aes_bitslice_rounds.inc"

#define HIP_OK(e)                                                                                   \
  do {                                                                                              \
    hipError_t r_ = (e);                                                                            \
    if (r_ != hipSuccess) {                                                                         \
      std::fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__);           \
      std::exit(2);                                                                                 \
    }                                                                                               \
  } while (0)

// kp: [11][NB][8] key planes (round 0 = whitening); io: [threads][NB][8] planes in / out
template <int NB, bool PACKED, bool KEYED, int BS>
__global__ __launch_bounds__(BS) void bs_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ kp, int iters,
                                                unsigned long long* __restrict__ cyc) {
  const size_t tid = (size_t)blockIdx.x * BS + threadIdx.x;
  uint32_t p[NB][8];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 8; ++i) p[b][i] = io[(tid * NB + b) * 8 + i];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KEYED) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[b][i] ^= kp[b * 8 + i];
    }
#pragma unroll 1
    for (int r = 1; r < 10; ++r) {
      if constexpr (KEYED) {
        uint32_t k[NB][8];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < 8; ++i) k[b][i] = kp[(r * NB + b) * 8 + i];
        if constexpr (PACKED) bsp_round(p, k);
        else bs_round(p, k);
      } else {
        if constexpr (PACKED) bsp_round_nokey(p);
        else bs_round_nokey(p);
      }
    }
    if constexpr (KEYED) {
      uint32_t k[NB][8];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) k[b][i] = kp[(10 * NB + b) * 8 + i];
      if constexpr (PACKED) bsp_round_last(p, k);
      else bs_round_last(p, k);
    } else {
      if constexpr (PACKED) bsp_round_nokey(p);
      else bs_round_nokey(p);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 8; ++i) io[(tid * NB + b) * 8 + i] = p[b][i];
  // the oldest wave of a SIMD wins issue arbitration and finishes long before its neighbours, so its own span is
  // not the kernel's: the two counters only give the CLOCK (shader ticks per 100 MHz tick); throughput is wall time
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

// ---- host reference (FIPS-197 AES-128, table S-box)
static uint8_t SB[256];
static void make_sbox() {
  uint8_t p = 1, q = 1;
  do {
    p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1b : 0));
    q ^= (uint8_t)(q << 1); q ^= (uint8_t)(q << 2); q ^= (uint8_t)(q << 4);
    if (q & 0x80) q ^= 0x09;
    auto rl = [](uint8_t v, int n) { return (uint8_t)((v << n) | (v >> (8 - n))); };
    SB[p] = (uint8_t)(q ^ rl(q, 1) ^ rl(q, 2) ^ rl(q, 3) ^ rl(q, 4) ^ 0x63);
  } while (p != 1);
  SB[0] = 0x63;
}
static uint8_t xt(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }
static void expand(const uint8_t key[16], uint8_t rk[176]) {
  memcpy(rk, key, 16);
  uint8_t rc = 1;
  for (int i = 16; i < 176; i += 4) {
    uint8_t t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
    if (i % 16 == 0) {
      uint8_t t0 = t[0];
      t[0] = (uint8_t)(SB[t[1]] ^ rc); t[1] = SB[t[2]]; t[2] = SB[t[3]]; t[3] = SB[t0];
      rc = xt(rc);
    }
    for (int j = 0; j < 4; ++j) rk[i + j] = (uint8_t)(rk[i - 16 + j] ^ t[j]);
  }
}
static void encrypt(const uint8_t rk[176], uint8_t s[16]) {
  for (int i = 0; i < 16; ++i) s[i] ^= rk[i];
  for (int r = 1; r <= 10; ++r) {
    uint8_t t[16];
    for (int c = 0; c < 4; ++c)
      for (int rr = 0; rr < 4; ++rr) t[4 * c + rr] = SB[s[4 * ((c + rr) % 4) + rr]];
    for (int c = 0; c < 4; ++c) {
      uint8_t* a = t + 4 * c;
      for (int rr = 0; rr < 4; ++rr) {
        uint8_t m = r == 10 ? a[rr]
                            : (uint8_t)(xt(a[rr]) ^ xt(a[(rr + 1) % 4]) ^ a[(rr + 1) % 4] ^ a[(rr + 2) % 4] ^ a[(rr + 3) % 4]);
        s[4 * c + rr] = (uint8_t)(m ^ rk[16 * r + 4 * c + rr]);
      }
    }
  }
}

// plane position of (block j, state byte b, bit i) in a lane's register file
template <bool PACKED>
static void locate(int j, int b, int& reg_byte, int& bit) {
  if (!PACKED) { reg_byte = b; bit = j; return; }
  const int col = b / 4, row = b % 4;
  reg_byte = 4 * (col % 2) + row;
  bit = j + (col >= 2 ? 16 : 0);
}

template <int NB, bool PACKED, bool KEYED, int BS>
static void run(const char* name, int waves_per_simd_goal) {
  constexpr int BLK = PACKED ? 16 : 32;
  int dev = 0;
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  // resident waves per SIMD come from the block size and the register count; grid = one block per CU x resident blocks
  int occ = 0;
  HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bs_kernel<NB, PACKED, KEYED, BS>, BS, 0));
  const int grid = ncu * (occ < 1 ? 1 : occ);
  const size_t threads = (size_t)grid * BS;
  std::vector<uint32_t> h(threads * NB * 8), kp(11 * NB * 8);
  // random blocks, fixed key
  const uint8_t key[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  uint8_t rk[176];
  expand(key, rk);
  for (int r = 0; r < 11; ++r)
    for (int b = 0; b < 16; ++b)
      for (int i = 0; i < 8; ++i) {
        int rb, bit0;
        locate<PACKED>(0, b, rb, bit0);
        const uint32_t m = PACKED ? (bit0 >= 16 ? 0xffff0000u : 0x0000ffffu) : 0xffffffffu;
        if ((rk[16 * r + b] >> i) & 1) kp[(r * NB + rb) * 8 + i] |= m;
      }
  std::vector<uint8_t> blocks(threads * BLK * 16);
  srand(7);
  for (auto& x : blocks) x = (uint8_t)rand();
  std::fill(h.begin(), h.end(), 0u);
  for (size_t t = 0; t < threads; ++t)
    for (int j = 0; j < BLK; ++j)
      for (int b = 0; b < 16; ++b)
        for (int i = 0; i < 8; ++i)
          if ((blocks[(t * BLK + j) * 16 + b] >> i) & 1) {
            int rb, bit;
            locate<PACKED>(j, b, rb, bit);
            h[(t * NB + rb) * 8 + i] |= 1u << bit;
          }
  uint32_t *d_io, *d_kp;
  unsigned long long* d_cyc;
  HIP_OK(hipMalloc(&d_io, h.size() * 4));
  HIP_OK(hipMalloc(&d_kp, kp.size() * 4));
  HIP_OK(hipMalloc(&d_cyc, 2 * grid * sizeof(unsigned long long)));
  HIP_OK(hipMemcpy(d_io, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_kp, kp.data(), kp.size() * 4, hipMemcpyHostToDevice));
  int bad = -1;
  if (KEYED) {  // one encryption, checked block by block on the first and last 64 lanes
    hipLaunchKernelGGL((bs_kernel<NB, PACKED, KEYED, BS>), dim3(grid), dim3(BS), 0, 0, d_io, d_kp, 1, d_cyc);
    std::vector<uint32_t> o(h.size());
    HIP_OK(hipMemcpy(o.data(), d_io, o.size() * 4, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t t = 0; t < threads; t = (t == 63 ? threads - 64 : t + 1)) {
      for (int j = 0; j < BLK; ++j) {
        uint8_t s[16];
        memcpy(s, &blocks[(t * BLK + j) * 16], 16);
        encrypt(rk, s);
        for (int b = 0; b < 16; ++b)
          for (int i = 0; i < 8; ++i) {
            int rb, bit;
            locate<PACKED>(j, b, rb, bit);
            if (((o[(t * NB + rb) * 8 + i] >> bit) & 1u) != (uint32_t)((s[b] >> i) & 1)) ++bad;
          }
      }
    }
    HIP_OK(hipMemcpy(d_io, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  const int iters = 200;
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w)
    hipLaunchKernelGGL((bs_kernel<NB, PACKED, KEYED, BS>), dim3(grid), dim3(BS), 0, 0, d_io, d_kp, iters, d_cyc);
  HIP_OK(hipEventRecord(e0));
  const int reps = 5;
  for (int w = 0; w < reps; ++w)
    hipLaunchKernelGGL((bs_kernel<NB, PACKED, KEYED, BS>), dim3(grid), dim3(BS), 0, 0, d_io, d_kp, iters, d_cyc);
  HIP_OK(hipEventRecord(e1));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(2 * grid);
  HIP_OK(hipMemcpy(cyc.data(), d_cyc, 2 * grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double csum = 0, rsum = 0;
  for (int g = 0; g < grid; ++g) { csum += (double)cyc[2 * g]; rsum += (double)cyc[2 * g + 1]; }
  const double clock_ghz = csum / rsum * 0.1;  // s_memrealtime ticks at 100 MHz
  const double blocks_total = (double)threads * BLK * iters * reps;
  const double ns_per_block = ms * 1e6 / blocks_total;
  // SIMD clocks per 64-block (= wave-wide) round, the unit of DESIGN.md's 143-150 clk for the T-table kernel: a SIMD
  // retires blocks_total * 10 / 64 of them per (4 * CUs) SIMDs in `ms`
  const double wps = (double)(occ < 1 ? 1 : occ) * BS / 64.0 / 4.0;
  const double clk_per_waveblock_round = (ms * 1e6 * clock_ghz) / (blocks_total * 10.0 / 64.0 / (4.0 * ncu));
  const double instr = (PACKED ? (KEYED ? 976.0 : 936.0) : (KEYED ? 1888.0 : 1808.0)) / BLK;  // wave-instructions per 64-block round
  std::printf("%-44s BS %4d  %.0f waves/SIMD  %s  %6.1f G block/s  %6.1f clk per 64-block round (T-table kernel: 143-150)  "
              "%.2f clk per VALU instruction  in-kernel clock %.2f GHz\n",
              name, BS, wps, bad < 0 ? "unchecked" : (bad == 0 ? "AES ok" : "WRONG"), 1.0 / ns_per_block,
              clk_per_waveblock_round, clk_per_waveblock_round / instr, clock_ghz);
  (void)waves_per_simd_goal;
  HIP_OK(hipFree(d_io)); HIP_OK(hipFree(d_kp)); HIP_OK(hipFree(d_cyc));
}

int main() {
  make_sbox();
  run<16, false, true, 256>("wide  32 blk/lane, keyed", 1);
  run<16, false, true, 512>("wide  32 blk/lane, keyed", 2);
  run<16, false, false, 256>("wide  32 blk/lane, no AddRoundKey (bound)", 1);
  run<16, false, false, 512>("wide  32 blk/lane, no AddRoundKey (bound)", 2);
  run<8, true, true, 512>("packed 16 blk/lane, keyed", 2);
  run<8, true, true, 1024>("packed 16 blk/lane, keyed", 4);
  run<8, true, false, 512>("packed 16 blk/lane, no AddRoundKey (bound)", 2);
  run<8, true, false, 1024>("packed 16 blk/lane, no AddRoundKey (bound)", 4);
  return 0;
}
