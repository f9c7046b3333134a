// Micro-benchmark: per-CU issue rate of the integer VALU ops the AES round uses (gfx950).
// Each kernel runs REP x 16 independent ops per lane; reports wave-instructions per clock per CU
// assuming the clock given by --mhz (effective clock printed from wall time is not known here,
// so we print ns and ops/ns; compare ratios).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define REP 8192

#define KERNEL(NAME, BODY)                                                            \
  __global__ __launch_bounds__(1024) void NAME(uint32_t* out, uint32_t a, uint32_t b) { \
    uint32_t v0 = threadIdx.x + a, v1 = v0 * 3 + b, v2 = v0 ^ 0x55, v3 = v1 + 7, v4 = v0 + 11, v5 = v1 ^ 13,  \
             v6 = v2 + 17, v7 = v3 ^ 19, w = threadIdx.x * 4, k = b;                   \
    for (int i = 0; i < REP; ++i) {                                                    \
      BODY BODY                                                                        \
    }                                                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7; \
  }

#define OP8(S) asm volatile(S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(w), "s"(k));

#define S_XOR(i) "v_xor_b32 %" #i ", %8, %" #i "\n"
#define S_BITOP(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %" #i " bitop3:0x96\n"
#define S_BITOPS(i) "v_bitop3_b32 %" #i ", %" #i ", %9, %8 bitop3:0x96\n"
#define S_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define S_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 24\n"
#define S_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %9, %8\n"
#define S_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 8, %8\n"
#define S_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define S_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %" #i "\n"
#define S_ADD(i) "v_add_u32 %" #i ", %8, %" #i "\n"
#define S_LSHR(i) "v_lshrrev_b32 %" #i ", 8, %" #i "\n"
#define S_SDWA(i) "v_mov_b32_sdwa %" #i ", %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n"
#define S_ORSDWA(i) "v_or_b32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
#define S_XOR3(i) "v_xor3_b32 %" #i ", %" #i ", %8, %" #i "\n"
#define S_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define S_PERMV(i) "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define S_ALIGNV(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", %8\n"
#define S_ANDORV(i) "v_and_or_b32 %" #i ", %" #i ", %8, %8\n"
#define S_BFEV(i) "v_bfe_u32 %" #i ", %" #i ", %8, %8\n"
#define S_XORS(i) "v_xor_b32 %" #i ", %9, %" #i "\n"
#define S_ANDLIT(i) "v_and_b32 %" #i ", 0xff00ff, %" #i "\n"
#define S_ANDINL(i) "v_and_b32 %" #i ", 15, %" #i "\n"
#define S_MOVS(i) "v_mov_b32 %" #i ", %9\n"
#define S_BITOPC(i) "v_bitop3_b32 %" #i ", %" #i ", 15, %8 bitop3:0x96\n"
#define S_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define S_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", %8, %8\n"
#define S_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define S_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %8\n"
#define S_XORE64(i) "v_xor_b32_e64 %" #i ", %8, %" #i "\n"
#define S_XORE64S(i) "v_xor_b32_e64 %" #i ", %9, %" #i "\n"
#define S_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define S_MADU16(i) "v_mad_u32_u16 %" #i ", %" #i ", %8, %8 op_sel:[1,0,0,0]\n"
#define S_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define S_MADU16LO(i) "v_mad_u32_u16 %" #i ", %" #i ", %8, %8\n"
#define S_MULU24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define S_LSHLSDWA(i) "v_lshlrev_b32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define S_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define S_PKMAD(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %8\n"

KERNEL(k_xor, OP8(S_XOR))
KERNEL(k_bitop3, OP8(S_BITOP))
KERNEL(k_bitop3_sgpr, OP8(S_BITOPS))
KERNEL(k_perm, OP8(S_PERM))
KERNEL(k_alignbit, OP8(S_ALIGN))
KERNEL(k_and_or, OP8(S_ANDOR))
KERNEL(k_lshl_or, OP8(S_LSHLOR))
KERNEL(k_bfe, OP8(S_BFE))
KERNEL(k_fma, OP8(S_FMA))
KERNEL(k_add, OP8(S_ADD))
KERNEL(k_lshr, OP8(S_LSHR))
KERNEL(k_mov_sdwa, OP8(S_SDWA))
KERNEL(k_or_sdwa, OP8(S_ORSDWA))
KERNEL(k_mov, OP8(S_MOV))
KERNEL(k_permv, OP8(S_PERMV))
KERNEL(k_alignv, OP8(S_ALIGNV))
KERNEL(k_andorv, OP8(S_ANDORV))
KERNEL(k_bfev, OP8(S_BFEV))
KERNEL(k_xors, OP8(S_XORS))
KERNEL(k_andlit, OP8(S_ANDLIT))
KERNEL(k_andinl, OP8(S_ANDINL))
KERNEL(k_movs, OP8(S_MOVS))
KERNEL(k_bitopc, OP8(S_BITOPC))
KERNEL(k_cndmask, OP8(S_CNDMASK))
KERNEL(k_lshladd, OP8(S_LSHLADD))
KERNEL(k_add3, OP8(S_ADD3))
KERNEL(k_or3, OP8(S_OR3))
KERNEL(k_xore64, OP8(S_XORE64))
KERNEL(k_xore64s, OP8(S_XORE64S))
KERNEL(k_pkadd, OP8(S_PKADD))
KERNEL(k_madu16, OP8(S_MADU16))
KERNEL(k_madu16lo, OP8(S_MADU16LO))
KERNEL(k_madu24, OP8(S_MADU24))
KERNEL(k_mulu24, OP8(S_MULU24))
KERNEL(k_lshlsdwa, OP8(S_LSHLSDWA))
KERNEL(k_addsdwa, OP8(S_ADDSDWA))
KERNEL(k_pkmad, OP8(S_PKMAD))
#ifdef HAVE_XOR3
KERNEL(k_xor3, OP8(S_XOR3))
#endif

template <typename K>
void run(const char* name, K kern, uint32_t* out, int blocks, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 1u, 2u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 1u, 2u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  double waves = (double)blocks * threads / 64;
  double winstr = waves * REP * 16.0;
  // wave-instr per ns per CU -> x (1/2.4) = per clk at 2.4 GHz
  double per_ns_cu = winstr / (ms * 1e6) / 256.0;
  printf("%-14s blocks %4d x %4d: %.3f ms  %.3f wave-instr/ns/CU  = %.2f lanes/clk/SIMD @2.4GHz (%.2f @2.1)\n", name, blocks, threads, ms,
         per_ns_cu, per_ns_cu / 2.4 * 64 / 4, per_ns_cu / 2.1 * 64 / 4);
}

int main() {
  uint32_t* out; hipMalloc(&out, 1024 * 1024 * 4 * 4);
  for (int threads : {1024}) {
    int blocks = threads == 1024 ? 512 : 2048;
    run("v_xor_b32", k_xor, out, blocks, threads);
    run("v_add_u32", k_add, out, blocks, threads);
    run("v_mov_b32", k_mov, out, blocks, threads);
    run("v_lshrrev", k_lshr, out, blocks, threads);
    run("v_bitop3", k_bitop3, out, blocks, threads);
    run("v_bitop3+sgpr", k_bitop3_sgpr, out, blocks, threads);
    run("v_perm_b32", k_perm, out, blocks, threads);
    run("v_alignbit", k_alignbit, out, blocks, threads);
    run("v_and_or", k_and_or, out, blocks, threads);
    run("v_lshl_or", k_lshl_or, out, blocks, threads);
    run("v_bfe_u32", k_bfe, out, blocks, threads);
    run("v_fma_f32", k_fma, out, blocks, threads);
    run("v_mov_sdwa", k_mov_sdwa, out, blocks, threads);
    run("v_or_sdwa", k_or_sdwa, out, blocks, threads);
    run("v_perm vvv", k_permv, out, blocks, threads);
    run("v_alignbit vvv", k_alignv, out, blocks, threads);
    run("v_and_or vvv", k_andorv, out, blocks, threads);
    run("v_bfe vvv", k_bfev, out, blocks, threads);
    run("v_xor e32 sgpr", k_xors, out, blocks, threads);
    run("v_and literal", k_andlit, out, blocks, threads);
    run("v_and inline", k_andinl, out, blocks, threads);
    run("v_mov sgpr", k_movs, out, blocks, threads);
    run("v_bitop3 inl", k_bitopc, out, blocks, threads);
    run("v_cndmask vcc", k_cndmask, out, blocks, threads);
    run("v_lshl_add vvv", k_lshladd, out, blocks, threads);
    run("v_add3 vvv", k_add3, out, blocks, threads);
    run("v_or3 vvv", k_or3, out, blocks, threads);
    run("v_mad_u32_u16 hi", k_madu16, out, blocks, threads);
    run("v_mad_u32_u16 lo", k_madu16lo, out, blocks, threads);
    run("v_mad_u32_u24", k_madu24, out, blocks, threads);
    run("v_mul_u32_u24", k_mulu24, out, blocks, threads);
    run("v_lshl sdwa b1", k_lshlsdwa, out, blocks, threads);
    run("v_add sdwa w1", k_addsdwa, out, blocks, threads);
    run("v_pk_mad_u16", k_pkmad, out, blocks, threads);
    run("v_xor e64", k_xore64, out, blocks, threads);
    run("v_xor e64 sgpr", k_xore64s, out, blocks, threads);
    run("v_pk_add_u16", k_pkadd, out, blocks, threads);
#ifdef HAVE_XOR3
    run("v_xor3", k_xor3, out, blocks, threads);
#endif
  }
  return 0;
}
