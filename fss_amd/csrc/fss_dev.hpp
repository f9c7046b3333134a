// Device-side building blocks for the MI355X (gfx950) DPF/DCF evaluation engine.
//
//  * AES-128 with the round function as T-table lookups in LDS: four tables T0..T3 (Te0 rotated by
//    0/8/16/24 bits), each entry replicated once per bank (32 copies), so lane l always reads bank
//    l % 32 and a wave64 ds_read_b32 is conflict-free whatever the 64 indices are (it is serviced as two
//    32-lane groups; bank = (addr / 4) % 32).  Layout and addressing: "LDS table layouts" below.
//  * ChaCha20 and the keyed BLAKE3 compression (the reference's other PRG and the VDPF hashes) as
//    straight VALU code, behind the same PRG interface (`Lds` / `ChaChaL` overloads).
//  * WaveQ: work distribution for the persistent kernels (per-workgroup LDS counter).
//  * State columns are kept as little-endian words, i.e. exactly the int4's
//    x, y, z, w -- no byte swaps around the cipher (the reference's software AES
//    loads big-endian words, include/fss/prg/aes128_mmo_soft.cuh:113-129; the
//    ciphertext bytes are the same).
//  * Round keys are wave-uniform, live in the kernel-argument segment and are read
//    with one scalar load per round; a per-lane choice between two keys (left /
//    right child) is  rkA ^ (mask & (rkA ^ rkB))  folded into the column XOR with
//    v_bitop3_b32.
//
// Replaces, for the device: fss::prg::Aes128Soft (include/fss/prg/aes128_mmo_soft.cuh:185-218)
// and its Encrypt (:124-174); computes fss::prg::Aes128Mmo<mul>::Gen
// (include/fss/prg/aes128_mmo.cuh:72-93) output by output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

namespace fssamd {

// Launch-plan tunables (FSS_AMD_ILP, FSS_AMD_DYN, FSS_AMD_FUSED, ... -- the A/B switches of tools/*.sh and of
// tests/test_gpu_errors.py::test_parity_under_forced_tunables) are honoured only in a process that sets
// FSS_AMD_TUNE=1.  A default process reads that one variable, once, and its launch plan does not depend on anything
// else in the caller's environment.
inline const char* tune_env(const char* name) {
  static const bool on = [] { const char* e = getenv("FSS_AMD_TUNE"); return e != nullptr && atoi(e) == 1; }();
  return on ? getenv(name) : nullptr;
}

// Group ids (shared with include/fss_amd.h).
enum : int { G_BYTES = 0, G_U8 = 1, G_U16 = 2, G_U32 = 3, G_U64 = 4, G_U128M127 = 5 };
enum : int { S_DPF = 0, S_DCF = 1, S_HT = 2, S_GROTTO = 3 };

// Wave-uniform constants of one engine context.  Passed to every kernel BY VALUE:
// kernel arguments live in the constant address space, so every round-key read is
// a scalar load (s_load_dword*) into SGPRs -- a pointer into ordinary global
// memory would be read with vector loads and pin ~90 VGPRs per lane.
// Te0 (little-endian, bytes (2S, S, S, 3S), row 0 in the low byte) is a separate
// 1 KiB device array, only read once per workgroup to fill LDS.
struct DevCtx {
  uint32_t rk[4][44];     // round keys of AES keys 0..3, little-endian words (uniform-key AES)
  // Per-lane two-key AES: round r holds {rkA[4r..4r+3], (rkA ^ rkB)[4r..4r+3]}, one
  // s_load_dwordx8 per round.
  uint32_t sel01[11][8];  // A = key 0, B = key 1   DPF / Grotto: left / right seed
  uint32_t sel02[11][8];  // A = key 0, B = key 2   DCF seed:  left / right
  uint32_t sel13[11][8];  // A = key 1, B = key 3   DCF value: left / right
  uint32_t hash_key[4];
  const uint32_t* self_dev;  // device-memory copy of this struct (for per-lane vector loads of round keys)
  uint32_t chacha_nonce[2];  // ChaCha PRG (contexts created with fss_amd_ctx_set_chacha)
  int chacha_rounds;
  uint32_t blake3_iv[8];     // VDPF: key of the keyed BLAKE3 hashes H and H' (fss::hash::Blake3::iv_)
  uint64_t uint_mod;         // modulus of Uint<T, mod> for T up to 64 bits; 0 = 2^(8 sizeof T)
  int hash_kind;             // VDPF hash pair: 0 = keyed BLAKE3 (blake3_iv[0..7]), 1 = SHA-256 keyed with blake3_iv[0..3]
};


// v_bitop3_b32 truth tables (a = 0xF0, b = 0xCC, c = 0xAA).
constexpr uint32_t kXor3 = 0xF0 ^ 0xCC ^ 0xAA;        // a ^ b ^ c
constexpr uint32_t kAndXor = (0xF0 & 0xCC) ^ 0xAA;    // (a & b) ^ c

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, kXor3);
}
__device__ __forceinline__ uint32_t and_xor(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, kAndXor);
}
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) {
  return __builtin_amdgcn_alignbit(x, x, 32 - n);
}

// ---- LDS table layout ---------------------------------------------------------
// Four tables T0..T3 (T_t = Te0 rotated left by 8t bits), each replicated 32x so that
// lane l always reads bank l % 32 (conflict-free for any 64 indices); 128 KiB, one
// 1024-thread workgroup per CU.  Rows are 256 B:
//   region A (first 64 KiB):  row idx = [T0[idx] x 32 lanes | T1[idx] x 32 lanes]
//   region B (second 64 KiB): row idx = [T2[idx] x 32 lanes | T3[idx] x 32 lanes]
// so a lookup address is  region << 16 | idx << 8 | half << 7 | (lane & 31) << 2  and
// the round needs no rotate instructions.  On gfx950 v_perm_b32 / v_alignbit_b32 /
// anything with an SGPR operand issue at 4 cycles per wave while v_bitop3_b32, shifts
// and 2-input logic on VGPRs issue at 2 (tools/ubench/valu_rate.hip), and the walk is
// VALU-issue bound, so dropping the 12 rotates per round is worth ~1.4x (the single-table
// form with rotates, and the other variants measured and rejected, are listed in
// tools/ubench/REJECTED_VARIANTS.md).
constexpr int kLdsWords = 32768;

// a ? b : c bitwise  (a = 0xF0, b = 0xCC, c = 0xAA)
constexpr uint32_t kSelect = (0xF0 & 0xCC) | (~0xF0 & 0xAA & 0xFF);
constexpr uint32_t kAndOr = (0xF0 & 0xCC) | 0xAA;  // (a & b) | c

// A constant the compiler must keep in a VGPR (an SGPR or literal operand would put
// the consuming VALU op on the 4-cycle path).
__device__ __forceinline__ uint32_t vconst(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

struct Lds {
  const uint8_t* tab;  // LDS table base
  uint32_t lo[4];      // per table: region << 16 | half << 7 | (lane & 31) << 2
  uint32_t m_ff00, m_00ff, m_ff0000, m_ffff;
};

__device__ __forceinline__ Lds make_lds(const uint32_t* lds) {
  Lds L;
  L.tab = reinterpret_cast<const uint8_t*>(lds);
  const uint32_t l4 = (threadIdx.x & 31u) << 2;
  L.lo[0] = l4; L.lo[1] = l4 | 0x80u; L.lo[2] = l4 | 0x10000u; L.lo[3] = l4 | 0x10080u;
  L.m_ff00 = vconst(0xff00u); L.m_00ff = vconst(0xffu); L.m_ff0000 = vconst(0xff0000u); L.m_ffff = vconst(0xffffu);
  return L;
}

// Fill the workgroup's LDS tables.  Call from every thread, then __syncthreads().
__device__ __forceinline__ void lds_table_fill(uint32_t* lds, const uint32_t* __restrict__ te0) {
  if (blockDim.x == 1024) {
    // word i = tid + 1024 k holds entry ((tid >> 6) + 16 k) & 255 of table ((k >> 4) << 1) | ((tid >> 5) & 1):
    // 16 distinct entries per lane, requested together (one memory latency instead of 32 in turn;
    // 0.8 us per workgroup instead of 3.0, tools/ubench/fill_rate.hip).
    const int tid = threadIdx.x, odd = (tid >> 5) & 1;
    uint32_t t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = te0[((tid >> 6) + 16 * k) & 255];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int tbl = ((k >> 4) << 1) | odd;
      lds[tid + 1024 * k] = __builtin_amdgcn_alignbit(t[k & 15], t[k & 15], (32 - 8 * tbl) & 31);
    }
    return;
  }
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) {
    uint32_t t = te0[(i >> 6) & 255];
    int tbl = ((i >> 14) << 1) | ((i >> 5) & 1);
    lds[i] = tbl ? __builtin_amdgcn_alignbit(t, t, 32 - 8 * tbl) : t;
  }
}

#define FSS_LDS_PROLOGUE(te0)                                              \
  __shared__ __attribute__((aligned(16))) uint32_t lds_te[kLdsWords];     \
  lds_table_fill(lds_te, te0);                                            \
  __syncthreads();                                                        \
  const Lds L = make_lds(lds_te)

// Stands in for `Lds` in kernels instantiated for the ChaCha PRG: no tables, no LDS.
struct ChaChaL {};

// Kernel prologue for either PRG: LT = Lds fills the AES tables, LT = ChaChaL does nothing.
#define FSS_PRG_PROLOGUE(LT, te0)                                            \
  LT L{};                                                                    \
  if constexpr (std::is_same_v<LT, Lds>) {                                   \
    __shared__ __attribute__((aligned(16))) uint32_t lds_te[kLdsWords];     \
    lds_table_fill(lds_te, te0);                                            \
    __syncthreads();                                                        \
    L = make_lds(lds_te);                                                \
  }

// One table lookup: T_T[byte K of w]  (T_T = Te0 rotated left by 8T bits).
template <int K, int T>
__device__ __forceinline__ uint32_t lk(const Lds& L, uint32_t w) {
  uint32_t addr;
  if (K == 1) {
    addr = __builtin_amdgcn_bitop3_b32(w, L.m_ff00, L.lo[T], kAndOr);  // byte 1 is already at idx << 8
  } else {
    // result bytes: [0] = lo byte 0, [1] = byte K of w, [2] = lo byte 2 (region), [3] = 0
    constexpr uint32_t sel = 0x0c020000u | ((4u + K) << 8);
    addr = __builtin_amdgcn_perm(w, L.lo[T], sel);
  }
  return *reinterpret_cast<const uint32_t*>(L.tab + addr);
}

// Opaque zero in an SGPR that depends on a lane value: placed at the head of an
// AES round it keeps that round's key load inside the round (issued next to the
// round's LDS lookups, consumed after them) instead of letting LICM hoist all 88
// key dwords out of the level loop, which overflows the SGPR file into
// v_writelane / v_readlane spills.
__device__ __forceinline__ uint32_t round_anchor(uint32_t dep) {
  uint32_t z = 0;
  asm volatile("" : "+s"(z) : "v"(dep));
  return z;
}

// AES-128 encryption of `in` followed by the Matyas-Meyer-Oseas feed-forward:
//   out = AES_k(in) ^ in.
// SEL = false: `key` = 44 round-key words of one wave-uniform key.
// SEL = true:  `key` = DevCtx::selXY, per lane k = key A where mask == 0 and key B
// where mask == ~0.
// Pointer to the key words of round r, readable one round at a time.
__device__ __forceinline__ const uint32_t* round_key(const Lds&, const uint32_t* key, int word, uint32_t dep) {
  return key + word + round_anchor(dep);
}

// LSB3: only bit 0 of out[3] (the control bit of a child) is wanted; out[3]'s other bits and out[0..2] are
// unspecified.  That bit comes from one S-box lookup of the last round, which needs one column of round 9:
// 133 lookups instead of 160 (the unused ones of the two last rounds disappear as dead code).
template <bool SEL, bool LSB3 = false>
__device__ __forceinline__ void aes_mmo(const Lds& L, const uint32_t* __restrict__ key, uint32_t mask,
                                        const uint32_t in[4], uint32_t out[4]) {
  constexpr int RS = SEL ? 8 : 4;  // key words per round
  uint32_t s0, s1, s2, s3;
  uint32_t kc[RS];  // key words of the round being computed, fetched one phase ahead
  {
    const uint32_t* k = round_key(L, key, 0, in[0]);
    const uint32_t* k1 = round_key(L, key, RS, in[1]);
    if (SEL) {
      s0 = xor3(in[0], k[0], mask & k[4]);
      s1 = xor3(in[1], k[1], mask & k[5]);
      s2 = xor3(in[2], k[2], mask & k[6]);
      s3 = xor3(in[3], k[3], mask & k[7]);
    } else {
      s0 = in[0] ^ k[0]; s1 = in[1] ^ k[1]; s2 = in[2] ^ k[2]; s3 = in[3] ^ k[3];
    }
#pragma unroll
    for (int j = 0; j < RS; ++j) kc[j] = k1[j];
  }
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    uint32_t a0 = lk<0, 0>(L, s0), a1 = lk<0, 0>(L, s1), a2 = lk<0, 0>(L, s2), a3 = lk<0, 0>(L, s3);
    uint32_t b0 = lk<1, 1>(L, s1), b1 = lk<1, 1>(L, s2), b2 = lk<1, 1>(L, s3), b3 = lk<1, 1>(L, s0);
    uint32_t c0 = lk<2, 2>(L, s2), c1 = lk<2, 2>(L, s3), c2 = lk<2, 2>(L, s0), c3 = lk<2, 2>(L, s1);
    uint32_t d0 = lk<3, 3>(L, s3), d1 = lk<3, 3>(L, s0), d2 = lk<3, 3>(L, s1), d3 = lk<3, 3>(L, s2);
    // The next round's key load is anchored on a looked-up value: it is issued right after
    // this round's LDS wait, and has the whole VALU phase plus the next LDS phase to land.
    const uint32_t* kn = round_key(L, key, RS * (r + 1), (LSB3 && r == 9) ? a3 : a0);
    uint32_t knx[RS];
#pragma unroll
    for (int j = 0; j < RS; ++j) knx[j] = kn[j];
    uint32_t x0 = xor3(a0, b0, c0);
    uint32_t x1 = xor3(a1, b1, c1);
    uint32_t x2 = xor3(a2, b2, c2);
    uint32_t x3 = xor3(a3, b3, c3);
    if (SEL) {
      s0 = xor3(x0, and_xor(mask, kc[4], d0), kc[0]);
      s1 = xor3(x1, and_xor(mask, kc[5], d1), kc[1]);
      s2 = xor3(x2, and_xor(mask, kc[6], d2), kc[2]);
      s3 = xor3(x3, and_xor(mask, kc[7], d3), kc[3]);
    } else {
      s0 = xor3(x0, d0, kc[0]);
      s1 = xor3(x1, d1, kc[1]);
      s2 = xor3(x2, d2, kc[2]);
      s3 = xor3(x3, d3, kc[3]);
    }
#pragma unroll
    for (int j = 0; j < RS; ++j) kc[j] = knx[j];
  }
  // Last round: S-box only, ShiftRows, AddRoundKey, then ^ in.
  if constexpr (LSB3) {
    static_assert(!SEL || !LSB3, "control-bit-only blocks use one key");
    out[0] = out[1] = out[2] = 0u;
    out[3] = xor3(lk<0, 2>(L, s3), in[3], kc[3]);  // byte 0 of word 3 = S(byte 0 of s3) ^ k ^ in
  } else {
    const uint32_t* k = kc;
    // T2 has S in byte 0, T3 in byte 1, T0 in byte 2, T1 in byte 3: each lookup already
    // holds its S-box byte in place; three bitwise selects assemble the column.
    uint32_t a0 = lk<0, 2>(L, s0), a1 = lk<0, 2>(L, s1), a2 = lk<0, 2>(L, s2), a3 = lk<0, 2>(L, s3);
    uint32_t b0 = lk<1, 3>(L, s1), b1 = lk<1, 3>(L, s2), b2 = lk<1, 3>(L, s3), b3 = lk<1, 3>(L, s0);
    uint32_t c0 = lk<2, 0>(L, s2), c1 = lk<2, 0>(L, s3), c2 = lk<2, 0>(L, s0), c3 = lk<2, 0>(L, s1);
    uint32_t d0 = lk<3, 1>(L, s3), d1 = lk<3, 1>(L, s0), d2 = lk<3, 1>(L, s1), d3 = lk<3, 1>(L, s2);
#define FSS_COL(a, b, c, d)                                                           \
  __builtin_amdgcn_bitop3_b32(L.m_ffff, __builtin_amdgcn_bitop3_b32(L.m_00ff, a, b, kSelect), \
                              __builtin_amdgcn_bitop3_b32(L.m_ff0000, c, d, kSelect), kSelect)
    uint32_t p0 = FSS_COL(a0, b0, c0, d0), p1 = FSS_COL(a1, b1, c1, d1), p2 = FSS_COL(a2, b2, c2, d2),
             p3 = FSS_COL(a3, b3, c3, d3);
#undef FSS_COL
    if (SEL) {
      out[0] = xor3(and_xor(mask, k[4], p0), in[0], k[0]);
      out[1] = xor3(and_xor(mask, k[5], p1), in[1], k[1]);
      out[2] = xor3(and_xor(mask, k[6], p2), in[2], k[2]);
      out[3] = xor3(and_xor(mask, k[7], p3), in[3], k[3]);
    } else {
      out[0] = xor3(p0, in[0], k[0]);
      out[1] = xor3(p1, in[1], k[1]);
      out[2] = xor3(p2, in[2], k[2]);
      out[3] = xor3(p3, in[3], k[3]);
    }
  }
}

// ---- two interleaved AES streams ------------------------------------------------
// A single stream alternates "16 lookups in flight" and "VALU on the 16 results": within a wave
// the LDS latency is dead time, and with the 128 KiB tables only 4 waves share a SIMD.  aes_mmo2
// runs two independent blocks A and B half a round apart:
//     load A | VALU(B) + next addresses(B) | wait | load B | VALU(A) + next addresses(A) | wait | ...
// so each stream's lookups fly while the other stream's ~96 issue cycles of VALU run.  At every
// wait exactly one stream (and its round-key scalar load) is outstanding, so `lgkmcnt(0)` is the
// exact wait; the waits are explicit and fenced with sched_barrier so the compiler keeps the
// order (its own waitcnt pass then has nothing left to insert).
template <int K, int T>
__device__ __forceinline__ uint32_t lk_addr(const Lds& L, uint32_t w) {
  if (K == 1) return __builtin_amdgcn_bitop3_b32(w, L.m_ff00, L.lo[T], kAndOr);
  constexpr uint32_t sel = 0x0c020000u | ((4u + K) << 8);
  return __builtin_amdgcn_perm(w, L.lo[T], sel);
}

// Addresses of the 16 lookups of one round: [0..3] byte 0 of s0..s3, [4..7] byte 1 of s1,s2,s3,s0,
// [8..11] byte 2 of s2,s3,s0,s1, [12..15] byte 3 of s3,s0,s1,s2.  LAST picks the tables that hold
// S in the output byte position.
template <bool LAST>
__device__ __forceinline__ void aes_addr16(const Lds& L, const uint32_t s[4], uint32_t ad[16]) {
  constexpr int T0 = LAST ? 2 : 0, T1 = LAST ? 3 : 1, T2 = LAST ? 0 : 2, T3 = LAST ? 1 : 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ad[j] = lk_addr<0, T0>(L, s[j]);
    ad[4 + j] = lk_addr<1, T1>(L, s[(j + 1) & 3]);
    ad[8 + j] = lk_addr<2, T2>(L, s[(j + 2) & 3]);
    ad[12 + j] = lk_addr<3, T3>(L, s[(j + 3) & 3]);
  }
}

__device__ __forceinline__ void aes_load16(const Lds& L, const uint32_t ad[16], uint32_t v[16]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const uint32_t*>(L.tab + ad[j]);
}

template <bool SEL>
__device__ __forceinline__ void aes_mix(const uint32_t v[16], const uint32_t* k, const uint32_t* kd, uint32_t mask,
                                        uint32_t s[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t x = xor3(v[j], v[4 + j], v[8 + j]), d = v[12 + j];
    s[j] = SEL ? xor3(x, and_xor(mask, kd[j], d), k[j]) : xor3(x, d, k[j]);
  }
}

// OW (output words wanted): 4 = all of out[0..3]; 2 / 1 = out[0..OW-1] whole plus bit 0 of out[3]
// (the control bit) -- what a leaf needs when the output group reads only the low 8 / 4 bytes of the seed
// (Uint<uint64_t> and narrower, uint.cuh:49-68).  The other outputs are unspecified and their last-round
// lookups (7 resp. 11 of the 16) disappear as dead code.
template <bool SEL, bool LSB3 = false, int OW = 4>
__device__ __forceinline__ void aes_fin(const Lds& L, const uint32_t v[16], const uint32_t* k, const uint32_t* kd,
                                        uint32_t mask, const uint32_t in[4], uint32_t out[4]) {
  if constexpr (LSB3) {  // see aes_mmo: v[3] = the last-round lookup of byte 0 of s3
    out[0] = out[1] = out[2] = 0u;
    out[3] = xor3(v[3], in[3], k[3]);
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (OW < 4) {
      static_assert(!SEL || OW == 4, "narrow outputs are for uniform-key blocks");
      if (j >= OW) {
        out[j] = j == 3 ? xor3(v[3], in[3], k[3]) : 0u;  // byte 0 of word 3 only (as LSB3)
        continue;
      }
    }
    uint32_t p = __builtin_amdgcn_bitop3_b32(L.m_ffff, __builtin_amdgcn_bitop3_b32(L.m_00ff, v[j], v[4 + j], kSelect),
                                             __builtin_amdgcn_bitop3_b32(L.m_ff0000, v[8 + j], v[12 + j], kSelect),
                                             kSelect);
    out[j] = SEL ? xor3(and_xor(mask, kd[j], p), in[j], k[j]) : xor3(p, in[j], k[j]);
  }
}

#define FSS_FENCE() __builtin_amdgcn_sched_barrier(0)
#define FSS_DRAIN_LGKM()                  \
  do {                                    \
    FSS_FENCE();                          \
    __builtin_amdgcn_s_waitcnt(0xC07F);   \
    FSS_FENCE();                          \
  } while (0)

// outA = AES_{keyA}(inA) ^ inA and outB = AES_{keyB}(inB) ^ inB, interleaved.  keyA / keyB and
// maskA / maskB as for aes_mmo<SEL>; they may be the same or different key schedules.
template <bool SEL, bool LSB3 = false, int OW = 4>
__device__ __forceinline__ void aes_mmo2(const Lds& L, const uint32_t* __restrict__ keyA,
                                         const uint32_t* __restrict__ keyB, uint32_t maskA, uint32_t maskB,
                                         const uint32_t inA[4], const uint32_t inB[4], uint32_t outA[4],
                                         uint32_t outB[4]) {
  constexpr int RS = SEL ? 8 : 4;
  constexpr int RL = RS;  // words fetched per main round by the scalar loads
  uint32_t sA[4], sB[4], vA[16], vB[16], kA[RL], kB[RL];
  {
    const uint32_t* k0 = round_key(L, keyA, 0, inA[0]);
    const uint32_t* k1 = round_key(L, keyB, 0, inB[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sA[j] = SEL ? xor3(inA[j], k0[j], maskA & k0[4 + j]) : (inA[j] ^ k0[j]);
      sB[j] = SEL ? xor3(inB[j], k1[j], maskB & k1[4 + j]) : (inB[j] ^ k1[j]);
    }
  }
  aes_addr16<false>(L, sA, vA);
  FSS_FENCE();
  {
    const uint32_t* kn = round_key(L, keyA, RS, sA[0]);
    aes_load16(L, vA, vA);
#pragma unroll
    for (int j = 0; j < RL; ++j) kA[j] = kn[j];
  }
  FSS_FENCE();
  aes_addr16<false>(L, sB, vB);
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    FSS_DRAIN_LGKM();  // A's round-r lookups and keys have landed; nothing else is outstanding
    {
      const uint32_t* kn = round_key(L, keyB, RS * r, sB[0]);
      aes_load16(L, vB, vB);
#pragma unroll
      for (int j = 0; j < RL; ++j) kB[j] = kn[j];
    }
    FSS_FENCE();
    aes_mix<SEL>(vA, kA, kA + 4, maskA, sA);  // B's lookups fly during A's VALU
    if (r < 9) aes_addr16<false>(L, sA, vA);
    else aes_addr16<true>(L, sA, vA);
    FSS_DRAIN_LGKM();  // B's round-r lookups and keys
    {
      const uint32_t* kn = round_key(L, keyA, RS * (r + 1), sA[(LSB3 && r == 9) ? 3 : 0]);
      aes_load16(L, vA, vA);
#pragma unroll
      for (int j = 0; j < RL; ++j) kA[j] = kn[j];
    }
    FSS_FENCE();
    aes_mix<SEL>(vB, kB, kB + 4, maskB, sB);
    if (r < 9) aes_addr16<false>(L, sB, vB);
    else aes_addr16<true>(L, sB, vB);
  }
  uint32_t kLA[RS], kLB[RS];  // last-round keys, all RS words from the scalar copy
  FSS_DRAIN_LGKM();  // A's last-round lookups
  {
    const uint32_t* kn = round_key(L, keyB, RS * 10, sB[LSB3 ? 3 : 0]);
    const uint32_t* kna = round_key(L, keyA, RS * 10, sB[LSB3 ? 3 : 1]);
    aes_load16(L, vB, vB);
#pragma unroll
    for (int j = 0; j < RS; ++j) { kLB[j] = kn[j]; kLA[j] = kna[j]; }
  }
  FSS_DRAIN_LGKM();
  aes_fin<SEL, LSB3, OW>(L, vA, kLA, kLA + 4, maskA, inA, outA);
  aes_fin<SEL, LSB3, OW>(L, vB, kLB, kLB + 4, maskB, inB, outB);
}

// The DevCtx a kernel received by value as its FIRST argument, addressed in the kernarg segment
// (offset 0) so that its fields stay scalar loads whatever the optimizer makes of &K.
__device__ __forceinline__ const DevCtx* first_kernarg(const DevCtx& K) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)K;
  return (const DevCtx*)__builtin_amdgcn_kernarg_segment_ptr();
#else
  return &K;  // host pass of the single-source compile; never executed
#endif
}

// ------------------------------------------------------------------ ChaCha PRG
// fss::prg::ChaCha<mul, rounds> (include/fss/prg/chacha.cuh:24-128): one ChaCha permutation of
// [constant | seed | seed | 0, 0, nonce] per Gen, XOR feed-forward, rows 0..3 are the outputs
// (mul = 1: row 1; mul = 2: rows 0, 1; mul = 4: rows 0..3).  Pure VALU, no tables.
__device__ __forceinline__ void chacha_qr(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
  a += b; d ^= a; d = rotl32(d, 16);
  c += d; b ^= c; b = rotl32(b, 12);
  a += b; d ^= a; d = rotl32(d, 8);
  c += d; b ^= c; b = rotl32(b, 7);
}

// rows[r][j] after the feed-forward XORs of Gen (chacha.cuh:115-126).  WIDE: the 32-byte-key
// constant of mul = 4, otherwise the 16-byte-key constant of mul = 1, 2.
template <bool WIDE>
__device__ __forceinline__ void chacha_rows(const DevCtx* __restrict__ c, const uint32_t seed[4], uint32_t x[16]) {
  constexpr uint32_t k0 = 0x61707865u, k1 = WIDE ? 0x3320646eu : 0x3120646eu, k2 = WIDE ? 0x79622d32u : 0x79622d36u,
                     k3 = 0x6b206574u;
  const uint32_t n0 = c->chacha_nonce[0], n1 = c->chacha_nonce[1];
  x[0] = k0; x[1] = k1; x[2] = k2; x[3] = k3;
#pragma unroll
  for (int j = 0; j < 4; ++j) { x[4 + j] = seed[j]; x[8 + j] = seed[j]; }
  x[12] = 0; x[13] = 0; x[14] = n0; x[15] = n1;
  const int rounds = c->chacha_rounds;
#pragma unroll 1
  for (int i = 0; i < rounds; i += 2) {
    chacha_qr(x[0], x[4], x[8], x[12]);
    chacha_qr(x[1], x[5], x[9], x[13]);
    chacha_qr(x[2], x[6], x[10], x[14]);
    chacha_qr(x[3], x[7], x[11], x[15]);
    chacha_qr(x[0], x[5], x[10], x[15]);
    chacha_qr(x[1], x[6], x[11], x[12]);
    chacha_qr(x[2], x[7], x[8], x[13]);
    chacha_qr(x[3], x[4], x[9], x[14]);
  }
  x[0] ^= k0; x[1] ^= k1; x[2] ^= k2; x[3] ^= k3;
#pragma unroll
  for (int j = 0; j < 4; ++j) { x[4 + j] ^= seed[j]; x[8 + j] ^= seed[j]; }
  x[14] ^= n0; x[15] ^= n1;
}

// ------------------------------------------------------------------ dynamic work distribution
// A persistent grid whose waves each own a fixed 1/Nth of the batch finishes ragged: a SIMD issues its
// oldest wave first, so that wave retires at about half of the kernel time and the younger ones finish
// on a half-empty SIMD (measured: mean wave lifetime 71 % of the kernel, SQ_WAVE_CYCLES).  So the batch is
// cut into chunks of 64 units (one per lane), workgroup b owns chunks b, b + G, b + 2G, ... (G = grid size:
// every CU gets the same share), and the waves of a workgroup take their workgroup's chunks one at a time
// from a counter in LDS: all 16 waves of a CU stay resident until the CU's share is used up (+15 % on the
// headline walk).  dyn = 0: every wave walks its own fixed chunks (small batches, where a wave would get
// fewer than ~4 chunks and the imbalance of whole chunks costs more than the ragged finish).
#define FSS_WAVEQ_PROLOGUE()            \
  __shared__ uint32_t waveq_ctr_;       \
  if (threadIdx.x == 0) waveq_ctr_ = 0; \
  __syncthreads()
// Next chunk index (global numbering) for this wave; wave-uniform.
__device__ __forceinline__ int64_t waveq_fetch(uint32_t* lds_ctr) {
  uint32_t v = 0;
  if ((threadIdx.x & 63) == 0) v = atomicAdd(lds_ctr, 1u);
  v = __builtin_amdgcn_readfirstlane(v);
  return (int64_t)blockIdx.x + (int64_t)gridDim.x * v;
}
// Loop control shared by the persistent kernels: UNITS_PER_LANE units per lane and chunk, `unit0` = first
// unit of this lane in the current chunk.  Usage:
//   FSS_WAVEQ_PROLOGUE();  WaveQ q(dyn, &waveq_ctr_, total_units, units_per_lane);
//   for (int64_t u; q.next(u);) { ... lane handles units u + j * q.stride(), j < units_per_lane ... }
//
// Launch-wide tail (gq != nullptr).  Equal shares per CU still finish ragged ACROSS CUs: the XCDs do not hold the
// same clock under load (in-kernel clocks 1.90 - 2.13 GHz across the workgroups of one independent-keys launch,
// 2.32 - 2.39 for a shared key), so the slowest CU's workgroup ends 3 % (shared key) to 11 % (independent keys)
// after the fastest (per-wave stamps, tools/keys_clock.py).  So only the first `static_chunks` chunks are dealt out
// per workgroup; a workgroup that has used up its share takes the remaining chunks one at a time from a counter in
// device memory (gq[0]) that all workgroups share -- a few thousand device-scope atomics per launch, each requested a
// whole chunk ahead of its use.  Requests to one word serialise at a few tens of millions per second, so the host
// only enables the tail where a chunk lasts tens of microseconds (the point walks from ~24 AES blocks per lane and
// chunk on, the Grotto walk; not the EvalAll stages: Half-Tree 2^25 leaves 0.53 -> 0.82 ms with it).  Claiming a
// block of chunks (one per wave) per request -- first ticket of a block makes the request, the others pick the base
// up from LDS -- removes that limit but balances a whole round of a workgroup at a time: measured on one box, it gave
// a third of the per-chunk tail's gain on the long walks and nothing on short walks or stages, so it is not kept
// (profiles/r2/tail_queue_blocked_ab.txt).  gq[1] counts the workgroups that have left the loop; the last one
// zeroes both words, so the pair is ready for the next launch on the same stream (the host hands every stream its
// own pair).
struct WaveQ {
  uint32_t* ctr;
  int64_t nchunks, cur, per_chunk;
  bool dyn;
  uint32_t* gq;
  int64_t static_chunks;
  __device__ __forceinline__ int64_t fetch() {
    int64_t c = waveq_fetch(ctr);
    if (gq != nullptr && c >= static_chunks) {
      uint32_t g = 0;
      if ((threadIdx.x & 63) == 0) g = atomicAdd(gq, 1u);
      c = static_chunks + (int64_t)__builtin_amdgcn_readfirstlane(g);
    }
    return c;
  }
  __device__ __forceinline__ WaveQ(int dyn_, uint32_t* lds_ctr, int64_t total_units, int units_per_lane,
                                   uint32_t* gq_ = nullptr, int64_t static_chunks_ = 0)
      : ctr(lds_ctr), per_chunk(64 * (int64_t)units_per_lane), dyn(dyn_ != 0), gq(dyn_ != 0 ? gq_ : nullptr),
        static_chunks(static_chunks_) {
    nchunks = (total_units + per_chunk - 1) / per_chunk;
    // static: the wave's own chunks, spaced by the number of waves in the grid
    cur = dyn ? fetch() : ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  }
  // first unit of this lane in the next chunk of this wave; false when the wave is done
  __device__ __forceinline__ bool next(int64_t& unit0) {
    if (cur >= nchunks) return false;
    unit0 = cur * per_chunk + (threadIdx.x & 63);
    cur = dyn ? fetch() : cur + (int64_t)gridDim.x * (blockDim.x >> 6);
    return true;
  }
  // The chunk the NEXT call of next() will hand out (already claimed: `cur` is requested a chunk ahead), without
  // claiming another one: lets a walk request that chunk's inputs before it starts on the current chunk.
  __device__ __forceinline__ bool peek(int64_t& unit0) const {
    if (cur >= nchunks) return false;
    unit0 = cur * per_chunk + (threadIdx.x & 63);
    return true;
  }
  // After the loop, every wave of the launch (a workgroup barrier inside): hand the launch-wide counter back zeroed.
  // One count per workgroup, not per wave: 4096 requests to one word at the end of a launch would take tens of
  // microseconds by themselves.
  __device__ __forceinline__ void done() {
    if (gq == nullptr) return;  // wave-uniform, launch-uniform
    __syncthreads();            // every wave of this workgroup has made its last request
    if (threadIdx.x == 0 && atomicAdd(gq + 1, 1u) == gridDim.x - 1) {  // ... and so has every other workgroup
      gq[0] = 0u;
      gq[1] = 0u;
    }
  }
};

// Orders this wave's LDS accesses around a hand-over between its lanes (the LDS executes a wave's
// instructions in order; this keeps the compiler from moving them across the hand-over).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Rotating wave priority, for launches too small for WaveQ's dynamic distribution: the four waves that
// share a SIMD take turns, level by level, at being preferred by the instruction arbiter instead of the
// oldest always winning, so they finish closer together (+6 % at 2^20 points, +9 % at 2^24 when static;
// nothing on top of the dynamic distribution).
__device__ __forceinline__ void rotate_prio(int step) {
  switch (((threadIdx.x >> 8) + step) & 3) {  // waves w, w+4, w+8, w+12 of a workgroup share a SIMD
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// Progress-ranked wave priority (rot_prio = 2): closed loop instead of taking turns.  Every wave publishes how many
// levels it has walked; its priority is the number of waves of its SIMD that are ahead of it, so whoever falls behind
// is preferred until it has caught up.  Per-wave stamps showed why turns are not enough: with fixed shares the four
// waves of a SIMD still leave in order of age, 72 / 55 / 27 / 8 us before the end of a 320 us launch (2^20 points),
// and the LDS is not kept busy by the ones that remain.
#define FSS_PROGRESS_PROLOGUE()                                   \
  __shared__ __attribute__((aligned(16))) uint32_t wave_prog_[16]; \
  if (threadIdx.x < 16) wave_prog_[threadIdx.x] = 0
// every_mask: the exchange happens when (steps done & every_mask) == 0 -- 3 for the point walks (a step = one level).
__device__ __forceinline__ void rank_prio(uint32_t* prog, uint32_t& mine, uint32_t slack, uint32_t every_mask = 3u) {
  // The LDS is the pipe this kernel is bound by, so the exchange is made cheap: every fourth level, one single-lane
  // write and one four-lane read (a full-wave 16-byte broadcast read per level cost 3.6 % at 2^24 points).
  ++mine;
  if (mine & every_mask) return;
  const int w = threadIdx.x >> 6, simd = w & 3, age = w >> 2;  // waves w, w+4, w+8, w+12 of a workgroup share a SIMD
  const int lane = threadIdx.x & 63;
  uint32_t other = 0;
  if (lane < 4) {
    if (lane == age) prog[simd * 4 + age] = mine;
    other = *reinterpret_cast<const volatile uint32_t*>(prog + simd * 4 + lane);
  }
  // waves within `slack` levels of each other are left to the arbiter
  const int ahead = __popcll(__ballot(other > mine + slack));
  switch (ahead) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// ------------------------------------------------------------------ BLAKE3 (VDPF hashes)
// fss::hash::Blake3 (include/fss/hash/blake3.cuh:13-172): one keyed-hash compression (7 rounds) of a
// single block, flags CHUNK_START | CHUNK_END | ROOT | KEYED_HASH, counter 0; out = first 8 words of the
// output (v[0..7] ^ v[8..15]).  Fully unrolled so the message schedule is register renaming and the
// zero message words of the 32-byte inputs fold away.  Pure VALU like ChaCha.
__device__ __forceinline__ void blake3_g(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, uint32_t x, uint32_t y) {
  a += b + x; d = rotl32(d ^ a, 16);
  c += d; b = rotl32(b ^ c, 20);
  a += b + y; d = rotl32(d ^ a, 24);
  c += d; b = rotl32(b ^ c, 25);
}
__device__ __forceinline__ void blake3_compress8(const DevCtx* __restrict__ c, const uint32_t msg[16], uint32_t block_len,
                                                 uint32_t out[8]) {
  constexpr uint32_t kFlags = 1u | 2u | 8u | 16u;
  uint32_t v[16], m[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = c->blake3_iv[i];
  v[8] = 0x6A09E667u; v[9] = 0xBB67AE85u; v[10] = 0x3C6EF372u; v[11] = 0xA54FF53Au;
  v[12] = 0; v[13] = 0; v[14] = block_len; v[15] = kFlags;
#pragma unroll
  for (int i = 0; i < 16; ++i) m[i] = msg[i];
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    blake3_g(v[0], v[4], v[8], v[12], m[0], m[1]);
    blake3_g(v[1], v[5], v[9], v[13], m[2], m[3]);
    blake3_g(v[2], v[6], v[10], v[14], m[4], m[5]);
    blake3_g(v[3], v[7], v[11], v[15], m[6], m[7]);
    blake3_g(v[0], v[5], v[10], v[15], m[8], m[9]);
    blake3_g(v[1], v[6], v[11], v[12], m[10], m[11]);
    blake3_g(v[2], v[7], v[8], v[13], m[12], m[13]);
    blake3_g(v[3], v[4], v[9], v[14], m[14], m[15]);
    if (r < 6) {  // blake3.cuh:43-52
      uint32_t t[16] = {m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]};
#pragma unroll
      for (int i = 0; i < 16; ++i) m[i] = t[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = v[i] ^ v[8 + i];
}
// XorHashable: H(a, b) = compress(a with lsb 0 | b) | compress(a with lsb 1 | b), block_len 32 (blake3.cuh:157-169)
__device__ __forceinline__ void blake3_xorhash(const DevCtx* __restrict__ c, const uint32_t a[4], const uint32_t b[4],
                                               uint32_t out[16]) {
  uint32_t m[16] = {a[0], a[1], a[2], a[3] & ~1u, b[0], b[1], b[2], b[3], 0, 0, 0, 0, 0, 0, 0, 0};
  blake3_compress8(c, m, 32, out);
  m[3] |= 1u;
  blake3_compress8(c, m, 32, out + 8);
}

// ------------------------------------------------------------------ SHA-256 (VDPF hashes, second pair)
// fss::hash::Sha256 (include/fss/hash/sha256.cuh:15-62): EVP_Digest(key | message) with a 16-byte key --
// host-only in the reference; here FIPS 180-4 on the VALU, fully unrolled, 16-word rolling schedule.
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
__device__ __forceinline__ void sha256_compress(uint32_t h[8], const uint32_t wi[16]) {
  constexpr uint32_t Kc[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t w[16], a[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = wi[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = h[i];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    if (i >= 16) {
      const uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      const uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
      const uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
    }
    const uint32_t S1 = xor3(rotr32(a[4], 6), rotr32(a[4], 11), rotr32(a[4], 25));
    const uint32_t ch = __builtin_amdgcn_bitop3_b32(a[4], a[5], a[6], 0xCA);  // e ? f : g
    const uint32_t t1 = a[7] + S1 + ch + Kc[i] + w[i & 15];
    const uint32_t S0 = xor3(rotr32(a[0], 2), rotr32(a[0], 13), rotr32(a[0], 22));
    const uint32_t mj = __builtin_amdgcn_bitop3_b32(a[0], a[1], a[2], 0xE8);  // majority
    a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + S0 + mj;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] += a[i];
}
__device__ __forceinline__ void sha256_init(uint32_t h[8]) {
  h[0] = 0x6a09e667u; h[1] = 0xbb67ae85u; h[2] = 0x3c6ef372u; h[3] = 0xa54ff53au;
  h[4] = 0x510e527fu; h[5] = 0x9b05688cu; h[6] = 0x1f83d9abu; h[7] = 0x5be0cd19u;
}
// digest of key | a | b (48 bytes, one block) as 8 little-endian int4 words
__device__ __forceinline__ void sha256_48(const DevCtx* __restrict__ c, const uint32_t a[4], const uint32_t b[4], uint32_t out[8]) {
  uint32_t w[16], h[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) { w[i] = __builtin_bswap32(c->blake3_iv[i]); w[4 + i] = __builtin_bswap32(a[i]); w[8 + i] = __builtin_bswap32(b[i]); }
  w[12] = 0x80000000u; w[13] = 0; w[14] = 0; w[15] = 48 * 8;
  sha256_init(h);
  sha256_compress(h, w);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = __builtin_bswap32(h[i]);
}
// digest of key | m[0..15] (80 bytes, two blocks)
__device__ __forceinline__ void sha256_80(const DevCtx* __restrict__ c, const uint32_t m[16], uint32_t out[8]) {
  uint32_t w[16], h[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = __builtin_bswap32(c->blake3_iv[i]);
#pragma unroll
  for (int i = 0; i < 12; ++i) w[4 + i] = __builtin_bswap32(m[i]);
  sha256_init(h);
  sha256_compress(h, w);
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = __builtin_bswap32(m[12 + i]);
  w[4] = 0x80000000u;
#pragma unroll
  for (int i = 5; i < 15; ++i) w[i] = 0;
  w[15] = 80 * 8;
  sha256_compress(h, w);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = __builtin_bswap32(h[i]);
}

// The VDPF hashes by the context's hash pair: H(a, b) -> 16 words (XorHashable), H'(m) -> 8 words (Hashable).
__device__ __forceinline__ void vdpf_xorhash(const DevCtx* __restrict__ c, const uint32_t a[4], const uint32_t b[4],
                                             uint32_t out[16]) {
  if (c->hash_kind == 1) {
    uint32_t a0[4] = {a[0], a[1], a[2], a[3] & ~1u}, a1[4] = {a[0], a[1], a[2], a[3] | 1u};
    sha256_48(c, a0, b, out);
    sha256_48(c, a1, b, out + 8);
  } else {
    blake3_xorhash(c, a, b, out);
  }
}
__device__ __forceinline__ void vdpf_hash(const DevCtx* __restrict__ c, const uint32_t m[16], uint32_t out[8]) {
  if (c->hash_kind == 1) sha256_80(c, m, out);
  else blake3_compress8(c, m, 64, out);
}

// ------------------------------------------------------------------ PRG interface
// What the walks need from a PRG, overloaded on the kernel's table type (Lds = AES-128-MMO in LDS,
// ChaChaL = ChaCha).  "take" = only the output(s) of the child selected per lane by mask
// (0 = left, ~0 = right); "gen" = all outputs.
//   mul 2 (DPF, Grotto): [left, right]              mul 4 (DCF): [sl, vl, sr, vr]   (dcf.cuh:223)
//   mul 1 (Half-Tree):   h = Gen(hash_key ^ node)[0]
__device__ __forceinline__ void prg_gen2(const Lds& L, const DevCtx* __restrict__ c, const uint32_t s[4], uint32_t l[4],
                                         uint32_t r[4]) {
  aes_mmo2<false>(L, c->rk[0], c->rk[1], 0, 0, s, s, l, r);
}
__device__ __forceinline__ void prg_take2(const Lds& L, const DevCtx* __restrict__ c, uint32_t m, const uint32_t s[4],
                                          uint32_t o[4]) {
  aes_mmo<true>(L, &c->sel01[0][0], m, s, o);
}
__device__ __forceinline__ void prg_take2x2(const Lds& L, const DevCtx* __restrict__ c, uint32_t ma, uint32_t mb,
                                            const uint32_t sa[4], const uint32_t sb[4], uint32_t oa[4], uint32_t ob[4]) {
  aes_mmo2<true>(L, &c->sel01[0][0], &c->sel01[0][0], ma, mb, sa, sb, oa, ob);
}
__device__ __forceinline__ void prg_gen4(const Lds& L, const DevCtx* __restrict__ c, const uint32_t s[4], uint32_t sl[4],
                                         uint32_t vl[4], uint32_t sr[4], uint32_t vr[4]) {
  aes_mmo2<false>(L, c->rk[0], c->rk[1], 0, 0, s, s, sl, vl);
  aes_mmo2<false>(L, c->rk[2], c->rk[3], 0, 0, s, s, sr, vr);
}
__device__ __forceinline__ void prg_take4(const Lds& L, const DevCtx* __restrict__ c, uint32_t m, const uint32_t s[4],
                                          uint32_t os[4], uint32_t ov[4]) {
  aes_mmo2<true>(L, &c->sel02[0][0], &c->sel13[0][0], m, m, s, s, os, ov);
}
// half of prg_gen4: the seed and value block of ONE side, the same for every lane (`right` uniform: plain schedules)
__device__ __forceinline__ void prg_gen4_side(const Lds& L, const DevCtx* __restrict__ c, const uint32_t s[4], bool right,
                                              uint32_t os[4], uint32_t ov[4]) {
  aes_mmo2<false>(L, c->rk[right ? 2 : 0], c->rk[right ? 3 : 1], 0, 0, s, s, os, ov);
}
__device__ __forceinline__ void prg_hash(const Lds& L, const DevCtx* __restrict__ c, const uint32_t in[4], uint32_t h[4]) {
  aes_mmo<false>(L, c->rk[0], 0, in, h);
}
__device__ __forceinline__ void prg_hash2(const Lds& L, const DevCtx* __restrict__ c, const uint32_t ia[4],
                                          const uint32_t ib[4], uint32_t ha[4], uint32_t hb[4]) {
  aes_mmo2<false>(L, c->rk[0], c->rk[0], 0, 0, ia, ib, ha, hb);
}
// Leaf-level blocks whose outputs feed a narrow output group: words [0, OW) and the control bit only (aes_fin).
template <int OW>
__device__ __forceinline__ void prg_gen2_narrow(const Lds& L, const DevCtx* __restrict__ c, const uint32_t s[4], uint32_t l[4],
                                                uint32_t r[4]) {
  aes_mmo2<false, false, OW>(L, c->rk[0], c->rk[1], 0, 0, s, s, l, r);
}
template <int OW>
__device__ __forceinline__ void prg_hash2_narrow(const Lds& L, const DevCtx* __restrict__ c, const uint32_t ia[4],
                                                 const uint32_t ib[4], uint32_t ha[4], uint32_t hb[4]) {
  aes_mmo2<false, false, OW>(L, c->rk[0], c->rk[0], 0, 0, ia, ib, ha, hb);
}
// Control bits only (Grotto: leaves of the tree, left siblings of the path walk): bit 0 of the results is
// bit 0 of word 3 of Gen(s)[0] / Gen(s)[1]; the other bits are unspecified.
__device__ __forceinline__ uint32_t prg_tbit_left(const Lds& L, const DevCtx* __restrict__ c, const uint32_t s[4]) {
  uint32_t o[4];
  aes_mmo<false, true>(L, c->rk[0], 0, s, o);
  return o[3];
}
__device__ __forceinline__ void prg_tbits2(const Lds& L, const DevCtx* __restrict__ c, const uint32_t s[4], uint32_t& tl,
                                           uint32_t& tr) {
  uint32_t l[4], r[4];
  aes_mmo2<false, true>(L, c->rk[0], c->rk[1], 0, 0, s, s, l, r);
  tl = l[3]; tr = r[3];
}

__device__ __forceinline__ void prg_gen2(const ChaChaL&, const DevCtx* __restrict__ c, const uint32_t s[4], uint32_t l[4],
                                         uint32_t r[4]) {
  uint32_t x[16];
  chacha_rows<false>(c, s, x);
#pragma unroll
  for (int j = 0; j < 4; ++j) { l[j] = x[j]; r[j] = x[4 + j]; }
}
__device__ __forceinline__ void prg_take2(const ChaChaL& L, const DevCtx* __restrict__ c, uint32_t m, const uint32_t s[4],
                                          uint32_t o[4]) {
  uint32_t l[4], r[4];
  prg_gen2(L, c, s, l, r);  // one permutation yields both children
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = __builtin_amdgcn_bitop3_b32(m, r[j], l[j], kSelect);
}
__device__ __forceinline__ void prg_take2x2(const ChaChaL& L, const DevCtx* __restrict__ c, uint32_t ma, uint32_t mb,
                                            const uint32_t sa[4], const uint32_t sb[4], uint32_t oa[4], uint32_t ob[4]) {
  prg_take2(L, c, ma, sa, oa);
  prg_take2(L, c, mb, sb, ob);
}
__device__ __forceinline__ void prg_gen4(const ChaChaL&, const DevCtx* __restrict__ c, const uint32_t s[4], uint32_t sl[4],
                                         uint32_t vl[4], uint32_t sr[4], uint32_t vr[4]) {
  uint32_t x[16];
  chacha_rows<true>(c, s, x);
#pragma unroll
  for (int j = 0; j < 4; ++j) { sl[j] = x[j]; vl[j] = x[4 + j]; sr[j] = x[8 + j]; vr[j] = x[12 + j]; }
}
__device__ __forceinline__ void prg_take4(const ChaChaL& L, const DevCtx* __restrict__ c, uint32_t m, const uint32_t s[4],
                                          uint32_t os[4], uint32_t ov[4]) {
  uint32_t sl[4], vl[4], sr[4], vr[4];
  prg_gen4(L, c, s, sl, vl, sr, vr);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    os[j] = __builtin_amdgcn_bitop3_b32(m, sr[j], sl[j], kSelect);
    ov[j] = __builtin_amdgcn_bitop3_b32(m, vr[j], vl[j], kSelect);
  }
}
__device__ __forceinline__ void prg_hash(const ChaChaL&, const DevCtx* __restrict__ c, const uint32_t in[4], uint32_t h[4]) {
  uint32_t x[16];
  chacha_rows<false>(c, in, x);  // mul = 1: row 1 (chacha.cuh:116-118)
#pragma unroll
  for (int j = 0; j < 4; ++j) h[j] = x[4 + j];
}
__device__ __forceinline__ void prg_hash2(const ChaChaL& L, const DevCtx* __restrict__ c, const uint32_t ia[4],
                                          const uint32_t ib[4], uint32_t ha[4], uint32_t hb[4]) {
  prg_hash(L, c, ia, ha);
  prg_hash(L, c, ib, hb);
}
template <int OW>
__device__ __forceinline__ void prg_gen2_narrow(const ChaChaL& L, const DevCtx* __restrict__ c, const uint32_t s[4],
                                                uint32_t l[4], uint32_t r[4]) {
  prg_gen2(L, c, s, l, r);  // one permutation yields every word anyway
}
template <int OW>
__device__ __forceinline__ void prg_hash2_narrow(const ChaChaL& L, const DevCtx* __restrict__ c, const uint32_t ia[4],
                                                 const uint32_t ib[4], uint32_t ha[4], uint32_t hb[4]) {
  prg_hash2(L, c, ia, ib, ha, hb);
}
__device__ __forceinline__ void prg_tbits2(const ChaChaL& L, const DevCtx* __restrict__ c, const uint32_t s[4], uint32_t& tl,
                                           uint32_t& tr) {
  uint32_t l[4], r[4];
  prg_gen2(L, c, s, l, r);
  tl = l[3]; tr = r[3];
}
// ------------------------------------------------------------------ output groups
// fss::group::Bytes (include/fss/group/bytes.cuh:19-43) and fss::group::Uint<T, mod>
// (include/fss/group/uint.cuh:27-88).  The group is a wave-uniform runtime value.
// A group value is accumulated in `Acc` with positive terms only and negated once
// at the end for party 1: negation is a homomorphism of every group here and the
// arithmetic is exact, so this equals the reference's term-by-term signs bit for
// bit.  Uint<uint8/16/32/64> all accumulate mod 2^64 and are reduced in finish();
// Uint<__uint128_t, 1 << 127> is arithmetic mod 2^127 on x | y<<32 | z<<64 | (w>>1)<<96.
// A group as the kernels see it: the id and, for Uint<T, mod> with a modulus other than 2^(8 sizeof T)
// (uint.cuh:17-30, T up to 64 bits), that modulus; mod == 0 is the wrapping group.  Wave-uniform.
struct Grp {
  int id;
  uint64_t mod;
};
__device__ __forceinline__ Grp make_grp(int id, const DevCtx* __restrict__ c) {
  return Grp{id, (id >= G_U8 && id <= G_U64) ? c->uint_mod : 0ull};
}

struct Acc {
  uint32_t v[4] = {0, 0, 0, 0};

  static __device__ __forceinline__ unsigned __int128 to128(const uint32_t b[4]) {
    return (unsigned __int128)b[0] | ((unsigned __int128)b[1] << 32) | ((unsigned __int128)b[2] << 64) |
           ((unsigned __int128)b[3] << 96);
  }
  __device__ __forceinline__ void set128(unsigned __int128 r) {
    v[0] = (uint32_t)r; v[1] = (uint32_t)(r >> 32); v[2] = (uint32_t)(r >> 64); v[3] = (uint32_t)(r >> 96);
  }
  // Uint<T, mod>::From for mod != 0 (uint.cuh:36-53): the low sizeof(T) bytes, reduced.
  static __device__ __forceinline__ uint64_t from_mod(const Grp& g, const uint32_t b[4]) {
    uint64_t x = (uint64_t)b[0] | ((uint64_t)b[1] << 32);
    if (g.id == G_U8) x &= 0xffull;
    else if (g.id == G_U16) x &= 0xffffull;
    else if (g.id == G_U32) x &= 0xffffffffull;
    return x % g.mod;
  }
  __device__ __forceinline__ void add_mod(const Grp& g, uint64_t x) {  // uint.cuh:20-24, both operands < mod
    uint64_t a = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    a = a >= g.mod - x ? a + x - g.mod : a + x;
    v[0] = (uint32_t)a; v[1] = (uint32_t)(a >> 32);
  }
  // b: 16-byte buffer whose control bit the caller has cleared where the reference does.
  __device__ __forceinline__ void add(const Grp& g, const uint32_t b[4]) {
    if (g.id == G_BYTES) {
      v[0] ^= b[0]; v[1] ^= b[1]; v[2] ^= b[2]; v[3] ^= b[3];
    } else if (g.id == G_U128M127) {
      uint32_t t[4] = {b[0], b[1], b[2], b[3] >> 1};  // uint.cuh:58-62
      set128(to128(v) + to128(t));
    } else if (g.mod) {
      add_mod(g, from_mod(g, b));
    } else {
      uint64_t r = ((uint64_t)v[0] | ((uint64_t)v[1] << 32)) + ((uint64_t)b[0] | ((uint64_t)b[1] << 32));
      v[0] = (uint32_t)r; v[1] = (uint32_t)(r >> 32);
    }
  }
  __device__ __forceinline__ void sub(const Grp& g, const uint32_t b[4]) {
    if (g.id == G_BYTES) {
      v[0] ^= b[0]; v[1] ^= b[1]; v[2] ^= b[2]; v[3] ^= b[3];
    } else if (g.id == G_U128M127) {
      uint32_t t[4] = {b[0], b[1], b[2], b[3] >> 1};
      set128(to128(v) - to128(t));
    } else if (g.mod) {
      const uint64_t x = from_mod(g, b);
      add_mod(g, x ? g.mod - x : 0ull);  // uint.cuh:26-31
    } else {
      uint64_t r = ((uint64_t)v[0] | ((uint64_t)v[1] << 32)) - ((uint64_t)b[0] | ((uint64_t)b[1] << 32));
      v[0] = (uint32_t)r; v[1] = (uint32_t)(r >> 32);
    }
  }
  __device__ __forceinline__ void add_if(const Grp& g, uint32_t m, const uint32_t b[4]) {
    uint32_t t[4] = {m & b[0], m & b[1], m & b[2], m & b[3]};
    add(g, t);
  }
  __device__ __forceinline__ uint4 finish(const Grp& g, bool neg) const {
    if (g.id == G_BYTES) return make_uint4(v[0], v[1], v[2], v[3]);  // bytes.cuh:26-28: -x == x
    if (g.id == G_U128M127) {
      unsigned __int128 r = to128(v);
      if (neg) r = (unsigned __int128)0 - r;
      r &= (((unsigned __int128)1) << 127) - 1;
      return make_uint4((uint32_t)r, (uint32_t)(r >> 32), (uint32_t)(r >> 64), ((uint32_t)(r >> 96)) << 1);  // uint.cuh:76-82
    }
    uint64_t r = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    if (g.mod) {
      if (neg && r) r = g.mod - r;
      return make_uint4((uint32_t)r, (uint32_t)(r >> 32), 0, 0);
    }
    if (neg) r = 0ull - r;
    if (g.id == G_U8) r &= 0xffull;
    else if (g.id == G_U16) r &= 0xffffull;
    else if (g.id == G_U32) r &= 0xffffffffull;
    return make_uint4((uint32_t)r, (uint32_t)(r >> 32), 0, 0);
  }
};

}  // namespace fssamd
