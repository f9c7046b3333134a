// Device-side building blocks for the MI355X (gfx950) DPF/DCF evaluation engine.
//
//  * AES-128 with the round function as T-table lookups in LDS.  The 256-entry
//    Te0 table is replicated once per lane of a wavefront ("row" = 64 x 4 B =
//    256 B per table index), so lane l always hits bank l % 32 and a wave64
//    ds_read_b32 is conflict-free whatever the 64 indices are (ds_read_b32 is
//    serviced as two 32-lane groups; bank = (addr / 4) % 32).  The 256-byte row
//    also makes the LDS byte address  idx << 8 | lane * 4  a single v_perm_b32.
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

namespace fssamd {

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
};

constexpr int kLdsTableBytes = 256 * 256;  // 64 KiB: 256 indices x 64 lane copies x 4 B

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

struct Lds {
  const uint8_t* tab;  // LDS table base
  uint32_t lo;         // (lane & 63) * 4
};

// Fill the workgroup's LDS table.  Call from every thread, then __syncthreads().
__device__ __forceinline__ void lds_table_fill(uint32_t* lds, const uint32_t* __restrict__ te0) {
  for (int i = threadIdx.x; i < 256 * 64; i += blockDim.x) lds[i] = te0[i >> 6];
}

// One table lookup: Te0[byte K of w].
template <int K>
__device__ __forceinline__ uint32_t te(const Lds& L, uint32_t w) {
  // result bytes: [0] = lane offset, [1] = byte K of w, [2], [3] = 0
  constexpr uint32_t sel = 0x0c0c0000u | ((4u + K) << 8);
  uint32_t addr = __builtin_amdgcn_perm(w, L.lo, sel);
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
template <bool SEL>
__device__ __forceinline__ void aes_mmo(const Lds& L, const uint32_t* __restrict__ key, uint32_t mask,
                                        const uint32_t in[4], uint32_t out[4]) {
  constexpr int RS = SEL ? 8 : 4;  // key words per round
  uint32_t s0, s1, s2, s3;
  {
    const uint32_t* k = key + round_anchor(in[0]);
    if (SEL) {
      s0 = xor3(in[0], k[0], mask & k[4]);
      s1 = xor3(in[1], k[1], mask & k[5]);
      s2 = xor3(in[2], k[2], mask & k[6]);
      s3 = xor3(in[3], k[3], mask & k[7]);
    } else {
      s0 = in[0] ^ k[0]; s1 = in[1] ^ k[1]; s2 = in[2] ^ k[2]; s3 = in[3] ^ k[3];
    }
  }
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    const uint32_t* k = key + RS * r + round_anchor(s0);
    uint32_t a0 = te<0>(L, s0), a1 = te<0>(L, s1), a2 = te<0>(L, s2), a3 = te<0>(L, s3);
    uint32_t b0 = te<1>(L, s1), b1 = te<1>(L, s2), b2 = te<1>(L, s3), b3 = te<1>(L, s0);
    uint32_t c0 = te<2>(L, s2), c1 = te<2>(L, s3), c2 = te<2>(L, s0), c3 = te<2>(L, s1);
    uint32_t d0 = te<3>(L, s3), d1 = te<3>(L, s0), d2 = te<3>(L, s1), d3 = te<3>(L, s2);
    uint32_t x0 = xor3(a0, rotl32(b0, 8), rotl32(c0, 16));
    uint32_t x1 = xor3(a1, rotl32(b1, 8), rotl32(c1, 16));
    uint32_t x2 = xor3(a2, rotl32(b2, 8), rotl32(c2, 16));
    uint32_t x3 = xor3(a3, rotl32(b3, 8), rotl32(c3, 16));
    if (SEL) {
      s0 = xor3(x0, and_xor(mask, k[4], rotl32(d0, 24)), k[0]);
      s1 = xor3(x1, and_xor(mask, k[5], rotl32(d1, 24)), k[1]);
      s2 = xor3(x2, and_xor(mask, k[6], rotl32(d2, 24)), k[2]);
      s3 = xor3(x3, and_xor(mask, k[7], rotl32(d3, 24)), k[3]);
    } else {
      s0 = xor3(x0, rotl32(d0, 24), k[0]);
      s1 = xor3(x1, rotl32(d1, 24), k[1]);
      s2 = xor3(x2, rotl32(d2, 24), k[2]);
      s3 = xor3(x3, rotl32(d3, 24), k[3]);
    }
  }
  // Last round: S-box only (byte 1 of a Te0 entry is S), ShiftRows, AddRoundKey, then ^ in.
  {
    const uint32_t* k = key + RS * 10 + round_anchor(s0);
    uint32_t a0 = te<0>(L, s0), a1 = te<0>(L, s1), a2 = te<0>(L, s2), a3 = te<0>(L, s3);
    uint32_t b0 = te<1>(L, s1), b1 = te<1>(L, s2), b2 = te<1>(L, s3), b3 = te<1>(L, s0);
    uint32_t c0 = te<2>(L, s2), c1 = te<2>(L, s3), c2 = te<2>(L, s0), c3 = te<2>(L, s1);
    uint32_t d0 = te<3>(L, s3), d1 = te<3>(L, s0), d2 = te<3>(L, s1), d3 = te<3>(L, s2);
    // p = S(a) | S(b) << 8 ; q = S(c) << 16 | S(d) << 24   (perm: bytes 4-7 = src0, 0-3 = src1)
    constexpr uint32_t selp = 0x0c0c0501u, selq = 0x05010c0cu;
    uint32_t p0 = __builtin_amdgcn_perm(b0, a0, selp), q0 = __builtin_amdgcn_perm(d0, c0, selq);
    uint32_t p1 = __builtin_amdgcn_perm(b1, a1, selp), q1 = __builtin_amdgcn_perm(d1, c1, selq);
    uint32_t p2 = __builtin_amdgcn_perm(b2, a2, selp), q2 = __builtin_amdgcn_perm(d2, c2, selq);
    uint32_t p3 = __builtin_amdgcn_perm(b3, a3, selp), q3 = __builtin_amdgcn_perm(d3, c3, selq);
    if (SEL) {
      out[0] = xor3(xor3(p0, q0, in[0]), mask & k[4], k[0]);
      out[1] = xor3(xor3(p1, q1, in[1]), mask & k[5], k[1]);
      out[2] = xor3(xor3(p2, q2, in[2]), mask & k[6], k[2]);
      out[3] = xor3(xor3(p3, q3, in[3]), mask & k[7], k[3]);
    } else {
      out[0] = xor3(p0, q0, in[0]) ^ k[0];
      out[1] = xor3(p1, q1, in[1]) ^ k[1];
      out[2] = xor3(p2, q2, in[2]) ^ k[2];
      out[3] = xor3(p3, q3, in[3]) ^ k[3];
    }
  }
}

// ------------------------------------------------------------------ output groups
// fss::group::Bytes (include/fss/group/bytes.cuh:19-43) and fss::group::Uint<T, mod>
// (include/fss/group/uint.cuh:27-88).  The group is a wave-uniform runtime value.
// A group value is accumulated in `Acc` with positive terms only and negated once
// at the end for party 1: negation is a homomorphism of every group here and the
// arithmetic is exact, so this equals the reference's term-by-term signs bit for
// bit.  Uint<uint8/16/32/64> all accumulate mod 2^64 and are reduced in finish();
// Uint<__uint128_t, 1 << 127> is arithmetic mod 2^127 on x | y<<32 | z<<64 | (w>>1)<<96.
struct Acc {
  uint32_t v[4] = {0, 0, 0, 0};

  static __device__ __forceinline__ unsigned __int128 to128(const uint32_t b[4]) {
    return (unsigned __int128)b[0] | ((unsigned __int128)b[1] << 32) | ((unsigned __int128)b[2] << 64) |
           ((unsigned __int128)b[3] << 96);
  }
  __device__ __forceinline__ void set128(unsigned __int128 r) {
    v[0] = (uint32_t)r; v[1] = (uint32_t)(r >> 32); v[2] = (uint32_t)(r >> 64); v[3] = (uint32_t)(r >> 96);
  }
  // b: 16-byte buffer whose control bit the caller has cleared where the reference does.
  __device__ __forceinline__ void add(int g, const uint32_t b[4]) {
    if (g == G_BYTES) {
      v[0] ^= b[0]; v[1] ^= b[1]; v[2] ^= b[2]; v[3] ^= b[3];
    } else if (g == G_U128M127) {
      uint32_t t[4] = {b[0], b[1], b[2], b[3] >> 1};  // uint.cuh:58-62
      set128(to128(v) + to128(t));
    } else {
      uint64_t r = ((uint64_t)v[0] | ((uint64_t)v[1] << 32)) + ((uint64_t)b[0] | ((uint64_t)b[1] << 32));
      v[0] = (uint32_t)r; v[1] = (uint32_t)(r >> 32);
    }
  }
  __device__ __forceinline__ void sub(int g, const uint32_t b[4]) {
    if (g == G_BYTES) {
      v[0] ^= b[0]; v[1] ^= b[1]; v[2] ^= b[2]; v[3] ^= b[3];
    } else if (g == G_U128M127) {
      uint32_t t[4] = {b[0], b[1], b[2], b[3] >> 1};
      set128(to128(v) - to128(t));
    } else {
      uint64_t r = ((uint64_t)v[0] | ((uint64_t)v[1] << 32)) - ((uint64_t)b[0] | ((uint64_t)b[1] << 32));
      v[0] = (uint32_t)r; v[1] = (uint32_t)(r >> 32);
    }
  }
  __device__ __forceinline__ void add_if(int g, uint32_t m, const uint32_t b[4]) {
    uint32_t t[4] = {m & b[0], m & b[1], m & b[2], m & b[3]};
    add(g, t);
  }
  __device__ __forceinline__ uint4 finish(int g, bool neg) const {
    if (g == G_BYTES) return make_uint4(v[0], v[1], v[2], v[3]);  // bytes.cuh:26-28: -x == x
    if (g == G_U128M127) {
      unsigned __int128 r = to128(v);
      if (neg) r = (unsigned __int128)0 - r;
      r &= (((unsigned __int128)1) << 127) - 1;
      return make_uint4((uint32_t)r, (uint32_t)(r >> 32), (uint32_t)(r >> 64), ((uint32_t)(r >> 96)) << 1);  // uint.cuh:76-82
    }
    uint64_t r = (uint64_t)v[0] | ((uint64_t)v[1] << 32);
    if (neg) r = 0ull - r;
    if (g == G_U8) r &= 0xffull;
    else if (g == G_U16) r &= 0xffffull;
    else if (g == G_U32) r &= 0xffffffffull;
    return make_uint4((uint32_t)r, (uint32_t)(r >> 32), 0, 0);
  }
};

}  // namespace fssamd
