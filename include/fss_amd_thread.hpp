// fss_amd_thread.hpp -- Gen / Eval of one key by ONE GPU thread, for calls from inside a caller's own kernel
// (the reference's members are __host__ __device__ and its samples use them that way: samples/dpf_dcf_gpu.cu:39-80,
// src/bench_gpu.cu:85-95).  Included by fss_amd.hpp; the scheme classes forward their device-side Gen / Eval here.
//
// Everything below is __device__ code written against the two hooks only -- Prgable (`prg.Gen(seed)`) and Groupable
// (`From / + / - / Into`) -- so a caller's own PRG or group type works as well as the built-in ones, provided its
// members are callable on the device.  It is the per-thread form: one thread walks one key.  With the AES PRGs the
// tables are in LDS (the caller's for fss::prg::Aes128Soft, one bank-replicated copy per workgroup for fss::prg::Aes128Mmo), the round keys in
// scalar registers, and a level computes only the child it takes (GenSide).  The engine's batched kernels behind
// EvalPoints / fss::gpu::* (bank-conflict-free replicated tables, level-major keys) remain the fast forms; nothing here
// is used by them.
//
// Correction-word slots are addressed as raw int4 pairs (slot i = words 2i, 2i+1), the layout all three Cw structs
// share: word 0 = s, word 1 = v (DCF) or the flag byte tr / extra at byte 16 (DPF, Half-Tree).
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <cstdint>
#include <span>
#include <tuple>

namespace fss {

// ---------------------------------------------------------------- the VDPF hashes (hash.cuh:15-30, hash/blake3.cuh, hash/sha256.cuh)
// Hashable:    hash.Hash(span<const int4, 4>)        -> array<int4, 2>   (the paper's H': 64 B -> 32 B)
// XorHashable: hash.Hash(tuple<int4, const int4>)    -> array<int4, 4>   (the paper's H: two digests, the first word's
//              control bit 0 / 1 as domain separation)
// Both hash types are written from the published algorithms (BLAKE3 spec section 2; FIPS 180-4), callable on host and
// device.  The engine's kernels hash by themselves (fss_amd/csrc/fss_dev.hpp) from the same key material: the objects
// here are what a caller constructs (`fss::hash::Blake3 h{iv}`), what Vdpf's per-thread device forms call, and what the
// host side folds a proof with.
namespace hash {
namespace detail {
__host__ __device__ constexpr uint32_t Rotr(uint32_t v, int n) { return v >> n | v << (32 - n); }
// message word used at position i of round r: the spec's permutation applied r times
struct Blake3Schedule {
  uint8_t at[7][16];
};
constexpr Blake3Schedule BuildBlake3Schedule() {
  constexpr uint8_t perm[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
  Blake3Schedule s{};
  for (int i = 0; i < 16; ++i) s.at[0][i] = static_cast<uint8_t>(i);
  for (int r = 1; r < 7; ++r)
    for (int i = 0; i < 16; ++i) s.at[r][i] = s.at[r - 1][perm[i]];
  return s;
}
inline constexpr Blake3Schedule kBlake3Schedule = BuildBlake3Schedule();
// The first eight output words of ONE keyed-hash compression of a single-block message (flags CHUNK_START | CHUNK_END |
// ROOT | KEYED_HASH, counter 0): all either Hash needs.
__host__ __device__ inline void Blake3Block(const uint32_t key[8], const uint32_t m[16], uint32_t block_len, uint32_t out[8]) {
  uint32_t v[16] = {key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                    0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0u, 0u, block_len, 1u | 2u | 8u | 16u};
  auto mix = [&v](int a, int b, int c, int d, uint32_t x, uint32_t y) {
    v[a] += v[b] + x; v[d] = Rotr(v[d] ^ v[a], 16);
    v[c] += v[d];     v[b] = Rotr(v[b] ^ v[c], 12);
    v[a] += v[b] + y; v[d] = Rotr(v[d] ^ v[a], 8);
    v[c] += v[d];     v[b] = Rotr(v[b] ^ v[c], 7);
  };
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const uint8_t* w = kBlake3Schedule.at[r];
#pragma unroll
    for (int j = 0; j < 4; ++j) mix(j, 4 + j, 8 + j, 12 + j, m[w[2 * j]], m[w[2 * j + 1]]);                                   // columns
#pragma unroll
    for (int j = 0; j < 4; ++j) mix(j, 4 + (j + 1) % 4, 8 + (j + 2) % 4, 12 + (j + 3) % 4, m[w[8 + 2 * j]], m[w[9 + 2 * j]]);  // diagonals
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = v[i] ^ v[8 + i];
}
__host__ __device__ inline void Words(int4 v, uint32_t* w) {
  w[0] = static_cast<uint32_t>(v.x); w[1] = static_cast<uint32_t>(v.y); w[2] = static_cast<uint32_t>(v.z); w[3] = static_cast<uint32_t>(v.w);
}
__host__ __device__ inline int4 Int4(const uint32_t* w) {
  return int4{static_cast<int>(w[0]), static_cast<int>(w[1]), static_cast<int>(w[2]), static_cast<int>(w[3])};
}
// SHA-256 compression of one 64-byte block given as sixteen big-endian words
__host__ __device__ inline void Sha256Block(uint32_t h[8], const uint32_t block[16]) {
  constexpr uint32_t k[64] = {
      0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u,
      0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu,
      0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u,
      0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
      0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u,
      0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
      0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
  uint32_t w[64];
  for (int i = 0; i < 16; ++i) w[i] = block[i];
  for (int i = 16; i < 64; ++i)
    w[i] = w[i - 16] + (Rotr(w[i - 15], 7) ^ Rotr(w[i - 15], 18) ^ w[i - 15] >> 3) + w[i - 7] + (Rotr(w[i - 2], 17) ^ Rotr(w[i - 2], 19) ^ w[i - 2] >> 10);
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; ++i) {
    const uint32_t t1 = hh + (Rotr(e, 6) ^ Rotr(e, 11) ^ Rotr(e, 25)) + ((e & f) ^ (~e & g)) + k[i] + w[i];
    const uint32_t t2 = (Rotr(a, 2) ^ Rotr(a, 13) ^ Rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
// SHA-256 of `nwords` little-endian words of message bytes (48 or 80 bytes here), digest as little-endian words of its bytes
__host__ __device__ inline void Sha256Words(const uint32_t* msg, int nwords, uint32_t out[8]) {
  uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  const int blocks = (nwords * 4 + 9 + 63) / 64;
  for (int bk = 0; bk < blocks; ++bk) {
    uint32_t blk[16];
    for (int i = 0; i < 16; ++i) {
      const int wi = 16 * bk + i;
      blk[i] = wi < nwords ? __builtin_bswap32(msg[wi]) : (wi == nwords ? 0x80000000u : 0u);
    }
    if (bk == blocks - 1) blk[15] = static_cast<uint32_t>(nwords) * 32u;  // the length in bits (below 2^32)
    Sha256Block(h, blk);
  }
  for (int i = 0; i < 8; ++i) out[i] = __builtin_bswap32(h[i]);
}
}  // namespace detail

// fss::hash::Blake3 (hash/blake3.cuh:24-172): keyed BLAKE3, the caller's 32-byte IV as the key.
class Blake3 {
 public:
  int32_t iv[8];  // public: the engine's contexts are keyed by it
  static constexpr int kEngineHashKind = 0;
  __host__ __device__ explicit Blake3(std::span<const int4, 2> iv2) : iv{iv2[0].x, iv2[0].y, iv2[0].z, iv2[0].w, iv2[1].x, iv2[1].y, iv2[1].z, iv2[1].w} {}
  __host__ __device__ explicit Blake3(const int4 iv2[2]) : Blake3(std::span<const int4, 2>(iv2, 2)) {}
  // H': one compression of the 64-byte message, the first 32 bytes of its output (blake3.cuh:145-149)
  __host__ __device__ std::array<int4, 2> Hash(std::span<const int4, 4> msg) const {
    uint32_t k[8], m[16], o[8];
    for (int i = 0; i < 8; ++i) k[i] = static_cast<uint32_t>(iv[i]);
    for (int i = 0; i < 4; ++i) detail::Words(msg[i], m + 4 * i);
    detail::Blake3Block(k, m, 64, o);
    return {detail::Int4(o), detail::Int4(o + 4)};
  }
  // H: (a with control bit 0 | b) and (a with control bit 1 | b), 32-byte messages, 32 bytes each (blake3.cuh:160-171)
  __host__ __device__ std::array<int4, 4> Hash(std::tuple<int4, const int4> msg) const {
    uint32_t k[8], m[16] = {}, o0[8], o1[8];
    for (int i = 0; i < 8; ++i) k[i] = static_cast<uint32_t>(iv[i]);
    detail::Words(std::get<0>(msg), m);
    detail::Words(std::get<1>(msg), m + 4);
    m[3] &= ~1u;
    detail::Blake3Block(k, m, 32, o0);
    m[3] |= 1u;
    detail::Blake3Block(k, m, 32, o1);
    return {detail::Int4(o0), detail::Int4(o0 + 4), detail::Int4(o1), detail::Int4(o1 + 4)};
  }
  __host__ __device__ bool SameKey(const Blake3& o) const {
    bool same = true;
    for (int i = 0; i < 8; ++i) same = same && iv[i] == o.iv[i];
    return same;
  }
};
// fss::hash::Sha256 (hash/sha256.cuh:25-99): SHA-256 of the 16-byte key followed by the message.  Host-only in the
// reference (OpenSSL EVP_Digest); plain code here, callable on the device as well.
class Sha256 {
 public:
  int4 key;
  static constexpr int kEngineHashKind = 1;
  __host__ __device__ explicit Sha256(int4 key_) : key(key_) {}
  __host__ __device__ std::array<int4, 2> Hash(std::span<const int4, 4> msg) const {  // sha256.cuh:43-60: 80 bytes
    uint32_t m[20], o[8];
    detail::Words(key, m);
    for (int i = 0; i < 4; ++i) detail::Words(msg[i], m + 4 + 4 * i);
    detail::Sha256Words(m, 20, o);
    return {detail::Int4(o), detail::Int4(o + 4)};
  }
  __host__ __device__ std::array<int4, 4> Hash(std::tuple<int4, const int4> msg) const {  // sha256.cuh:69-90: two 48-byte messages
    uint32_t m[12], o0[8], o1[8];
    detail::Words(key, m);
    detail::Words(std::get<0>(msg), m + 4);
    detail::Words(std::get<1>(msg), m + 8);
    m[7] &= ~1u;
    detail::Sha256Words(m, 12, o0);
    m[7] |= 1u;
    detail::Sha256Words(m, 12, o1);
    return {detail::Int4(o0), detail::Int4(o0 + 4), detail::Int4(o1), detail::Int4(o1 + 4)};
  }
  __host__ __device__ bool SameKey(const Sha256& o) const { return key.x == o.key.x && key.y == o.key.y && key.z == o.key.z && key.w == o.key.w; }
};
}  // namespace hash

namespace thread {

__host__ __device__ inline int4 X4(int4 a, int4 b) { return int4{a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w}; }
__host__ __device__ inline bool Ctl(int4 v) { return v.w & 1; }
__host__ __device__ inline int4 WithCtl(int4 v, bool bit) {
  v.w = (v.w & ~1) | (bit ? 1 : 0);
  return v;
}
__host__ __device__ inline int4 If(bool c, int4 v) { return c ? v : int4{0, 0, 0, 0}; }
// bit `pos` of x, for every In up to __uint128_t
template <typename In>
__host__ __device__ inline bool BitAt(In x, int pos) {
  return static_cast<bool>((x >> pos) & 1);
}

// ---------------------------------------------------------------- AES-128 (FIPS-197), tables built at compile time
// Little-endian words: the int4's x, y, z, w are the four state columns, byte 0 of the block in the low byte of x
// (the byte order of aes128_mmo.cuh:80-84, which encrypts the int4's bytes in memory order).  kAes is the source the
// LDS copies are filled from (WaveTables) and what aes_detail::ComputeTe0 / Sbox return.
struct AesTables {
  uint32_t te0[256];  // column (2, 1, 1, 3) * S[i], row 0 in the low byte
  uint8_t sbox[256];
};
constexpr uint8_t Times2(uint8_t v) { return static_cast<uint8_t>((v << 1) ^ ((v & 0x80) ? 0x1b : 0)); }
constexpr uint8_t Rotl8(uint8_t v, int n) { return static_cast<uint8_t>((v << n) | (v >> (8 - n))); }
constexpr AesTables BuildAesTables() {
  AesTables t{};
  // p runs through the powers of the generator 3 and q through their inverses; S[p] = affine(q)
  uint8_t p = 1, q = 1;
  do {
    p = static_cast<uint8_t>(p ^ Times2(p));
    q = static_cast<uint8_t>(q ^ (q << 1));
    q = static_cast<uint8_t>(q ^ (q << 2));
    q = static_cast<uint8_t>(q ^ (q << 4));
    if (q & 0x80) q = static_cast<uint8_t>(q ^ 0x09);
    t.sbox[p] = static_cast<uint8_t>(q ^ Rotl8(q, 1) ^ Rotl8(q, 2) ^ Rotl8(q, 3) ^ Rotl8(q, 4) ^ 0x63);
  } while (p != 1);
  t.sbox[0] = 0x63;
  for (int i = 0; i < 256; ++i) {
    const uint32_t s = t.sbox[i], s2 = Times2(t.sbox[i]), s3 = s2 ^ s;
    t.te0[i] = s2 | s << 8 | s << 16 | s3 << 24;
  }
  return t;
}
inline constexpr AesTables kAes = BuildAesTables();
static_assert(kAes.sbox[0x53] == 0xed && kAes.te0[0] == 0xa56363c6u);  // FIPS-197 fig. 7; Te0[0] = c66363a5 big-endian

__device__ inline uint32_t RotL(uint32_t v, int n) { return __builtin_rotateleft32(v, n); }

// ---------------------------------------------------------------- AES-128 on tables in LDS, one block per thread
// The per-thread form of the cipher for calls from inside a caller's kernel (fss::prg::Aes128Soft, and the device
// side of fss::prg::Aes128Mmo): the round table Te0 (1 KiB) and the S-box (256 B) are read from wherever the caller
// keeps them -- LDS in every case that matters (include/fss/prg/aes128_mmo_soft.cuh:197-207 asks for __shared__
// tables; Aes128Mmo keeps a bank-replicated copy per workgroup, WaveTables below) -- and the round keys are wave-uniform values
// (scalar registers), never per-thread byte arrays.
//
// BE = the word order of the state and of the table.  The reference's software AES works on big-endian words
// (LoadBE32, aes128_mmo_soft.cuh:113-129) with Te0[i] = 2S << 24 | S << 16 | S << 8 | 3S (ComputeTe0, :55-60);
// this header's own tables (kAes) are the little-endian mirror image.  The cipher is the same: output byte k of
// column j is byte k of column j + k run through the table and rotated by k bytes -- to the right for big-endian words,
// to the left for little-endian ones.
template <bool BE>
__host__ __device__ constexpr int AesBytePos(int k) { return BE ? 24 - 8 * k : 8 * k; }
template <bool BE>
__host__ __device__ inline uint32_t AesRot(uint32_t v, int k) {  // the table entry for row k
  if (k == 0) return v;
  return BE ? (v >> (8 * k) | v << (32 - 8 * k)) : (v << (8 * k) | v >> (32 - 8 * k));
}
__host__ __device__ inline uint32_t ByteSwap(uint32_t v) { return __builtin_bswap32(v); }
// Three-input logic in one instruction on the device (v_bitop3_b32; the compiler's own selection leaves chains of
// two-input v_xor: 200 per block in round 4's per-thread cipher, profiles/r5/README.md).
__host__ __device__ inline uint32_t Xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}
__host__ __device__ inline uint32_t AndXor(uint32_t m, uint32_t d, uint32_t x) {  // (m & d) ^ x
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(m, d, x, 0x6a);
#else
  return (m & d) ^ x;
#endif
}
// x ^ y ^ (round-key word j of round r): a key type may fold its own structure in (AesKeySelect: one more bitop3 for the
// per-lane choice instead of a select and a move); the default is one three-input XOR.
template <typename Key>
__host__ __device__ inline uint32_t FoldKey(const Key& key, int r, int j, uint32_t x, uint32_t y) {
  if constexpr (requires { key.Fold(r, j, x, y); }) return key.Fold(r, j, x, y);
  else return Xor3(x, y, key.W(r, j));
}

// How a table is addressed: a plain pointer (host code, or tables somewhere else than LDS), or a 32-bit LDS address
// (the form the device path converts the caller's pointers to: ds_read_b32 / ds_read_u8 instead of flat loads).
struct PlainTables {
  static constexpr bool kUnrollRounds = false;  // the general path: compact code (one rolled round), any address space
  const uint32_t* te0;
  const uint8_t* sbox;
  __host__ __device__ uint32_t T(uint32_t i) const { return te0[i]; }
  __host__ __device__ uint32_t S(uint32_t i) const { return sbox[i]; }
};
// The round table with every entry repeated once per LDS bank: word 32 i + b holds Te0[i] for b = 0 .. 31 (32 KiB).  Lane
// l reads copy l % 32, so the 64 lookups of a wave instruction never share a bank (the layout of the engine's own
// tables, fss_amd/csrc/fss_dev.hpp, with one table and the rotations left to the VALU).
struct BankedPlainTables {  // host code / tables outside LDS: copy 0
  static constexpr bool kUnrollRounds = false;
  const uint32_t* te0;
  const uint8_t* sbox;
  __host__ __device__ uint32_t T(uint32_t i) const { return te0[32 * i]; }
  __host__ __device__ uint32_t S(uint32_t i) const { return sbox[i]; }
};
#if defined(__HIP_DEVICE_COMPILE__)
// Addresses into a bank-replicated table (128-byte rows) for the four bytes of a state word in six instructions
// instead of eight (v_bfe + v_lshl_add each): the word is split once into its even and its odd bytes, one per 16-bit
// half (v_and, v_perm), and every lookup address is then ONE v_mad_u32_u16 -- half x 128 + this lane's base -- with
// op_sel picking the half (the compiler has no pattern for it; 80 of a block's 580 VALU instructions, and the VALU is
// what bounds a per-thread walk).  The row size and the v_perm selector sit in vector registers: a walk's scalar
// registers hold round keys, and two more scalar constants made the compiler park keys in VGPR lanes.
struct BankedRows {
  const __attribute__((address_space(3))) uint32_t* te0_lane;  // table base + this lane's bank
  uint32_t row_bytes, odd_sel;
  __device__ explicit BankedRows(const __attribute__((address_space(3))) uint32_t* lane_base) : te0_lane(lane_base) {
    asm("v_mov_b32 %0, 0x80" : "=v"(row_bytes));
    asm("v_mov_b32 %0, 0x0c030c01" : "=v"(odd_sel));
  }
  struct Halves {
    uint32_t even, odd;  // bytes (2, 0) and (3, 1) of the word, zero-extended to 16 bits each
  };
  __device__ Halves Split(uint32_t w) const { return Halves{w & 0x00ff00ffu, __builtin_amdgcn_perm(0u, w, odd_sel)}; }
  template <int pos>  // pos: the byte's bit position in the word, 0 / 8 / 16 / 24
  __device__ uint32_t TAt(const Halves& h) const {
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(te0_lane)), src = (pos & 8) ? h.odd : h.even;
    uint32_t addr;
    if constexpr (pos & 16) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(addr) : "v"(src), "v"(row_bytes), "v"(base));
    else asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(addr) : "v"(src), "v"(row_bytes), "v"(base));
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(static_cast<uintptr_t>(addr));
  }
  __device__ uint32_t T(uint32_t i) const { return te0_lane[32 * i]; }
};
struct BankedLdsTables : BankedRows {
  static constexpr bool kUnrollRounds = true;
  const __attribute__((address_space(3))) uint8_t* sbox;
  __device__ BankedLdsTables(const __attribute__((address_space(3))) uint32_t* lane_base, const __attribute__((address_space(3))) uint8_t* sbox_)
      : BankedRows(lane_base), sbox(sbox_) {}
  __device__ uint32_t S(uint32_t i) const { return sbox[i]; }
};
// ... the same layout with no separate S-box: S(x) is byte 1 of the little-endian round-table entry (2S, S, S, 3S), read
// from this lane's own bank as well (a 256-byte S-box would put 64 random byte reads on 64 words of 32 banks).
struct BankedLdsRoundTable : BankedRows {
  static constexpr bool kUnrollRounds = true;
  static constexpr bool kSboxInTable = true;
  using BankedRows::BankedRows;
  __device__ uint32_t S(uint32_t i) const { return te0_lane[32 * i] >> 8 & 0xff; }
};
struct LdsTables {
  static constexpr bool kUnrollRounds = true;  // the fast path: nine straight-line rounds, keys by constant index
  const __attribute__((address_space(3))) uint32_t* te0;
  const __attribute__((address_space(3))) uint8_t* sbox;
  __device__ uint32_t T(uint32_t i) const { return te0[i]; }
  __device__ uint32_t S(uint32_t i) const { return sbox[i]; }
};
#endif

// The sixteen table entries of a round: t[j][k] = T(byte k of state word (j + k) % 4) (ShiftRows in the indices).
// (tables that hold the entry already rotated for its row -- the wide four-table layout, WideTab -- say kFourTables: the
// lookup for row k goes to table k and the round applies no rotation)
template <bool BE, typename Tab>
__host__ __device__ inline uint32_t AesRowRot(uint32_t v, int k) {
  if constexpr (requires { Tab::kFourTables; }) return v;
  else return AesRot<BE>(v, k);
}
template <bool BE, typename Tab>
__host__ __device__ inline void AesLookups(const Tab& tab, const uint32_t c[4], uint32_t t[4][4]) {
  if constexpr (requires { Tab::kFourTables; }) {
    static_assert(!BE, "the four-table layout is little-endian");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[j][0] = tab.template Lk<0, 0>(c[j]);
      t[j][1] = tab.template Lk<1, 1>(c[(j + 1) & 3]);
      t[j][2] = tab.template Lk<2, 2>(c[(j + 2) & 3]);
      t[j][3] = tab.template Lk<3, 3>(c[(j + 3) & 3]);
    }
  } else if constexpr (requires { tab.Split(0u); }) {
    const typename Tab::Halves h[4] = {tab.Split(c[0]), tab.Split(c[1]), tab.Split(c[2]), tab.Split(c[3])};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[j][0] = tab.template TAt<AesBytePos<BE>(0)>(h[j]);
      t[j][1] = tab.template TAt<AesBytePos<BE>(1)>(h[(j + 1) & 3]);
      t[j][2] = tab.template TAt<AesBytePos<BE>(2)>(h[(j + 2) & 3]);
      t[j][3] = tab.template TAt<AesBytePos<BE>(3)>(h[(j + 3) & 3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) t[j][k] = tab.T(c[(j + k) & 3] >> AesBytePos<BE>(k) & 0xff);
  }
}

// The last round: SubBytes, ShiftRows and the key, no MixColumns.  BankedLdsRoundTable has no S-box of its
// own: S(x) is byte 1 of the entry, and two v_perm put the four of them in their places (three instructions a word
// instead of four extractions and three merges).
template <bool BE, typename Tab, typename Key>
__host__ __device__ inline void AesLastRound(const Tab& tab, const Key& key, const uint32_t c[4], uint32_t out[4]) {
  if constexpr (requires { Tab::kFourTables; }) {
    // T2 has S in byte 0, T3 in byte 1, T0 in byte 2, T1 in byte 3: every lookup holds its S-box byte in place
#pragma unroll
    for (int j = 0; j < 4; ++j)
      out[j] = FoldKey(key, 10, j, tab.LastColumn(tab.template Lk<0, 2>(c[j]), tab.template Lk<1, 3>(c[(j + 1) & 3]),
                                                   tab.template Lk<2, 0>(c[(j + 2) & 3]), tab.template Lk<3, 1>(c[(j + 3) & 3])), 0u);
  } else if constexpr (requires { Tab::kSboxInTable; }) {
    uint32_t t[4][4];
    AesLookups<BE>(tab, c, t);
    // byte 1 of t[j][k] goes to byte AesBytePos(k) / 8 of the word: selector 1 = byte 1 of the second operand, 5 = byte 1
    // of the first, 0x0c = zero
    constexpr auto sel = [](int ka, int kb) {
      uint32_t v = 0x0c0c0c0cu;
      v = (v & ~(0xffu << AesBytePos<BE>(ka))) | 0x01u << AesBytePos<BE>(ka);
      v = (v & ~(0xffu << AesBytePos<BE>(kb))) | 0x05u << AesBytePos<BE>(kb);
      return v;
    };
#pragma unroll
    for (int j = 0; j < 4; ++j)
      out[j] = FoldKey(key, 10, j, __builtin_amdgcn_perm(t[j][1], t[j][0], sel(0, 1)), __builtin_amdgcn_perm(t[j][3], t[j][2], sel(2, 3)));
  } else {
    uint32_t u[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) u[j][k] = tab.S(c[(j + k) & 3] >> AesBytePos<BE>(k) & 0xff);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      out[j] = FoldKey(key, 10, j, u[j][0] << AesBytePos<BE>(0) | u[j][1] << AesBytePos<BE>(1),
                       u[j][2] << AesBytePos<BE>(2) | u[j][3] << AesBytePos<BE>(3));
  }
}

// One block: out = AES_k(in).  `Key` supplies the round-key words: key.W(r, j) = word j of round r's key, in the word
// order of the state.  The nine table rounds are unrolled (the reference's loop is `#pragma unroll 1`,
// aes128_mmo_soft.cuh:131, and reloads its key bytes from the stack every round).
template <bool BE, typename Tab, typename Key>
__host__ __device__ inline void AesEncrypt(const Tab& tab, const Key& key, const uint32_t in[4], uint32_t out[4]) {
  uint32_t c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = in[j] ^ key.W(0, j);
  auto round = [&](int r) {
    uint32_t t[4][4];
    AesLookups<BE>(tab, c, t);  // 16 lookups in flight
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c[j] = FoldKey(key, r, j, Xor3(t[j][0], AesRowRot<BE, Tab>(t[j][1], 1), AesRowRot<BE, Tab>(t[j][2], 2)), AesRowRot<BE, Tab>(t[j][3], 3));
  };
  if constexpr (Tab::kUnrollRounds) {
#pragma unroll
    for (int r = 1; r < 10; ++r) round(r);
  } else {
#pragma unroll 1
    for (int r = 1; r < 10; ++r) round(r);
  }
  AesLastRound<BE>(tab, key, c, out);
}

// Two blocks of the same plaintext under two key schedules, round by round side by side: 32 lookups in flight per lane
// instead of 16, one block's table latency under the other's arithmetic (the engine's aes_mmo2, fss_amd/csrc/fss_dev.hpp).
// What a DCF level needs: the seed and the value block of the child it takes.
template <bool BE, typename Tab, typename KeyA, typename KeyB>
__host__ __device__ inline void AesEncrypt2(const Tab& tab, const KeyA& ka, const KeyB& kb, const uint32_t in[4], uint32_t oa[4],
                                            uint32_t ob[4]) {
  uint32_t a[4], b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = in[j] ^ ka.W(0, j); b[j] = in[j] ^ kb.W(0, j); }
  auto round = [&](int r) {
    uint32_t ta[4][4], tb[4][4];
    AesLookups<BE>(tab, a, ta);
    AesLookups<BE>(tab, b, tb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = FoldKey(ka, r, j, Xor3(ta[j][0], AesRowRot<BE, Tab>(ta[j][1], 1), AesRowRot<BE, Tab>(ta[j][2], 2)), AesRowRot<BE, Tab>(ta[j][3], 3));
      b[j] = FoldKey(kb, r, j, Xor3(tb[j][0], AesRowRot<BE, Tab>(tb[j][1], 1), AesRowRot<BE, Tab>(tb[j][2], 2)), AesRowRot<BE, Tab>(tb[j][3], 3));
    }
  };
  if constexpr (Tab::kUnrollRounds) {
#pragma unroll
    for (int r = 1; r < 10; ++r) round(r);
  } else {
#pragma unroll 1
    for (int r = 1; r < 10; ++r) round(r);
  }
  AesLastRound<BE>(tab, ka, a, oa);  // (its lookups are independent of the other block's: the scheduler interleaves them)
  AesLastRound<BE>(tab, kb, b, ob);
}

// Round keys as 44 words (one wave-uniform schedule), and the per-lane choice between two schedules A and B given as
// A and D = A ^ B:  k = A ^ (mask & D), mask = 0 or ~0 per lane -- the taken child of a tree level without computing
// the other one (the engine's aes_mmo<SEL>, fss_amd/csrc/fss_dev.hpp).
struct AesKeyWords {
  const uint32_t* rk;
  __host__ __device__ uint32_t W(int r, int j) const { return rk[4 * r + j]; }
};
struct AesKeySelect {
  const uint32_t* a;
  const uint32_t* d;
  uint32_t mask;
  __host__ __device__ uint32_t W(int r, int j) const { return a[4 * r + j] ^ (mask & d[4 * r + j]); }
  __host__ __device__ uint32_t Fold(int r, int j, uint32_t x, uint32_t y) const {
    return AndXor(mask, d[4 * r + j], Xor3(x, y, a[4 * r + j]));
  }
};

// Key schedule in state word order (FIPS-197 5.2) from the key as four little-endian words, S-box lookups through `S`.
// Straight-line code (every index a constant): the 44 words stay in registers.
template <bool BE, typename SboxFn>
__host__ __device__ inline void AesExpandKeyFromWords(const uint32_t kw[4], uint32_t rk[44], SboxFn S) {
  uint32_t w[44];  // little-endian words while expanding
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = kw[i];
  uint32_t rcon = 1;
#pragma unroll
  for (int i = 4; i < 44; ++i) {
    uint32_t t = w[i - 1];
    if (i % 4 == 0) {
      t = t >> 8 | t << 24;
      t = S(t & 0xff) | S(t >> 8 & 0xff) << 8 | S(t >> 16 & 0xff) << 16 | S(t >> 24) << 24;
      t ^= rcon;
      rcon = Times2(static_cast<uint8_t>(rcon));
    }
    w[i] = w[i - 4] ^ t;
  }
#pragma unroll
  for (int i = 0; i < 44; ++i) rk[i] = BE ? ByteSwap(w[i]) : w[i];
}
// ... from 16 key bytes
template <bool BE, typename SboxFn>
__host__ __device__ inline void AesExpandKeyWords(const unsigned char key[16], uint32_t rk[44], SboxFn S) {
  uint32_t kw[4];
  for (int i = 0; i < 4; ++i)
    kw[i] = static_cast<uint32_t>(key[4 * i]) | static_cast<uint32_t>(key[4 * i + 1]) << 8 |
            static_cast<uint32_t>(key[4 * i + 2]) << 16 | static_cast<uint32_t>(key[4 * i + 3]) << 24;
  AesExpandKeyFromWords<BE>(kw, rk, S);
}

#if defined(__HIP_DEVICE_COMPILE__)
// Linear index of the calling thread in its workgroup, and the wave it belongs to.
__device__ inline unsigned LinearTid() { return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); }

// The lanes of a wave hand LDS contents to each other: order the wave's own LDS accesses around the hand-over.
__device__ inline void WaveLdsSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A value every active lane holds identically, moved to a scalar register.
__device__ inline uint32_t Uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// Do all active lanes hold the same n words?
__device__ inline bool SameInAllLanes(const uint32_t* w, int n) {
  uint32_t diff = 0;
  for (int i = 0; i < n; ++i) diff |= w[i] ^ Uniform(w[i]);
  return __ballot(diff != 0) == 0;
}

// 64 bits that tell this launch (and this workgroup) from earlier ones: LDS is not cleared between kernels, so a marker
// left behind by an earlier kernel must not pass for this launch's.  The queue's packet counter (dispatch id: monotonic
// per queue, never reused) is folded with the AQL packet's address and its kernel-argument address (which tell queues
// apart; by themselves they repeat -- the ring wraps, argument chunks are recycled) and with the workgroup id.  The
// guarantee is probabilistic: two launches on different queues with equal counters would have to collide in the other
// 64 bits as well.
// The dispatch id: a clang builtin where the toolchain has one; on the toolchains this header was written against (ROCm 7.x,
// clang 19 - 22: no builtin) the LLVM intrinsic reached through an asm-labelled declaration -- which works because clang
// maps the label onto the intrinsic's name, not because anything documents it; any other compiler takes the fallback: the
// packet's completion-signal handle (a fresh value per in-flight packet of a queue) in the counter's place.  Define
// FSS_AMD_NO_DISPATCH_ID_INTRINSIC to force the fallback.
#if __has_builtin(__builtin_amdgcn_dispatch_id)
__device__ inline uint64_t DispatchCounter(const uint64_t*) { return __builtin_amdgcn_dispatch_id(); }
#elif defined(__clang__) && __clang_major__ >= 17 && __clang_major__ <= 22 && !defined(FSS_AMD_NO_DISPATCH_ID_INTRINSIC)
extern "C" __device__ uint64_t fss_amd_llvm_dispatch_id(void) __asm("llvm.amdgcn.dispatch.id");
__device__ inline uint64_t DispatchCounter(const uint64_t*) { return fss_amd_llvm_dispatch_id(); }
#else
__device__ inline uint64_t DispatchCounter(const uint64_t* pkt) { return pkt[7]; }  // hsa_kernel_dispatch_packet_t::completion_signal
#endif
__device__ inline void LaunchNonce(uint32_t n[2]) {
  const uint64_t* pkt = reinterpret_cast<const uint64_t*>(__builtin_amdgcn_dispatch_ptr());
  const uint64_t wg = blockIdx.x + gridDim.x * (blockIdx.y + static_cast<uint64_t>(gridDim.y) * blockIdx.z);
  uint64_t v = DispatchCounter(pkt) * 0x9e3779b97f4a7c15ull;
  v ^= reinterpret_cast<uint64_t>(pkt) * 0xc2b2ae3d27d4eb4full ^ pkt[5];  // kernarg_address
  v = (v ^ (v >> 29)) * 0xbf58476d1ce4e5b9ull + wg;
  n[0] = Uniform(static_cast<uint32_t>(v));
  n[1] = Uniform(static_cast<uint32_t>(v >> 32));
}

// fss::prg::Aes128Mmo on the device has no table arguments (the reference's class is host-only, aes128_mmo.cuh:75-77):
// the workgroup keeps ONE bank-replicated copy of the round table in LDS (32 KiB: entry i at words 32 i .. 32 i + 31, lane
// l reads copy l % 32 -- no bank conflicts whatever the 64 indices of a wave instruction are; a plain 1 KiB copy per
// wave, round 4's form, served them in 4-5 turns), and the S-box is byte 1 of the same entries.  No workgroup barrier
// may be used to fill it -- a kernel may call Gen from divergent code, after some of its threads have returned -- so
// EVERY wave writes the whole table itself before its first use (32 16-byte stores per lane; all waves write the same
// values, so a wave only ever needs its OWN stores to have landed, which a wave-level sync gives it; whatever another
// wave stores at any time is what is there already).  `ready` is the caller's per-thread flag: a thread's first call
// always fills (nothing is assumed about what the LDS held before).  Every wave also has a 16-byte marker, written after
// its fill and bound to this launch, and every call checks it: a PRG object that reaches a wave through memory with
// its flag already set by another wave (a __shared__ or global scheme, a pointer handed between waves -- the object is
// meant to be a per-thread value) still finds this wave's marker missing and fills.
struct WorkgroupAesTable {
  uint32_t te0[256 * 32];
  uint32_t marker[16][4];
};
// BE: the table in the reference's big-endian word order (Te0[i] = 2S << 24 | S << 16 | S << 8 | 3S, aes128_mmo_soft.cuh:55-60)
// for fss::prg::Aes128Soft, whose schedules and state words are big-endian; a copy of its own (a kernel that uses both PRG
// types on the device holds both, 64 KiB).  S(x) is byte 1 of an entry in either order.
template <bool BE>
__device__ inline BankedLdsRoundTable WaveTablesT(bool& ready) {
  __shared__ __attribute__((aligned(16))) WorkgroupAesTable tab;
  uint32_t* mark = tab.marker[(LinearTid() >> 6) & 15];
  uint32_t nonce[2];
  LaunchNonce(nonce);
  const uint32_t want[4] = {BE ? 0x66737362u : 0x66737361u, 0x6d645f74u, nonce[0], nonce[1]};
  const bool missing = !ready || ((mark[0] ^ want[0]) | (mark[1] ^ want[1]) | (mark[2] ^ want[2]) | (mark[3] ^ want[3])) != 0;
  if (__ballot(missing) != 0) {
    const uint64_t act = __ballot(1);
    const int cnt = __popcll(act);
    const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(act >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(act), 0u));
    uint4* rows = reinterpret_cast<uint4*>(tab.te0);  // 2048 groups of four copies
    for (int i = rank; i < 2048; i += cnt) {
      const uint32_t v = BE ? ByteSwap(kAes.te0[i >> 3]) : kAes.te0[i >> 3];
      rows[i] = make_uint4(v, v, v, v);
    }
    WaveLdsSync();  // the table first, then the marker that says this wave has written it
    mark[0] = want[0]; mark[1] = want[1]; mark[2] = want[2]; mark[3] = want[3];  // every active lane, the same values
    WaveLdsSync();
  }
  ready = true;
  return BankedLdsRoundTable((const __attribute__((address_space(3))) uint32_t*)(tab.te0) + (__lane_id() & 31));
}
__device__ inline BankedLdsRoundTable WaveTables(bool& ready) { return WaveTablesT<false>(ready); }

// The LDS home of a two- or four-key PRG's round keys (DPF: 2 x 44 words, DCF: 4 x 44).  176 wave-uniform words do not
// fit the scalar register file next to a walk's own state -- the compiler would park them in VGPR lanes and from there
// in scratch (1 KiB per lane, round 3) -- and 88 fill it: the compiler reloads or spills some every level, and the
// per-lane choice between two scalar schedules costs an instruction per key word.  So every wave keeps ONE copy in LDS
// and the cipher reads the words a round needs from there: every lane reads the same address (Gen) or one of two (the
// taken child's schedule, chosen in the address), which the LDS serves as broadcasts without bank conflicts.  The slot
// is written when a thread constructs its PRG object on the device, or at the thread's first call for an object that
// came from the host (`slot_ready` is the object's per-thread flag: nothing is assumed about what the LDS held before);
// it carries a tag derived from the key bytes and this launch, checked by every call: a call that finds another tag
// (the object reached this wave through memory with its flag set by another wave, or two key sets alternate in one
// wave) re-derives the schedules from the object's key bytes and rewrites the slot.
template <int mul>
struct WaveKeySlot {
  uint32_t tag[4];
  uint32_t rk[mul * 44];  // the schedules, key i at words 44 i .. 44 i + 43, state word order
};
template <int mul>
__device__ inline WaveKeySlot<mul>* MyKeySlot() {
  __shared__ __attribute__((aligned(16))) WaveKeySlot<mul> slots[16];
  return &slots[(LinearTid() >> 6) & 15];
}
using LdsWords = const __attribute__((address_space(3))) uint32_t*;
struct LdsKeyWords {
  LdsWords rk;
  __device__ uint32_t W(int r, int j) const { return rk[4 * r + j]; }
};
// A per-lane choice between two schedules of the slot, made in the ADDRESS: each lane reads its own schedule (two
// addresses per wave instruction, 88 words = 24 banks apart (DPF: 44 = 12): two broadcasts, no conflict) -- one LDS read per key word
// (round 4 read both schedules and selected: two reads and a v_bfi per word).
struct LdsKeyPerLane {
  LdsWords rk;  // a or b, per lane
  __device__ uint32_t W(int r, int j) const { return rk[4 * r + j]; }
};
#endif

// The round keys of a PRG object with `mul` AES keys, as the per-thread cipher wants them.
//  * Layout: schedules that a tree level chooses between per lane are stored as A and D = A ^ B, so the taken child
//    costs one block (AesKeySelect) and the other child's key is A ^ D:
//      mul = 1: [k0]      mul = 2: [k0, k0 ^ k1]      mul = 4 (DCF, outputs sl, vl, sr, vr): [k0, k1, k0 ^ k2, k1 ^ k3]
//  * On the device a single schedule (mul = 1) is wave-uniform values in scalar registers, two or four live in the wave's
//    LDS slot (WaveKeySlot) in plain form -- when every lane of the wave was constructed from the same keys (`uniform`;
//    the normal case: one key set per kernel).  Lanes with keys of their own fall back to expanding their key per call.
//  * lds2: two schedules in the LDS slot as well (four always are).  What pays with the bank-replicated table, whose
//    lookups leave the LDS pipe half idle; the reference-shaped Aes128Soft, bound by its 1 KiB table's bank conflicts,
//    keeps its two schedules in scalar registers (1.93 G evals/s against 1.77 with them in LDS).
template <int mul, bool BE, bool lds2 = false>
struct AesSchedules {
  uint32_t keyw[mul][4];  // the raw keys, little-endian words (per lane)
  uint32_t rk[mul][44];   // A / D layout above, state word order
  bool uniform;
  mutable bool slot_ready = false;  // mul >= 2 on the device: this thread has seen its wave's LDS copy of the schedules filled
  static constexpr bool kSlotted = mul == 4 || (mul == 2 && lds2);  // (device) the schedules live in the wave's LDS slot

  template <typename SboxFn>
  __host__ __device__ void Set(const unsigned char* const* keys, SboxFn S) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (kSlotted) {
      // two or four schedules: they live in the wave's LDS slot (WaveKeySlot), never in this object's registers
      for (int i = 0; i < mul; ++i)
        for (int j = 0; j < 4; ++j)
          keyw[i][j] = static_cast<uint32_t>(keys[i][4 * j]) | static_cast<uint32_t>(keys[i][4 * j + 1]) << 8 |
                       static_cast<uint32_t>(keys[i][4 * j + 2]) << 16 | static_cast<uint32_t>(keys[i][4 * j + 3]) << 24;
      uniform = SameInAllLanes(&keyw[0][0], 4 * mul);
      if (uniform) {
        FillSlot(S);
        slot_ready = true;
      }
      return;
    }
#endif
    uint32_t full[mul][44];
    for (int i = 0; i < mul; ++i) {
      for (int j = 0; j < 4; ++j)
        keyw[i][j] = static_cast<uint32_t>(keys[i][4 * j]) | static_cast<uint32_t>(keys[i][4 * j + 1]) << 8 |
                     static_cast<uint32_t>(keys[i][4 * j + 2]) << 16 | static_cast<uint32_t>(keys[i][4 * j + 3]) << 24;
      AesExpandKeyWords<BE>(keys[i], full[i], S);
    }
    for (int w = 0; w < 44; ++w) {
      rk[0][w] = full[0][w];
      if constexpr (mul == 2) rk[1][w] = full[0][w] ^ full[1][w];
      if constexpr (mul == 4) {
        rk[1][w] = full[1][w];
        rk[2][w] = full[0][w] ^ full[2][w];
        rk[3][w] = full[1][w] ^ full[3][w];
      }
    }
    uniform = true;
#if defined(__HIP_DEVICE_COMPILE__)
    uniform = SameInAllLanes(&keyw[0][0], 4 * mul);
    for (int i = 0; i < mul; ++i)
      for (int w = 0; w < 44; ++w) rk[i][w] = Uniform(rk[i][w]);  // scalar registers (lane 0's when not uniform: unused then)
#endif
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // ---- mul >= 2 on the device: the wave's LDS copy of the schedules
  __device__ void Tag(uint32_t t[4]) const {  // 128 bits of the key bytes (and the word order the slot is written in)
    uint32_t nonce[2];
    LaunchNonce(nonce);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      t[j] = keyw[0][j] ^ RotL(keyw[mul > 1 ? 1 : 0][j], 7) ^ RotL(keyw[mul > 2 ? 2 : 0][j], 13) ^ RotL(keyw[mul > 3 ? 3 : 0][j], 19) ^
             (BE ? 0x9e3779b9u : 0u) ^ nonce[j & 1];
  }
  // Expand the keys one after the other (44 words live at a time) into this wave's slot; every active lane
  // writes the same values.  S: the S-box the expansion looks up through.
  template <typename SboxFn>
  __device__ void FillSlot(SboxFn S) const {
    WaveKeySlot<mul>* s = MyKeySlot<mul>();
    WaveLdsSync();  // the wave's earlier reads of the slot are done
#pragma unroll 1  // one expansion at a time: 44 words live, not 176
    for (int i = 0; i < mul; ++i) {
      uint32_t kw[4], w[44];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // keyw[i] by selects, not by a runtime index (that would put the array on the stack)
        kw[j] = keyw[0][j];
#pragma unroll
        for (int q = 1; q < mul; ++q) kw[j] = i == q ? keyw[q][j] : kw[j];
      }
      AesExpandKeyFromWords<BE>(kw, w, S);
#pragma unroll
      for (int q = 0; q < 44; ++q) s->rk[44 * i + q] = w[q];
    }
    uint32_t t[4];
    Tag(t);
#pragma unroll
    for (int j = 0; j < 4; ++j) s->tag[j] = t[j];
    WaveLdsSync();
  }
  // This wave's copy of the schedules, re-derived from the key bytes first if the slot holds something else.  nullptr
  // when the lanes active NOW do not all hold the same key set -- `uniform` was decided when the objects were built
  // (always true on the host), but lanes may since have picked between two objects (`odd ? sch_b : sch_a`), or objects
  // built in divergent branches may meet after reconvergence: one slot cannot serve two key sets at once, and the
  // callers then take the per-lane expansion.
  template <typename Tab>
  __device__ LdsWords SlotKeys(const Tab& tab) const {
    WaveKeySlot<mul>* s = MyKeySlot<mul>();
    uint32_t t[4];
    Tag(t);
    if (!SameInAllLanes(t, 4)) return nullptr;
    const bool stale = !slot_ready || ((s->tag[0] ^ t[0]) | (s->tag[1] ^ t[1]) | (s->tag[2] ^ t[2]) | (s->tag[3] ^ t[3])) != 0;
    if (__ballot(stale) != 0) FillSlot([&tab](uint32_t b) { return tab.S(b); });
    slot_ready = true;
    return (LdsWords)(s->rk);
  }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
  // The key slot checked (and refilled if stale) once, for a walk that then passes it to every Gen / GenSide (mul >= 2 with
  // wave-uniform keys; nullptr otherwise: nothing to keep).
  template <typename Tab>
  __device__ const void* ReadyKeys(const Tab& tab) const {
    if constexpr (kSlotted) {
      if (uniform) return (const void*)(uintptr_t)SlotKeys(tab);
    }
    return nullptr;
  }
#endif
  __host__ __device__ void KeyBytes(int i, unsigned char out[16]) const {
    for (int j = 0; j < 16; ++j) out[j] = static_cast<unsigned char>(keyw[i][j >> 2] >> (8 * (j & 3)));
  }

  struct KeyXor {  // A ^ D
    const uint32_t* a;
    const uint32_t* d;
    __host__ __device__ uint32_t W(int r, int j) const { return a[4 * r + j] ^ d[4 * r + j]; }
  };
  static __host__ __device__ void Words(int4 v, uint32_t w[4]) {
    w[0] = static_cast<uint32_t>(v.x); w[1] = static_cast<uint32_t>(v.y); w[2] = static_cast<uint32_t>(v.z); w[3] = static_cast<uint32_t>(v.w);
    if constexpr (BE) for (int j = 0; j < 4; ++j) w[j] = ByteSwap(w[j]);
  }
  static __host__ __device__ int4 Mmo(const uint32_t o[4], int4 seed) {  // ciphertext ^ plaintext (aes128_mmo.cuh:79-89)
    uint32_t c[4] = {o[0], o[1], o[2], o[3]};
    if constexpr (BE) for (int j = 0; j < 4; ++j) c[j] = ByteSwap(c[j]);
    return int4{static_cast<int>(c[0]) ^ seed.x, static_cast<int>(c[1]) ^ seed.y, static_cast<int>(c[2]) ^ seed.z,
                static_cast<int>(c[3]) ^ seed.w};
  }

  // out[i] = AES_{key i}(seed) ^ seed for every key.  ready (device, mul = 4): the wave's key slot as SlotKeys returned
  // it earlier in the same walk (PreparedAes below) -- the slot's tag is then not checked again.
  template <typename Tab>
  __host__ __device__ __attribute__((always_inline)) std::array<int4, mul> Gen(const Tab& tab, int4 seed, const void* ready = nullptr) const {
    std::array<int4, mul> out;
    uint32_t in[4], o[4];
    Words(seed, in);
    (void)ready;
    bool shared_keys = uniform;  // one key set for the whole wave (else: every lane expands its own)
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (kSlotted) {
      const LdsWords k = !uniform ? nullptr : (ready ? (LdsWords)(uintptr_t)ready : SlotKeys(tab));
      if (k) {
#pragma unroll
        for (int i = 0; i < mul; i += 2) {
          uint32_t o2[4];
          AesEncrypt2<BE>(tab, LdsKeyWords{k + 44 * i}, LdsKeyWords{k + 44 * (i + 1)}, in, o, o2);
          out[i] = Mmo(o, seed);
          out[i + 1] = Mmo(o2, seed);
        }
        return out;
      }
      shared_keys = false;  // (slotted objects keep no schedules of their own)
    }
#endif
    if (shared_keys) {
      if constexpr (mul == 2) {  // both children side by side (Gen of a DPF level)
        uint32_t o2[4];
        AesEncrypt2<BE>(tab, AesKeyWords{rk[0]}, KeyXor{rk[0], rk[1]}, in, o, o2);
        out[0] = Mmo(o, seed);
        out[1] = Mmo(o2, seed);
        return out;
      }
#pragma unroll
      for (int i = 0; i < mul; ++i) {
        const bool direct = mul == 1 || i == 0 || (mul == 4 && i == 1);
        if (direct) AesEncrypt<BE>(tab, AesKeyWords{rk[i]}, in, o);
        else AesEncrypt<BE>(tab, KeyXor{rk[mul == 2 ? 0 : i - 2], rk[i]}, in, o);
        out[i] = Mmo(o, seed);
      }
    } else {
#pragma unroll
      for (int i = 0; i < mul; ++i) {
        const Expanded e = PerLane(tab, keyw[i]);
        AesEncrypt<BE>(tab, AesKeyWords{e.w}, in, o);
        out[i] = Mmo(o, seed);
      }
    }
    return out;
  }
  // The outputs of one side only: mul = 2: {right ? out[1] : out[0]};  mul = 4: {right ? out[2] : out[0], right ? out[3] : out[1]}
  template <typename Tab>
  __host__ __device__ __attribute__((always_inline)) std::array<int4, (mul + 1) / 2> GenSide(const Tab& tab, int4 seed, bool right, const void* ready = nullptr) const {
    static_assert(mul == 2 || mul == 4);
    constexpr int half = mul / 2;
    std::array<int4, half> out;
    uint32_t in[4], o[4];
    Words(seed, in);
    (void)ready;
    bool shared_keys = uniform;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (kSlotted) {
      const LdsWords base = !uniform ? nullptr : (ready ? (LdsWords)(uintptr_t)ready : SlotKeys(tab));
      if (base) {
        if constexpr (mul == 4) {
          const LdsWords k = base + (right ? 88 : 0);  // [sl, vl | sr, vr]: this lane's side, keys 0, 1 or 2, 3
          uint32_t o2[4];
          AesEncrypt2<BE>(tab, LdsKeyPerLane{k}, LdsKeyPerLane{k + 44}, in, o, o2);  // seed and value block side by side
          out[0] = Mmo(o, seed);
          out[1] = Mmo(o2, seed);
        } else {
          const LdsWords k = base + (right ? 44 : 0);  // [left | right]: this lane's child
          AesEncrypt<BE>(tab, LdsKeyPerLane{k}, in, o);
          out[0] = Mmo(o, seed);
        }
        return out;
      }
      shared_keys = false;
    }
#endif
    if (shared_keys) {
      const uint32_t mask = right ? ~0u : 0u;
#pragma unroll
      for (int i = 0; i < half; ++i) {
        AesEncrypt<BE>(tab, AesKeySelect{rk[i], rk[half + i], mask}, in, o);
        out[i] = Mmo(o, seed);
      }
    } else {
#pragma unroll
      for (int i = 0; i < half; ++i) {
        uint32_t kw[4];  // this lane's own key for output i of its side (constant indices: no stack array)
#pragma unroll
        for (int j = 0; j < 4; ++j) kw[j] = right ? keyw[half + i][j] : keyw[i][j];
        const Expanded e = PerLane(tab, kw);
        AesEncrypt<BE>(tab, AesKeyWords{e.w}, in, o);
        out[i] = Mmo(o, seed);
      }
    }
    return out;
  }

 private:
  struct Expanded {
    uint32_t w[44];
  };
  template <typename Tab>
  __host__ __device__ static Expanded PerLane(const Tab& tab, const uint32_t kw[4]) {  // a lane's own key, expanded for one call
    Expanded e;
    AesExpandKeyFromWords<BE>(kw, e.w, [&tab](uint32_t b) { return tab.S(b); });
    return e;
  }
};

#if defined(__HIP_DEVICE_COMPILE__)
// A PRG whose per-call checks were made ONCE: the workgroup table filled and this wave's marker seen (WaveTables), the
// key slot's tag seen (mul = 4).  A walk takes one at its start (`prg.Prepared()`, PreparedPrg below) and calls it
// level after level: nothing can invalidate either while the thread stays inside the walk -- LDS belongs to this
// launch, and other waves only ever store the values that are there already.  Same Gen / GenSide as the PRG itself.
template <int mul, typename Ks, typename Tab>
struct PreparedAes {
  const Ks& ks;
  Tab tab;
  const void* keys;
  __device__ std::array<int4, mul> Gen(int4 seed) const { return ks.Gen(tab, seed, keys); }
  __device__ std::array<int4, (mul + 1) / 2> GenSide(int4 seed, bool right) const
    requires(mul >= 2)
  {
    return ks.GenSide(tab, seed, right, keys);
  }
};
// ---------------------------------------------------------------- the engine's table layout for per-thread callers (opt-in)
// fss::prg::Aes128MmoWide (include/fss_amd.hpp): the four-table, rotate-free layout of the engine's own kernels
// (fss_amd/csrc/fss_dev.hpp) behind the per-thread Gen / GenSide.  T_t = Te0 rotated left by 8 t bits, every entry
// repeated once per LDS bank:
//   words [0, 16384):      row i (64 words) = [T0[i] x 32 | T1[i] x 32]
//   words [16384, 32768):  row i            = [T2[i] x 32 | T3[i] x 32]
// so the byte address of a lookup is  region << 16 | index << 8 | half << 7 | (lane & 31) << 2  -- formed by ONE v_perm
// (or one v_bitop3 for byte 1) that drops the state byte into bits 8..15 of a per-lane constant -- no bank conflicts, and
// no rotate instructions in the round (twelve per round in the one-table form above: what bounds that form is VALU
// issue, include/fss_amd_thread.hpp's own counters in profiles/r5/dropin_device_pmc.txt).  The price is what makes it
// opt-in: 128 KiB of LDS, i.e. ONE workgroup per CU, which pays only with 1024-thread workgroups; the table sits at LDS
// address 0 (the byte insertion needs the base's low 17 bits clear) and is filled cooperatively behind a workgroup
// barrier -- the caller's kernel says so explicitly (WideTablesInit, every thread, uniform control flow).
// Round keys: scalar loads straight from where the PRG object lives -- the kernel-argument segment for an object passed
// to the kernel by value -- one round at a time (the load's address carries a dependency on a value of the round, or
// the compiler hoists all 88 / 176 words out of the level loop and spills them).
constexpr int kWideTeWords = 32768;
struct alignas(131072) WideAesTables {
  uint32_t te[kWideTeWords];
};
struct WideAesMarker {
  uint32_t w[4];
};
__device__ inline WideAesTables* WorkgroupWideTables() {
  __shared__ WideAesTables tab;
  return &tab;
}
__device__ inline WideAesMarker* WorkgroupWideMarker() {
  __shared__ WideAesMarker m;
  return &m;
}
// levels walked by each of the workgroup's sixteen waves (PreparedWide::LevelDone), zeroed by WideTablesInit
__device__ inline uint32_t* WorkgroupWideProgress() {
  __shared__ __attribute__((aligned(16))) uint32_t prog[16];
  return prog;
}
__device__ inline void WideMarkerWant(uint32_t want[4]) {
  uint32_t nonce[2];
  LaunchNonce(nonce);
  want[0] = 0x66737377u; want[1] = 0x69646554u; want[2] = nonce[0]; want[3] = nonce[1];
}
// Every thread of the workgroup, once, in uniform control flow, before the first Gen / Eval of a wide PRG.
__device__ inline void WideTablesInit() {
  WideAesTables* t = WorkgroupWideTables();
  const unsigned tid = LinearTid(), n = blockDim.x * blockDim.y * blockDim.z;
  // refused, loudly: a workgroup of another size (128 KiB of LDS per workgroup = one per CU: 256 threads would leave
  // three quarters of the CU's wave slots empty), or a table that the linker did not place at LDS address 0
  if (n != 1024 || static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint32_t*)(t->te))) != 0u)
    __builtin_trap();
  // word tid + 1024 k holds entry ((tid >> 6) + 16 k) & 255 of table ((k >> 4) << 1) | ((tid >> 5) & 1): sixteen distinct
  // entries per lane, requested together
  const int odd = (tid >> 5) & 1;
  uint32_t e[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) e[k] = kAes.te0[((tid >> 6) + 16 * k) & 255];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int tbl = ((k >> 4) << 1) | odd;
    t->te[tid + 1024 * k] = RotL(e[k & 15], (8 * tbl) & 31);
  }
  if (tid == 0) {
    uint32_t want[4];
    WideMarkerWant(want);
    WideAesMarker* m = WorkgroupWideMarker();
    m->w[0] = want[0]; m->w[1] = want[1]; m->w[2] = want[2]; m->w[3] = want[3];
  }
  if (tid < 16) WorkgroupWideProgress()[tid] = 0;
  __syncthreads();
}
// a constant the compiler must keep in a vector register (a literal or scalar operand puts the consuming instruction
// in the slow issue class)
__device__ inline uint32_t InVgpr(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}
// zero in a scalar register that the compiler must treat as depending on `dep`
__device__ inline uint32_t ZeroAfter(uint32_t dep) {
  uint32_t z = 0;
  asm volatile("" : "+s"(z) : "v"(dep));
  return z;
}
struct WideTab {
  uint32_t lo[4];  // per table: region << 16 | half << 7 | (lane & 31) << 2
  uint32_t m_ff00, m_00ff, m_ff0000, m_ffff;
  // Refuses (traps) unless WideTablesInit ran in this launch for this workgroup.  (No per-object "already checked" flag:
  // a PRG object that is written to stops being readable in place from the kernel-argument segment.)
  __device__ static WideTab Make() {
    {
      uint32_t want[4];
      WideMarkerWant(want);
      const WideAesMarker* m = WorkgroupWideMarker();
      if (((m->w[0] ^ want[0]) | (m->w[1] ^ want[1]) | (m->w[2] ^ want[2]) | (m->w[3] ^ want[3])) != 0) __builtin_trap();  // WideTablesInit() was not called
    }
    WideTab L;
    const uint32_t l4 = (__lane_id() & 31u) << 2;
    L.lo[0] = l4; L.lo[1] = l4 | 0x80u; L.lo[2] = l4 | 0x10000u; L.lo[3] = l4 | 0x10080u;
    L.m_ff00 = InVgpr(0xff00u); L.m_00ff = InVgpr(0xffu); L.m_ff0000 = InVgpr(0xff0000u); L.m_ffff = InVgpr(0xffffu);
    return L;
  }
  // address of T_T[byte K of w]
  template <int K, int T>
  __device__ uint32_t Addr(uint32_t w) const {
    if constexpr (K == 1) return __builtin_amdgcn_bitop3_b32(w, m_ff00, lo[T], 0xEA);  // (w & 0xff00) | lo: byte 1 already sits at index << 8
    else return __builtin_amdgcn_perm(w, lo[T], 0x0c020000u | ((4u + K) << 8));       // bytes: lo[0], byte K of w, lo[2], 0
  }
  __device__ static uint32_t At(uint32_t addr) {
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>(static_cast<uintptr_t>(addr));
  }
  template <int K, int T>
  __device__ uint32_t Lk(uint32_t w) const { return At(Addr<K, T>(w)); }
  // ... as a table type of the generic cipher (AesEncrypt / AesEncrypt2 with round keys from the wave's LDS slot: PRG
  // objects constructed per thread inside a kernel, fss::prg::Aes128SoftWide)
  static constexpr bool kUnrollRounds = true;
  static constexpr bool kFourTables = true;
  __device__ uint32_t S(uint32_t i) const { return At(lo[0] | i << 8) >> 8 & 0xff; }  // byte 1 of the T0 entry (2S, S, S, 3S)
  __device__ uint32_t LastColumn(uint32_t a, uint32_t b, uint32_t c, uint32_t d) const {
    return __builtin_amdgcn_bitop3_b32(m_ffff, __builtin_amdgcn_bitop3_b32(m_00ff, a, b, 0xCA), __builtin_amdgcn_bitop3_b32(m_ff0000, c, d, 0xCA), 0xCA);
  }
};
// Round keys of ONE schedule (44 words) or of a per-lane choice between two (per round: A[4] then D[4] = (A ^ B)[4]; the
// lane's key is A ^ (mask & D)), read a round at a time through a pointer the compiler sees as depending on the round.
template <bool SEL>
struct WideKeys {
  static constexpr int kPerRound = SEL ? 8 : 4;
  const uint32_t* base;
  __device__ const uint32_t* Round(int r, uint32_t dep) const { return base + kPerRound * r + ZeroAfter(dep); }
};
__device__ inline uint32_t Select(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }  // m ? a : b, bitwise

// out = AES_k(in) ^ in  on the wide tables; SEL: k = A ^ (mask & D) per lane.
template <bool SEL>
__device__ __attribute__((always_inline)) inline void WideAesMmo(const WideTab& L, WideKeys<SEL> key, uint32_t mask, const uint32_t in[4],
                                                                 uint32_t out[4]) {
  constexpr int RS = WideKeys<SEL>::kPerRound;
  uint32_t s[4], kc[RS];
  {
    const uint32_t* k0 = key.Round(0, in[0]);
    const uint32_t* k1 = key.Round(1, in[1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = SEL ? Xor3(in[j], k0[j], mask & k0[4 + j]) : (in[j] ^ k0[j]);
#pragma unroll
    for (int j = 0; j < RS; ++j) kc[j] = k1[j];
  }
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    uint32_t a[4], b[4], c[4], d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = L.Lk<0, 0>(s[j]);
      b[j] = L.Lk<1, 1>(s[(j + 1) & 3]);
      c[j] = L.Lk<2, 2>(s[(j + 2) & 3]);
      d[j] = L.Lk<3, 3>(s[(j + 3) & 3]);
    }
    const uint32_t* kn = key.Round(r + 1, a[0]);  // the next round's keys: requested after this round's lookups, used a round later
    uint32_t knx[RS];
#pragma unroll
    for (int j = 0; j < RS; ++j) knx[j] = kn[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t x = Xor3(a[j], b[j], c[j]);
      s[j] = SEL ? Xor3(x, AndXor(mask, kc[4 + j], d[j]), kc[j]) : Xor3(x, d[j], kc[j]);
    }
#pragma unroll
    for (int j = 0; j < RS; ++j) kc[j] = knx[j];
  }
  // last round: T2 has S in byte 0, T3 in byte 1, T0 in byte 2, T1 in byte 3 -- every lookup holds its S-box byte in
  // place, three bitwise selects assemble the column
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t a = L.Lk<0, 2>(s[j]), b = L.Lk<1, 3>(s[(j + 1) & 3]), c = L.Lk<2, 0>(s[(j + 2) & 3]), d = L.Lk<3, 1>(s[(j + 3) & 3]);
    const uint32_t p = Select(L.m_ffff, Select(L.m_00ff, a, b), Select(L.m_ff0000, c, d));
    out[j] = SEL ? Xor3(AndXor(mask, kc[4 + j], p), in[j], kc[j]) : Xor3(p, in[j], kc[j]);
  }
}
// Two blocks of the SAME plaintext under two key sets, round by round side by side (a DCF level's seed and value block;
// both children of a DPF level in Gen): 32 lookups in flight per lane.
template <bool SEL>
__device__ __attribute__((always_inline)) inline void WideAesMmo2(const WideTab& L, WideKeys<SEL> keyA, WideKeys<SEL> keyB, uint32_t maskA,
                                                                  uint32_t maskB, const uint32_t in[4], uint32_t outA[4], uint32_t outB[4]) {
  constexpr int RS = WideKeys<SEL>::kPerRound;
  uint32_t sa[4], sb[4], ka[RS], kb[RS];
  {
    const uint32_t* a0 = keyA.Round(0, in[0]);
    const uint32_t* b0 = keyB.Round(0, in[1]);
    const uint32_t* a1 = keyA.Round(1, in[2]);
    const uint32_t* b1 = keyB.Round(1, in[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sa[j] = SEL ? Xor3(in[j], a0[j], maskA & a0[4 + j]) : (in[j] ^ a0[j]);
      sb[j] = SEL ? Xor3(in[j], b0[j], maskB & b0[4 + j]) : (in[j] ^ b0[j]);
    }
#pragma unroll
    for (int j = 0; j < RS; ++j) { ka[j] = a1[j]; kb[j] = b1[j]; }
  }
#pragma unroll
  for (int r = 1; r < 10; ++r) {
    uint32_t ta[16], tb[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ta[j] = L.Lk<0, 0>(sa[j]); ta[4 + j] = L.Lk<1, 1>(sa[(j + 1) & 3]); ta[8 + j] = L.Lk<2, 2>(sa[(j + 2) & 3]); ta[12 + j] = L.Lk<3, 3>(sa[(j + 3) & 3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tb[j] = L.Lk<0, 0>(sb[j]); tb[4 + j] = L.Lk<1, 1>(sb[(j + 1) & 3]); tb[8 + j] = L.Lk<2, 2>(sb[(j + 2) & 3]); tb[12 + j] = L.Lk<3, 3>(sb[(j + 3) & 3]);
    }
    const uint32_t* na = keyA.Round(r + 1, ta[0]);
    const uint32_t* nb = keyB.Round(r + 1, tb[0]);
    uint32_t nxa[RS], nxb[RS];
#pragma unroll
    for (int j = 0; j < RS; ++j) { nxa[j] = na[j]; nxb[j] = nb[j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t xa = Xor3(ta[j], ta[4 + j], ta[8 + j]), xb = Xor3(tb[j], tb[4 + j], tb[8 + j]);
      sa[j] = SEL ? Xor3(xa, AndXor(maskA, ka[4 + j], ta[12 + j]), ka[j]) : Xor3(xa, ta[12 + j], ka[j]);
      sb[j] = SEL ? Xor3(xb, AndXor(maskB, kb[4 + j], tb[12 + j]), kb[j]) : Xor3(xb, tb[12 + j], kb[j]);
    }
#pragma unroll
    for (int j = 0; j < RS; ++j) { ka[j] = nxa[j]; kb[j] = nxb[j]; }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t a = L.Lk<0, 2>(sa[j]), b = L.Lk<1, 3>(sa[(j + 1) & 3]), c = L.Lk<2, 0>(sa[(j + 2) & 3]), d = L.Lk<3, 1>(sa[(j + 3) & 3]);
    const uint32_t pa = Select(L.m_ffff, Select(L.m_00ff, a, b), Select(L.m_ff0000, c, d));
    const uint32_t a2 = L.Lk<0, 2>(sb[j]), b2 = L.Lk<1, 3>(sb[(j + 1) & 3]), c2 = L.Lk<2, 0>(sb[(j + 2) & 3]), d2 = L.Lk<3, 1>(sb[(j + 3) & 3]);
    const uint32_t pb = Select(L.m_ffff, Select(L.m_00ff, a2, b2), Select(L.m_ff0000, c2, d2));
    outA[j] = SEL ? Xor3(AndXor(maskA, ka[4 + j], pa), in[j], ka[j]) : Xor3(pa, in[j], ka[j]);
    outB[j] = SEL ? Xor3(AndXor(maskB, kb[4 + j], pb), in[j], kb[j]) : Xor3(pb, in[j], kb[j]);
  }
}

// What a wide PRG hands a walk (and what its own Gen / GenSide run on): the per-lane table constants and pointers to the
// object's key words.  sel[p]: the A / D schedules of pair p -- mul = 2: {k0, k0 ^ k1};  mul = 4: {k0, k0 ^ k2}, {k1, k1 ^ k3};
// mul = 1: one plain schedule.
template <int mul>
struct PreparedWide {
  WideTab tab;
  const uint32_t* sel0;
  const uint32_t* sel1;
  mutable uint32_t levels_done = 0;  // (Prepared() starts it from what this wave has published: a thread may walk key after key)
  // Called by the walks once per level.  The sixteen waves of the workgroup start together and share one LDS pipe; the
  // instruction arbiter prefers the oldest wave of a SIMD, so without help the four waves of a SIMD finish one after
  // the other and the last ones run on a pipe they cannot fill alone -- and a workgroup that owns the CU's whole LDS
  // cannot be followed by the next one until its last wave has left.  Every fourth level a wave publishes how far it
  // is (one single-lane LDS write) and reads the three waves it shares its SIMD with (one four-lane read): its
  // priority is the number of them that are ahead (the engine's own closed loop, fss_amd/csrc/fss_dev.hpp: rank_prio).
  __device__ void LevelDone() const {
    ++levels_done;
    if (levels_done & 3u) return;
    uint32_t* prog = WorkgroupWideProgress();
    const unsigned w = (LinearTid() >> 6) & 15u, simd = w & 3u, age = w >> 2;  // waves w, w + 4, w + 8, w + 12 share a SIMD
    const unsigned lane = __lane_id();
    uint32_t other = 0;
    if (lane < 4) {
      if (lane == age) prog[simd * 4 + age] = levels_done;
      other = *reinterpret_cast<const volatile uint32_t*>(prog + simd * 4 + lane);
    }
    switch (__popcll(__ballot(other > levels_done))) {
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
  }
  // A walk has ended: the wave's count is published (a thread that walks key after key -- a grid-stride loop -- carries on
  // from it), its priority is the default again for whatever the caller's kernel does next.
  __device__ void WalkDone() const {
    uint32_t* prog = WorkgroupWideProgress();
    const unsigned w = (LinearTid() >> 6) & 15u;
    if (__lane_id() == 0) prog[(w & 3u) * 4 + (w >> 2)] = levels_done;
    __builtin_amdgcn_s_setprio(0);
  }
  __device__ static uint32_t PublishedLevels() {
    const unsigned w = (LinearTid() >> 6) & 15u;
    return Uniform(WorkgroupWideProgress()[(w & 3u) * 4 + (w >> 2)]);
  }
  __device__ static void Words(int4 v, uint32_t w[4]) {
    w[0] = static_cast<uint32_t>(v.x); w[1] = static_cast<uint32_t>(v.y); w[2] = static_cast<uint32_t>(v.z); w[3] = static_cast<uint32_t>(v.w);
  }
  __device__ static int4 Int4(const uint32_t w[4]) {
    return int4{static_cast<int>(w[0]), static_cast<int>(w[1]), static_cast<int>(w[2]), static_cast<int>(w[3])};
  }
  __device__ __attribute__((always_inline)) std::array<int4, mul> Gen(int4 seed) const {
    std::array<int4, mul> out;
    uint32_t in[4], o[4], o2[4];
    Words(seed, in);
    if constexpr (mul == 1) {
      WideAesMmo<false>(tab, WideKeys<false>{sel0}, 0u, in, o);
      out[0] = Int4(o);
    } else if constexpr (mul == 2) {
      WideAesMmo2<true>(tab, WideKeys<true>{sel0}, WideKeys<true>{sel0}, 0u, ~0u, in, o, o2);
      out[0] = Int4(o); out[1] = Int4(o2);
    } else {
      WideAesMmo2<true>(tab, WideKeys<true>{sel0}, WideKeys<true>{sel1}, 0u, 0u, in, o, o2);   // k0, k1
      out[0] = Int4(o); out[1] = Int4(o2);
      WideAesMmo2<true>(tab, WideKeys<true>{sel0}, WideKeys<true>{sel1}, ~0u, ~0u, in, o, o2);  // k2, k3
      out[2] = Int4(o); out[3] = Int4(o2);
    }
    return out;
  }
  __device__ __attribute__((always_inline)) std::array<int4, (mul + 1) / 2> GenSide(int4 seed, bool right) const
    requires(mul >= 2)
  {
    std::array<int4, mul / 2> out;
    uint32_t in[4], o[4];
    Words(seed, in);
    const uint32_t mask = right ? ~0u : 0u;
    if constexpr (mul == 2) {
      WideAesMmo<true>(tab, WideKeys<true>{sel0}, mask, in, o);
      out[0] = Int4(o);
    } else {
      uint32_t o2[4];
      WideAesMmo2<true>(tab, WideKeys<true>{sel0}, WideKeys<true>{sel1}, mask, mask, in, o, o2);
      out[0] = Int4(o); out[1] = Int4(o2);
    }
    return out;
  }
};
#endif

// The PRG a walk calls level after level: its prepared form where the type has one, the PRG itself otherwise.
template <typename Prg>
__device__ inline decltype(auto) PreparedPrg(const Prg& prg) {
  if constexpr (requires { prg.Prepared(); }) return prg.Prepared();
  else return (prg);
}

// ---------------------------------------------------------------- ChaCha (chacha.cuh:24-128)
// State rows: [constant | seed | seed | 0, 0, nonce]; `rounds` rounds; every row XORed with its input row.
// mul = 1 returns row 1, mul = 2 rows 0-1 (16-byte-key constant), mul = 4 all four rows (32-byte-key constant).
template <int mul, int rounds>
__device__ inline std::array<int4, mul> ChaChaGen(int nonce0, int nonce1, int4 seed) {
  const uint32_t sd[4] = {static_cast<uint32_t>(seed.x), static_cast<uint32_t>(seed.y), static_cast<uint32_t>(seed.z),
                          static_cast<uint32_t>(seed.w)};
  const uint32_t init[16] = {0x61707865u, mul <= 2 ? 0x3120646eu : 0x3320646eu, mul <= 2 ? 0x79622d36u : 0x79622d32u,
                             0x6b206574u, sd[0], sd[1], sd[2], sd[3], sd[0], sd[1], sd[2], sd[3], 0u, 0u,
                             static_cast<uint32_t>(nonce0), static_cast<uint32_t>(nonce1)};
  uint32_t x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = init[i];
  auto quarter = [&x](int a, int b, int c, int d) {
    x[a] += x[b]; x[d] = RotL(x[d] ^ x[a], 16);
    x[c] += x[d]; x[b] = RotL(x[b] ^ x[c], 12);
    x[a] += x[b]; x[d] = RotL(x[d] ^ x[a], 8);
    x[c] += x[d]; x[b] = RotL(x[b] ^ x[c], 7);
  };
#pragma unroll 2
  for (int r = 0; r < rounds; r += 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) quarter(j, 4 + j, 8 + j, 12 + j);  // columns
#pragma unroll
    for (int j = 0; j < 4; ++j) quarter(j, 4 + (j + 1) % 4, 8 + (j + 2) % 4, 12 + (j + 3) % 4);  // diagonals
  }
  std::array<int4, mul> out;
  constexpr int first = mul == 1 ? 1 : 0;
#pragma unroll
  for (int i = 0; i < mul; ++i) {
    const int row = 4 * (first + i);
    out[i] = int4{static_cast<int>(x[row] ^ init[row]), static_cast<int>(x[row + 1] ^ init[row + 1]),
                  static_cast<int>(x[row + 2] ^ init[row + 2]), static_cast<int>(x[row + 3] ^ init[row + 3])};
  }
  return out;
}

// ---------------------------------------------------------------- where a walk reads its key from
// The evaluators below take the key through a small view: S(i) = the seed word of level i's slot, V(i) = its second
// word (DCF), Flag(i) = the tr / extra flag of level i (DPF, Half-Tree's last level), Last() / LastV() = the output
// correction word.  KeyMajor is the reference's Cw array (slot i = words 2i, 2i + 1); the level-major views are what
// fss::gpu::*RelayoutGpu writes (include/fss/point_eval_gpu.cuh:39-91): cw_s[i * n + k], the flags of key k packed
// into one word, out_cw[k] -- read by thread k, so the loads of a wave are coalesced.
struct KeyMajorView {
  const int4* slots;
  __device__ int4 S(int i) const { return slots[2 * i]; }
  __device__ int4 V(int i) const { return slots[2 * i + 1]; }
  __device__ bool Flag(int i) const { return (slots[2 * i + 1].x & 0xff) != 0; }
};
// A key-major key read a cache line at a time.  A thread's key is its own memory: with one slot per level every 128-byte
// line (four slots) is touched on four levels, several microseconds apart -- and with a whole chip of threads streaming
// their keys (32 CUs x 1024 threads x one line = the XCD's entire 4 MiB L2) the line has been evicted in between: the
// per-thread walks fetched 1.8 x their keys from the fabric, 4.9 TB/s at 2.4 G evals/s, and stalled on it
// (profiles/r6/dropin_device_wide_pmc.txt).  Here a thread requests ALL the slots of a line when it reaches the line's
// first slot (one level ahead of their first use, so the loads land under a cipher) and keeps them in registers: every
// line is fetched once.  Slot j lives in buffer position j & 3 -- the same position for every lane, whatever the lane's
// own line phase (keys are 1056 bytes apart: four phases) -- so reads are by a wave-uniform index: no per-lane selects.
// kFullV: the slot's second word is a value (DCF) rather than a flag byte (DPF, Half-Tree): 32 instead of 20 registers.
template <bool kFullV>
struct KeyMajorLines {
  const int4* slots;
  int nslots;
  // The buffer is a shift register of named values: position q holds slot (next to be consumed) + q.  A line's loads
  // go to positions 0 .. 3 in order -- a line is requested exactly when its first slot is the next one to be consumed --
  // and Pop hands out position 0 and moves the rest down.  No runtime index anywhere: an indexed (or selected) read of
  // buffer members is turned into an indexed load by the compiler, and the buffer into a stack object.
  mutable int4 s0_, s1_, s2_, s3_;
  mutable int4 v0_, v1_, v2_, v3_;  // kFullV: the slots' second words; otherwise only .x (the flag byte's word) is kept
  // A line's loads land in staging registers first and move into the buffer after the cipher (Land): loads issued in a
  // conditional block straight into registers that the other lanes keep live are waited for by the compiler at the end of
  // that block -- a memory round trip BEFORE the cipher they were meant to land under, on every level.
  mutable int4 ts0_, ts1_, ts2_, ts3_;
  mutable int4 tv0_, tv1_, tv2_, tv3_;
  mutable bool fresh_ = false;
  // (the buffer starts as zeros, as the aggregate form `{slots, n}` left it: spelled out, so that -Wextra has nothing to say)
  __device__ KeyMajorLines(const int4* key_slots, int n)
      : slots(key_slots), nslots(n), s0_{}, s1_{}, s2_{}, s3_{}, v0_{}, v1_{}, v2_{}, v3_{}, ts0_{}, ts1_{}, ts2_{}, ts3_{}, tv0_{}, tv1_{}, tv2_{}, tv3_{} {}
  __device__ __attribute__((always_inline)) void Take(const int4* p, bool want, int4& s, int4& v) const {
    if (want) {
      s = p[0];
      if constexpr (kFullV) v = p[1];
      else v.x = p[1].x;
    }
  }
  // Called before the cipher under which slot j's loads are to land (j = the next slot Pop will hand out): requests slot
  // j's line if j is the first slot of one (or the key's first slot).  A lane's slots start a line every fourth level, at
  // a phase of its own (the key's address): `fresh` is one compare per level, the bounds are wave-uniform, and the
  // line's eight loads share one address.
  __device__ __attribute__((always_inline)) void Ahead(int j) const {
    fresh_ = false;
    if (j >= nslots) return;  // wave-uniform
    const int4* p = slots + 2 * j;
    if constexpr (kFullV) {
      // (DCF: the slot's own address decides; the phase form below measured 3 % slower here and 3 % faster for DPF,
      // profiles/r6/dropin_device_test.txt)
      const uintptr_t a = reinterpret_cast<uintptr_t>(p);
      fresh_ = j == 0 || (a & 127u) == 0;
      const int room = 4 - static_cast<int>((a >> 5) & 3u);  // slots from here to the end of the line
      if (fresh_) {
        Take(p, true, ts0_, tv0_);
        Take(p + 2, room > 1 && j + 1 < nslots, ts1_, tv1_);
        Take(p + 4, room > 2 && j + 2 < nslots, ts2_, tv2_);
        Take(p + 6, room > 3 && j + 3 < nslots, ts3_, tv3_);
      }
    } else {
      const int phase = static_cast<int>((reinterpret_cast<uintptr_t>(slots) >> 5) & 3u);  // slots from a line boundary back to slot 0
      if (j == 0) {
        fresh_ = true;
        const int room = 4 - phase;  // slots from slot 0 to the end of its line
        Take(p, true, ts0_, tv0_);
        Take(p + 2, room > 1 && 1 < nslots, ts1_, tv1_);
        Take(p + 4, room > 2 && 2 < nslots, ts2_, tv2_);
        Take(p + 6, room > 3 && 3 < nslots, ts3_, tv3_);
      } else {
        fresh_ = ((j + phase) & 3) == 0;
        if (fresh_) {
          Take(p, true, ts0_, tv0_);
          Take(p + 2, j + 1 < nslots, ts1_, tv1_);
          Take(p + 4, j + 2 < nslots, ts2_, tv2_);
          Take(p + 6, j + 3 < nslots, ts3_, tv3_);
        }
      }
    }
  }
  // After the cipher: the next slot's two words (for !kFullV only v.x is meaningful: the word that holds the flag byte), and
  // the buffer moved down by one -- for the lanes that requested a line before the cipher, out of the staging registers
  // (one select per register: "land, then shift" as two steps would be two moves).
  __device__ __attribute__((always_inline)) void Pop(int4& s, int4& v) const {
    // (the predicate reaches the selects through an opaque copy: selects that the compiler can prove to be governed by the
    // very condition of the loads' block are folded back into that block -- and the loads are waited for there again)
    uint64_t m = __ballot(fresh_);
    asm volatile("" : "+s"(m));
    const bool f = (m >> __lane_id()) & 1u;
    auto pick = [f](const int4& t, const int4& o) { return int4{f ? t.x : o.x, f ? t.y : o.y, f ? t.z : o.z, f ? t.w : o.w}; };
    auto pick_v = [f](const int4& t, const int4& o) {
      if constexpr (kFullV) return int4{f ? t.x : o.x, f ? t.y : o.y, f ? t.z : o.z, f ? t.w : o.w};
      else return int4{f ? t.x : o.x, 0, 0, 0};
    };
    s = pick(ts0_, s0_); v = pick_v(tv0_, v0_);
    s0_ = pick(ts1_, s1_); v0_ = pick_v(tv1_, v1_);
    s1_ = pick(ts2_, s2_); v1_ = pick_v(tv2_, v2_);
    s2_ = pick(ts3_, s3_); v2_ = pick_v(tv3_, v3_);
  }
};
// A walk in lines mode calls this on its point before the level loop: a use the compiler must satisfy HERE.  The point
// was typically loaded just before the walk (`xs[tid]`); left pending into the loop, its wait is repeated by the compiler
// on every iteration as a full `s_waitcnt vmcnt(0)` -- which would also wait for the line loads the cursor has in flight.
template <typename In>
__device__ __attribute__((always_inline)) inline void SettleBeforeLoop(const In& x) {
  const uint32_t lo = static_cast<uint32_t>(x);
  asm volatile("" : : "v"(lo));
  if constexpr (sizeof(In) > 4) {
    const uint32_t hi = static_cast<uint32_t>(x >> 32);
    asm volatile("" : : "v"(hi));
  }
  if constexpr (sizeof(In) > 8) {
    const uint32_t h2 = static_cast<uint32_t>(x >> 64), h3 = static_cast<uint32_t>(x >> 96);
    asm volatile("" : : "v"(h2), "v"(h3));
  }
}
__device__ __attribute__((always_inline)) inline void SettleBeforeLoop(const int4& v) {  // ... and the party's seed
  asm volatile("" : : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
}
// PRG types whose walks read key-major keys a line at a time (the wide form: its walks are bound by the LDS pipe and the
// memory system, with registers to spare; the default form is bound by VALU issue and keeps the two loads per level)
template <typename Prg>
concept LineWalkPrg = requires { requires Prg::kKeyMajorLines; };

struct DpfLevelMajorView {
  const int4* cw_s;
  const int4* out_cw;
  uint64_t tr_lo, tr_hi;  // tr bits of key k: levels 0 .. 63 and 64 .. 127
  int64_t n, k;
  int in_bits;
  __device__ int4 S(int i) const { return i == in_bits ? out_cw[k] : cw_s[i * n + k]; }
  __device__ bool Flag(int i) const { return ((i < 64 ? tr_lo >> i : tr_hi >> (i - 64)) & 1) != 0; }
};
struct DcfLevelMajorView {
  const int4* cw_s;
  const int4* cw_v;
  const int4* out_cw;  // the v word of slot in_bits
  int64_t n, k;
  int in_bits;
  __device__ int4 S(int i) const { return cw_s[i * n + k]; }
  __device__ int4 V(int i) const { return i == in_bits ? out_cw[k] : cw_v[i * n + k]; }
};
struct HtLevelMajorView {
  const int4* cw_s;
  uint32_t extra;  // bit 0: the flag of the last level
  int64_t n, k;
  __device__ int4 S(int i) const { return cw_s[i * n + k]; }
  __device__ bool Flag(int) const { return (extra & 1) != 0; }
};

// ---------------------------------------------------------------- DPF (dpf.cuh:93-214)
// Gen: both parties' paths towards alpha are walked together; per level the correction word cancels the seeds of
// the child OFF the path and leaves the control bits differing exactly ON it.
// The in_bits levels (slots 0 .. in_bits - 1); on return seed[p] / t[p] are the parties' states at the leaf alpha.
template <int in_bits, typename In, typename Prg>
__device__ inline void DpfGenLevels(const Prg& prg, int4* slots, const int4 s0s[2], In a, int4 seed[2], bool t[2]) {
  seed[0] = WithCtl(s0s[0], false);
  seed[1] = WithCtl(s0s[1], false);
  t[0] = false;
  t[1] = true;
#pragma unroll 1  // one level per iteration: a level is one to four whole ciphers of straight-line code
  for (int i = 0; i < in_bits; ++i) {
    const std::array<int4, 2> g[2] = {prg.Gen(seed[0]), prg.Gen(seed[1])};
    const int keep = BitAt(a, in_bits - 1 - i), lose = 1 - keep;
    const int4 s_cw = X4(WithCtl(g[0][lose], false), WithCtl(g[1][lose], false));
    const bool t_cw[2] = {static_cast<bool>(Ctl(g[0][0]) ^ Ctl(g[1][0]) ^ keep ^ 1),
                          static_cast<bool>(Ctl(g[0][1]) ^ Ctl(g[1][1]) ^ keep)};
    for (int p = 0; p < 2; ++p) {
      int4 s = WithCtl(g[p][keep], false);
      bool ct = Ctl(g[p][keep]);
      if (t[p]) {
        s = X4(s, s_cw);
        ct ^= t_cw[keep];
      }
      seed[p] = s;
      t[p] = ct;
    }
    slots[2 * i] = WithCtl(s_cw, t_cw[0]);
    slots[2 * i + 1] = int4{t_cw[1], 0, 0, 0};  // whole slot written (dpf.cuh:147-153)
  }
}
template <int in_bits, typename Group, typename In, typename Prg>
__device__ inline void DpfGen(const Prg& prg, int4* slots, const int4 s0s[2], In a, int4 beta) {
  int4 seed[2];
  bool t[2];
  DpfGenLevels<in_bits, In>(prg, slots, s0s, a, seed, t);
  Group v = Group::From(WithCtl(beta, false)) + (-Group::From(seed[0])) + Group::From(seed[1]);
  if (t[1]) v = -v;
  slots[2 * in_bits] = v.Into();
  slots[2 * in_bits + 1] = int4{0, 0, 0, 0};
}

// The walks below ask for the NEXT level's correction word before they start this level's cipher and apply it without a
// branch: a thread's key is its own cache lines (key-major: 32 bytes per level), and a load issued where it is used
// -- inside `if (t)` -- stalls the wave for a memory round trip per level with nothing of its own left to issue.
// The walk down to the leaf x: on return s / t are the leaf's seed (control bit cleared) and control bit, `last` the word
// of slot in_bits (the output correction word), requested a level ahead like every other slot.
template <int in_bits, typename In, typename Prg, typename Key>
__device__ __attribute__((always_inline)) inline void DpfWalkOn(const Prg& prg_, bool b, int4 s0, const Key& key, In x, int4& s_out, bool& t_out,
                                                                int4& last) {
  auto&& prg = PreparedPrg(prg_);
  constexpr bool kLines = requires { key.Ahead(0); };  // KeyMajorLines: slots arrive a cache line at a time
  int4 s = WithCtl(s0, false);
  bool t = b;
  int4 cw;
  bool flag;
  if constexpr (kLines) {
    int4 w;
    key.Ahead(0);
    key.Pop(cw, w);
    flag = (w.x & 0xff) != 0;
    SettleBeforeLoop(x);
    SettleBeforeLoop(s);
  } else {
    cw = key.S(0);
    flag = key.Flag(0);
  }
#pragma unroll 1  // one level per iteration: a level is one to four whole ciphers of straight-line code
  for (int i = 0; i < in_bits; ++i) {
    int4 cw_next;  // slot i + 1 (slot in_bits: the output correction word)
    bool flag_next;
    if constexpr (kLines) {
      key.Ahead(i + 1);  // requests the line of slot i + 1 if that slot starts one; taken from the buffer after the cipher
    } else {
      cw_next = key.S(i + 1);
      flag_next = key.Flag(i + 1 < in_bits ? i + 1 : i);  // (no branch around the load: it would be waited for at once)
    }
    const int side = BitAt(x, in_bits - 1 - i);
    int4 taken;  // Gen(s)[side]; a PRG that can compute one side alone is asked for just that
    if constexpr (requires { prg.GenSide(s, true); }) taken = prg.GenSide(s, side != 0)[0];
    else taken = prg.Gen(s)[side];
    const bool side_cw = side ? flag : Ctl(cw);
    s = X4(WithCtl(taken, false), If(t, WithCtl(cw, false)));
    t = Ctl(taken) ^ (t && side_cw);
    if constexpr (kLines) {
      int4 w;
      key.Pop(cw_next, w);
      flag_next = (w.x & 0xff) != 0;  // (slot in_bits has no flag: unused after the last level)
    }
    cw = cw_next;
    flag = flag_next;
    if constexpr (requires { prg.LevelDone(); }) prg.LevelDone();
  }
  if constexpr (requires { prg.WalkDone(); }) prg.WalkDone();
  s_out = s;
  t_out = t;
  last = cw;
}
template <int in_bits, typename Group, typename In, typename Prg, typename Key>
__device__ __attribute__((always_inline)) inline int4 DpfEvalOn(const Prg& prg, bool b, int4 s0, const Key& key, In x) {
  int4 s, cw;
  bool t;
  DpfWalkOn<in_bits, In>(prg, b, s0, key, x, s, t, cw);
  Group y = Group::From(s);
  if (t) y = y + Group::From(cw);
  if (b) y = -y;
  return y.Into();
}
template <int in_bits, typename Group, typename In, typename Prg>
__device__ __attribute__((always_inline)) inline int4 DpfEval(const Prg& prg, bool b, int4 s0, const int4* slots, In x) {
  if constexpr (LineWalkPrg<Prg>) return DpfEvalOn<in_bits, Group, In>(prg, b, s0, KeyMajorLines<false>{slots, in_bits + 1}, x);
  else return DpfEvalOn<in_bits, Group, In>(prg, b, s0, KeyMajorView{slots}, x);
}

// ---------------------------------------------------------------- DCF (dcf.cuh:108-276); PRG output [sl, vl, sr, vr]
template <int in_bits, typename Group, typename In, bool gt, typename Prg>
__device__ inline void DcfGen(const Prg& prg, int4* slots, const int4 s0s[2], In a, int4 beta) {
  int4 seed[2] = {WithCtl(s0s[0], false), WithCtl(s0s[1], false)};
  bool t[2] = {false, true};
  const Group gbeta = Group::From(WithCtl(beta, false));
  Group acc;  // zero
#pragma unroll 1  // one level per iteration: a level is one to four whole ciphers of straight-line code
  for (int i = 0; i < in_bits; ++i) {
    const std::array<int4, 4> g0 = prg.Gen(seed[0]), g1 = prg.Gen(seed[1]);
    const bool keep = BitAt(a, in_bits - 1 - i);  // the side ON the path; the other one is lost
    // [sl, vl, sr, vr] of each party, sorted into the kept and the lost side by selects (a runtime index into the
    // PRG's output array would put it on the stack)
    const int4 ks[2] = {keep ? g0[2] : g0[0], keep ? g1[2] : g1[0]}, kv[2] = {keep ? g0[3] : g0[1], keep ? g1[3] : g1[1]};
    const int4 ls[2] = {keep ? g0[0] : g0[2], keep ? g1[0] : g1[2]}, lv[2] = {keep ? g0[1] : g0[3], keep ? g1[1] : g1[3]};
    const int4 s_cw = X4(WithCtl(ls[0], false), WithCtl(ls[1], false));
    auto val = [](int4 v) { return Group::From(WithCtl(v, false)); };
    // the value word: everything accumulated so far is cancelled off the path, and beta is released there when the
    // off-path side is the side the predicate selects (x < alpha lies left of a 1 bit; x > alpha right of a 0 bit)
    Group v_cw = -acc + val(lv[1]) + (-val(lv[0]));
    if (gt ? !keep : keep) v_cw = v_cw + gbeta;
    if (t[1]) v_cw = -v_cw;
    acc = acc + (-val(kv[1])) + val(kv[0]);
    acc = t[1] ? acc + (-v_cw) : acc + v_cw;
    const bool t_cw[2] = {static_cast<bool>(Ctl(g0[0]) ^ Ctl(g1[0]) ^ keep ^ 1), static_cast<bool>(Ctl(g0[2]) ^ Ctl(g1[2]) ^ keep)};
    const bool t_keep = keep ? t_cw[1] : t_cw[0];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int4 s = WithCtl(ks[p], false);
      bool ct = Ctl(ks[p]);
      if (t[p]) {
        s = X4(s, s_cw);
        ct ^= t_keep;
      }
      seed[p] = s;
      t[p] = ct;
    }
    slots[2 * i] = WithCtl(s_cw, t_cw[0]);
    slots[2 * i + 1] = WithCtl(v_cw.Into(), t_cw[1]);
  }
  Group last = Group::From(seed[1]) + (-Group::From(seed[0])) + (-acc);
  if (t[1]) last = -last;
  slots[2 * in_bits] = int4{0, 0, 0, 0};
  slots[2 * in_bits + 1] = last.Into();
}

template <int in_bits, typename Group, typename In, typename Prg, typename Key>
__device__ __attribute__((always_inline)) inline int4 DcfEvalOn(const Prg& prg_, bool b, int4 s0, const Key& key, In x) {
  auto&& prg = PreparedPrg(prg_);
  constexpr bool kLines = requires { key.Ahead(0); };  // KeyMajorLines: slots arrive a cache line at a time
  int4 s = WithCtl(s0, false);
  bool t = b;
  if constexpr (kLines) {
    SettleBeforeLoop(x);
    SettleBeforeLoop(s);
  }
  Group acc;
  auto add = [b](Group sum, Group term) { return b ? sum + (-term) : sum + term; };  // party 1 subtracts
  // (no request-ahead of the correction words here: a level is two ciphers side by side, the loads measured as hidden,
  // and eight more live registers put a caller's kernel without launch bounds over its 128)
#pragma unroll 1  // one level per iteration: a level is one to four whole ciphers of straight-line code
  for (int i = 0; i < in_bits; ++i) {
    if constexpr (kLines) key.Ahead(i);  // slot i's line, if it starts one: lands under this level's ciphers
    const int side = BitAt(x, in_bits - 1 - i);
    int4 ts, tv;  // Gen(s)[2 side], Gen(s)[2 side + 1]
    if constexpr (requires { prg.GenSide(s, true); }) {
      const std::array<int4, 2> g = prg.GenSide(s, side != 0);
      ts = g[0]; tv = g[1];
    } else {
      const std::array<int4, 4> g = prg.Gen(s);
      ts = g[2 * side]; tv = g[2 * side + 1];
    }
    int4 cw_s, cw_v;
    if constexpr (kLines) {
      key.Pop(cw_s, cw_v);
    } else {
      cw_s = key.S(i);
      cw_v = key.V(i);
    }
    acc = add(acc, Group::From(WithCtl(tv, false)));
    if (t) acc = add(acc, Group::From(WithCtl(cw_v, false)));
    int4 child = WithCtl(ts, false);
    bool ct = Ctl(ts);
    if (t) {
      child = X4(child, WithCtl(cw_s, false));
      ct ^= side ? Ctl(cw_v) : Ctl(cw_s);
    }
    s = child;
    t = ct;
    if constexpr (requires { prg.LevelDone(); }) prg.LevelDone();
  }
  if constexpr (requires { prg.WalkDone(); }) prg.WalkDone();
  int4 last_v;
  if constexpr (kLines) {
    int4 unused;
    key.Ahead(in_bits);
    key.Pop(unused, last_v);
  } else {
    last_v = key.V(in_bits);
  }
  acc = add(acc, Group::From(s));
  if (t) acc = add(acc, Group::From(last_v));
  return acc.Into();
}
template <int in_bits, typename Group, typename In, typename Prg>
__device__ __attribute__((always_inline)) inline int4 DcfEval(const Prg& prg, bool b, int4 s0, const int4* slots, In x) {
  if constexpr (LineWalkPrg<Prg>) return DcfEvalOn<in_bits, Group, In>(prg, b, s0, KeyMajorLines<true>{slots, in_bits + 1}, x);
  else return DcfEvalOn<in_bits, Group, In>(prg, b, s0, KeyMajorView{slots}, x);
}

// ---------------------------------------------------------------- verifiable DPF (vdpf.cuh:103-257)
// The DPF's levels with no output slot (a key is in_bits slots), then: cs = H(Pack(a), s0) ^ H(Pack(a), s1) over the two
// parties' leaf seeds, the output correction word from the same seeds, and "retry" when the leaf control bits agree.
// Eval is the DPF walk plus one H per point, corrected by cs where the control bit is set.
template <typename In>
__host__ __device__ inline int4 PackIn(In val) {  // util::Pack (util.cuh:47-64)
  const unsigned __int128 v = static_cast<unsigned __int128>(val);
  return int4{static_cast<int>(static_cast<uint32_t>(v)), sizeof(In) > 4 ? static_cast<int>(static_cast<uint32_t>(v >> 32)) : 0,
              sizeof(In) > 8 ? static_cast<int>(static_cast<uint32_t>(v >> 64)) : 0,
              sizeof(In) > 8 ? static_cast<int>(static_cast<uint32_t>(v >> 96)) : 0};
}
struct VdpfKeyView {  // in_bits slots; "slot in_bits" is the key's separate output correction word
  const int4* slots;
  int4 ocw;
  int in_bits;
  __device__ int4 S(int i) const { return i == in_bits ? ocw : slots[2 * i]; }
  __device__ bool Flag(int i) const { return (slots[2 * i + 1].x & 0xff) != 0; }
};
template <int in_bits, typename Group, typename In, typename Prg, typename XorHash>
__device__ inline int VdpfGen(const Prg& prg, XorHash xor_hash, int4* slots, std::array<int4, 4>& cs, int4& ocw, const int4 s0s[2], In a,
                              int4 beta) {
  int4 seed[2];
  bool t[2];
  DpfGenLevels<in_bits, In>(prg, slots, s0s, a, seed, t);
  const int4 a_buf = PackIn(a);
  const std::array<int4, 4> h0 = xor_hash.Hash(std::tuple<int4, const int4>{a_buf, seed[0]});
  const std::array<int4, 4> h1 = xor_hash.Hash(std::tuple<int4, const int4>{a_buf, seed[1]});
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = X4(h0[j], h1[j]);
  if (t[0] == t[1]) return 1;  // the caller resamples the seeds (vdpf.cuh:171-172); ocw is left as it was
  Group v = Group::From(WithCtl(beta, false)) + (-Group::From(seed[0])) + Group::From(seed[1]);
  if (t[1]) v = -v;
  ocw = v.Into();
  return 0;
}
template <int in_bits, typename Group, typename In, typename Prg, typename XorHash>
__device__ inline std::array<int4, 4> VdpfEval(const Prg& prg, XorHash xor_hash, bool b, int4 s0, const int4* slots, const int4* cs, int4 ocw,
                                               In x, int4& y) {
  int4 s, cw;
  bool t;
  DpfWalkOn<in_bits, In>(prg, b, s0, VdpfKeyView{slots, ocw, in_bits}, x, s, t, cw);
  Group g = Group::From(s);
  if (t) g = g + Group::From(cw);
  if (b) g = -g;
  y = g.Into();
  std::array<int4, 4> pi = xor_hash.Hash(std::tuple<int4, const int4>{PackIn(x), s});
#pragma unroll
  for (int j = 0; j < 4; ++j) pi[j] = X4(pi[j], If(t, cs[j]));
  return pi;
}

// ---------------------------------------------------------------- Half-Tree DPF (half_tree_dpf.cuh:68-231)
// A node carries its control bit in place; H(v) = prg.Gen(hash_key ^ v)[0].  Levels above the last: the left child
// is H(node) (+ cw if t), the right child that XOR node.  Last level: one hash per leaf of SetLsb(node, x bit).
template <typename Prg>
__device__ inline int4 HtHash(const Prg& prg, int4 hash_key, int4 v) {
  return prg.Gen(X4(hash_key, v))[0];
}

template <int in_bits, typename Group, typename In, typename Prg>
__device__ inline void HtGen(const Prg& prg, int4 hash_key, int4* slots, int4& ocw, const int4 s0s[2], In a, int4 beta) {
  int4 node[2] = {WithCtl(s0s[0], false), WithCtl(s0s[1], true)};
#pragma unroll 1  // one level per iteration: a level is one to four whole ciphers of straight-line code
  for (int i = 0; i < in_bits - 1; ++i) {
    const int4 h[2] = {HtHash(prg, hash_key, node[0]), HtHash(prg, hash_key, node[1])};
    const bool a_bit = BitAt(a, in_bits - 1 - i);
    // off the path both parties must end on the same node: left children differ by h0 ^ h1, right ones by that ^ delta
    const int4 cw = X4(X4(h[0], h[1]), If(!a_bit, X4(node[0], node[1])));
    for (int p = 0; p < 2; ++p) node[p] = X4(X4(h[p], If(a_bit, node[p])), If(Ctl(node[p]), cw));
    slots[2 * i] = cw;
    slots[2 * i + 1] = int4{0, 0, 0, 0};
  }
  const int a_n = BitAt(a, 0);
  int4 h[2][2];  // [party][sigma]
  for (int p = 0; p < 2; ++p)
    for (int sg = 0; sg < 2; ++sg) h[p][sg] = HtHash(prg, hash_key, WithCtl(node[p], sg));
  const int4 hcw = X4(WithCtl(h[0][1 - a_n], false), WithCtl(h[1][1 - a_n], false));  // cancels the OTHER leaf
  const bool lcw[2] = {static_cast<bool>(Ctl(h[0][0]) ^ Ctl(h[1][0]) ^ (a_n == 0)),
                       static_cast<bool>(Ctl(h[0][1]) ^ Ctl(h[1][1]) ^ (a_n == 1))};
  slots[2 * (in_bits - 1)] = WithCtl(hcw, lcw[0]);
  slots[2 * (in_bits - 1) + 1] = int4{lcw[1], 0, 0, 0};
  int4 leaf[2];
  for (int p = 0; p < 2; ++p) leaf[p] = X4(h[p][a_n], If(Ctl(node[p]), WithCtl(hcw, lcw[a_n])));
  Group v = Group::From(WithCtl(beta, false)) + (-Group::From(WithCtl(leaf[0], false))) + Group::From(WithCtl(leaf[1], false));
  if (Ctl(leaf[1])) v = -v;
  ocw = v.Into();
}

template <int in_bits, typename Group, typename In, typename Prg, typename Key>
__device__ __attribute__((always_inline)) inline int4 HtEvalOn(const Prg& prg_, int4 hash_key, bool b, int4 s0, const Key& key, int4 ocw, In x) {
  auto&& prg = PreparedPrg(prg_);
  constexpr bool kLines = requires { key.Ahead(0); };
  int4 node = WithCtl(s0, b);
  int4 cw, cw_word1;  // the current level's slot: its seed word, and (lines mode) the word that holds its flag byte
  if constexpr (kLines) {
    key.Ahead(0);
    key.Pop(cw, cw_word1);
    SettleBeforeLoop(x);
    SettleBeforeLoop(node);
  } else {
    cw = key.S(0);
  }
#pragma unroll 1  // one level per iteration: a level is one to four whole ciphers of straight-line code
  for (int i = 0; i < in_bits - 1; ++i) {
    int4 cw_next;  // (see DpfEvalOn)
    if constexpr (kLines) key.Ahead(i + 1);
    else cw_next = key.S(i + 1);
    const int4 h = HtHash(prg, hash_key, node);
    node = X4(X4(h, If(BitAt(x, in_bits - 1 - i), node)), If(Ctl(node), cw));
    if constexpr (kLines) key.Pop(cw_next, cw_word1);
    cw = cw_next;
    if constexpr (requires { prg.LevelDone(); }) prg.LevelDone();
  }
  if constexpr (requires { prg.WalkDone(); }) prg.WalkDone();
  const bool sigma = BitAt(x, 0), t = Ctl(node);
  const int4 last = cw;  // key.S(in_bits - 1)
  bool last_flag;
  if constexpr (kLines) last_flag = (cw_word1.x & 0xff) != 0;
  else last_flag = key.Flag(in_bits - 1);
  const bool lcw = sigma ? last_flag : Ctl(last);
  const int4 h = HtHash(prg, hash_key, WithCtl(node, sigma));
  const int4 high = X4(WithCtl(h, false), If(t, WithCtl(last, false)));
  const bool low = Ctl(h) ^ (t && lcw);
  Group y = Group::From(high);
  if (low) y = y + Group::From(ocw);
  if (b) y = -y;
  return y.Into();
}
template <int in_bits, typename Group, typename In, typename Prg>
__device__ __attribute__((always_inline)) inline int4 HtEval(const Prg& prg, int4 hash_key, bool b, int4 s0, const int4* slots, int4 ocw, In x) {
  if constexpr (LineWalkPrg<Prg>) return HtEvalOn<in_bits, Group, In>(prg, hash_key, b, s0, KeyMajorLines<false>{slots, in_bits}, ocw, x);
  else return HtEvalOn<in_bits, Group, In>(prg, hash_key, b, s0, KeyMajorView{slots}, ocw, x);
}

}  // namespace thread
}  // namespace fss
