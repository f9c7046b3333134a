// fss_amd_thread.hpp -- Gen / Eval of one key by ONE GPU thread, for calls from inside a caller's own kernel
// (the reference's members are __host__ __device__ and its samples use them that way: samples/dpf_dcf_gpu.cu:39-80,
// src/bench_gpu.cu:85-95).  Included by fss_amd.hpp; the scheme classes forward their device-side Gen / Eval here.
//
// Everything below is __device__ code written against the two hooks only -- Prgable (`prg.Gen(seed)`) and Groupable
// (`From / + / - / Into`) -- so a caller's own PRG or group type works as well as the built-in ones, provided its
// members are callable on the device.  It is the compatibility form: one thread walks one key with its PRG in
// registers and tables in constant memory.  The engine's batched kernels behind EvalPoints / fss::gpu::* (LDS-resident
// tables, two AES streams per lane, level-major keys) are the fast forms; nothing here is used by them.
//
// Correction-word slots are addressed as raw int4 pairs (slot i = words 2i, 2i+1), the layout all three Cw structs
// share: word 0 = s, word 1 = v (DCF) or the flag byte tr / extra at byte 16 (DPF, Half-Tree).
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <cstdint>

namespace fss {
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
// Little-endian words throughout: the int4's x, y, z, w are the four state columns, byte 0 of the block in the low
// byte of x (the byte order of aes128_mmo.cuh:80-84, which encrypts the int4's bytes in memory order).
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

// Round keys as 44 little-endian words (FIPS-197 5.2), computed wherever the PRG object is constructed.
__host__ __device__ constexpr void AesExpandKey(const unsigned char key[16], uint32_t rk[44]) {
  for (int i = 0; i < 4; ++i)
    rk[i] = static_cast<uint32_t>(key[4 * i]) | static_cast<uint32_t>(key[4 * i + 1]) << 8 |
            static_cast<uint32_t>(key[4 * i + 2]) << 16 | static_cast<uint32_t>(key[4 * i + 3]) << 24;
  uint32_t rcon = 1;
  for (int i = 4; i < 44; ++i) {
    uint32_t w = rk[i - 1];
    if (i % 4 == 0) {
      w = w >> 8 | w << 24;  // RotWord on a little-endian word
      w = static_cast<uint32_t>(kAes.sbox[w & 0xff]) | static_cast<uint32_t>(kAes.sbox[w >> 8 & 0xff]) << 8 |
          static_cast<uint32_t>(kAes.sbox[w >> 16 & 0xff]) << 16 | static_cast<uint32_t>(kAes.sbox[w >> 24]) << 24;
      w ^= rcon;
      rcon = Times2(static_cast<uint8_t>(rcon));
    }
    rk[i] = rk[i - 4] ^ w;
  }
}

__device__ inline uint32_t RotL(uint32_t v, int n) { return __builtin_rotateleft32(v, n); }

// out = AES_rk(seed) ^ seed  (Matyas-Meyer-Oseas, aes128_mmo.cuh:79-89)
__device__ inline int4 AesMmo(const uint32_t rk[44], int4 seed) {
  const uint32_t in[4] = {static_cast<uint32_t>(seed.x), static_cast<uint32_t>(seed.y), static_cast<uint32_t>(seed.z),
                          static_cast<uint32_t>(seed.w)};
  uint32_t c[4] = {in[0] ^ rk[0], in[1] ^ rk[1], in[2] ^ rk[2], in[3] ^ rk[3]};
#pragma unroll 1
  for (int r = 1; r < 10; ++r) {
    uint32_t n[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)  // column j takes row k from column j + k (ShiftRows), Te_k = Te0 rotated by k bytes
      n[j] = kAes.te0[c[j] & 0xff] ^ RotL(kAes.te0[c[(j + 1) & 3] >> 8 & 0xff], 8) ^
             RotL(kAes.te0[c[(j + 2) & 3] >> 16 & 0xff], 16) ^ RotL(kAes.te0[c[(j + 3) & 3] >> 24], 24) ^ rk[4 * r + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = n[j];
  }
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    o[j] = (static_cast<uint32_t>(kAes.sbox[c[j] & 0xff]) | static_cast<uint32_t>(kAes.sbox[c[(j + 1) & 3] >> 8 & 0xff]) << 8 |
            static_cast<uint32_t>(kAes.sbox[c[(j + 2) & 3] >> 16 & 0xff]) << 16 |
            static_cast<uint32_t>(kAes.sbox[c[(j + 3) & 3] >> 24]) << 24) ^
           rk[40 + j] ^ in[j];
  return int4{static_cast<int>(o[0]), static_cast<int>(o[1]), static_cast<int>(o[2]), static_cast<int>(o[3])};
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

// ---------------------------------------------------------------- DPF (dpf.cuh:93-214)
// Gen: both parties' paths towards alpha are walked together; per level the correction word cancels the seeds of
// the child OFF the path and leaves the control bits differing exactly ON it.
template <int in_bits, typename Group, typename In, typename Prg>
__device__ inline void DpfGen(const Prg& prg, int4* slots, const int4 s0s[2], In a, int4 beta) {
  int4 seed[2] = {WithCtl(s0s[0], false), WithCtl(s0s[1], false)};
  bool t[2] = {false, true};
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
  Group v = Group::From(WithCtl(beta, false)) + (-Group::From(seed[0])) + Group::From(seed[1]);
  if (t[1]) v = -v;
  slots[2 * in_bits] = v.Into();
  slots[2 * in_bits + 1] = int4{0, 0, 0, 0};
}

template <int in_bits, typename Group, typename In, typename Prg>
__device__ inline int4 DpfEval(const Prg& prg, bool b, int4 s0, const int4* slots, In x) {
  int4 s = WithCtl(s0, false);
  bool t = b;
  for (int i = 0; i < in_bits; ++i) {
    const std::array<int4, 2> g = prg.Gen(s);
    const int side = BitAt(x, in_bits - 1 - i);
    const int4 cw = slots[2 * i];
    const bool side_cw = side ? (slots[2 * i + 1].x & 0xff) != 0 : Ctl(cw);
    int4 child = WithCtl(g[side], false);
    bool ct = Ctl(g[side]);
    if (t) {
      child = X4(child, WithCtl(cw, false));
      ct ^= side_cw;
    }
    s = child;
    t = ct;
  }
  Group y = Group::From(s);
  if (t) y = y + Group::From(slots[2 * in_bits]);
  if (b) y = -y;
  return y.Into();
}

// ---------------------------------------------------------------- DCF (dcf.cuh:108-276); PRG output [sl, vl, sr, vr]
template <int in_bits, typename Group, typename In, bool gt, typename Prg>
__device__ inline void DcfGen(const Prg& prg, int4* slots, const int4 s0s[2], In a, int4 beta) {
  int4 seed[2] = {WithCtl(s0s[0], false), WithCtl(s0s[1], false)};
  bool t[2] = {false, true};
  const Group gbeta = Group::From(WithCtl(beta, false));
  Group acc;  // zero
  for (int i = 0; i < in_bits; ++i) {
    const std::array<int4, 4> g[2] = {prg.Gen(seed[0]), prg.Gen(seed[1])};
    const int keep = BitAt(a, in_bits - 1 - i), lose = 1 - keep;
    const int4 s_cw = X4(WithCtl(g[0][2 * lose], false), WithCtl(g[1][2 * lose], false));
    auto val = [&g](int party, int side) { return Group::From(WithCtl(g[party][2 * side + 1], false)); };
    // the value word: everything accumulated so far is cancelled off the path, and beta is released there when the
    // off-path side is the side the predicate selects (x < alpha lies left of a 1 bit; x > alpha right of a 0 bit)
    Group v_cw = -acc + val(1, lose) + (-val(0, lose));
    if (gt ? lose == 1 : lose == 0) v_cw = v_cw + gbeta;
    if (t[1]) v_cw = -v_cw;
    acc = acc + (-val(1, keep)) + val(0, keep);
    acc = t[1] ? acc + (-v_cw) : acc + v_cw;
    const bool t_cw[2] = {static_cast<bool>(Ctl(g[0][0]) ^ Ctl(g[1][0]) ^ keep ^ 1),
                          static_cast<bool>(Ctl(g[0][2]) ^ Ctl(g[1][2]) ^ keep)};
    for (int p = 0; p < 2; ++p) {
      int4 s = WithCtl(g[p][2 * keep], false);
      bool ct = Ctl(g[p][2 * keep]);
      if (t[p]) {
        s = X4(s, s_cw);
        ct ^= t_cw[keep];
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

template <int in_bits, typename Group, typename In, typename Prg>
__device__ inline int4 DcfEval(const Prg& prg, bool b, int4 s0, const int4* slots, In x) {
  int4 s = WithCtl(s0, false);
  bool t = b;
  Group acc;
  auto add = [b](Group sum, Group term) { return b ? sum + (-term) : sum + term; };  // party 1 subtracts
  for (int i = 0; i < in_bits; ++i) {
    const std::array<int4, 4> g = prg.Gen(s);
    const int side = BitAt(x, in_bits - 1 - i);
    const int4 cw_s = slots[2 * i], cw_v = slots[2 * i + 1];
    acc = add(acc, Group::From(WithCtl(g[2 * side + 1], false)));
    if (t) acc = add(acc, Group::From(WithCtl(cw_v, false)));
    int4 child = WithCtl(g[2 * side], false);
    bool ct = Ctl(g[2 * side]);
    if (t) {
      child = X4(child, WithCtl(cw_s, false));
      ct ^= side ? Ctl(cw_v) : Ctl(cw_s);
    }
    s = child;
    t = ct;
  }
  acc = add(acc, Group::From(s));
  if (t) acc = add(acc, Group::From(slots[2 * in_bits + 1]));
  return acc.Into();
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

template <int in_bits, typename Group, typename In, typename Prg>
__device__ inline int4 HtEval(const Prg& prg, int4 hash_key, bool b, int4 s0, const int4* slots, int4 ocw, In x) {
  int4 node = WithCtl(s0, b);
  for (int i = 0; i < in_bits - 1; ++i) {
    const int4 h = HtHash(prg, hash_key, node);
    node = X4(X4(h, If(BitAt(x, in_bits - 1 - i), node)), If(Ctl(node), slots[2 * i]));
  }
  const bool sigma = BitAt(x, 0), t = Ctl(node);
  const int4 last = slots[2 * (in_bits - 1)];
  const bool lcw = sigma ? (slots[2 * (in_bits - 1) + 1].x & 0xff) != 0 : Ctl(last);
  const int4 h = HtHash(prg, hash_key, WithCtl(node, sigma));
  const int4 high = X4(WithCtl(h, false), If(t, WithCtl(last, false)));
  const bool low = Ctl(h) ^ (t && lcw);
  Group y = Group::From(high);
  if (low) y = y + Group::From(ocw);
  if (b) y = -y;
  return y.Into();
}

}  // namespace thread
}  // namespace fss
