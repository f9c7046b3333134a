// host_kat_test.cpp -- the HOST side of include/fss_amd.hpp's software AES (no GPU needed; runs under -m "not gpu"):
// the reference's aes_detail helpers and fss::prg::Aes128Soft / Aes128SoftBanked called on the host, against the
// constants the reference's own tests hold (Te0 / S-box / Rcon values, the PRG known answers) and FIPS-197
// appendix C.1.  The same cipher code (fss_amd_thread.hpp: AesEncrypt, AesLookups, AesLastRound on plain tables) is what
// the per-thread device path instantiates with its LDS tables (tests/cxx/dropin_device_test.cpp checks that one).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "fss_amd.hpp"

static int g_fail = 0;
#define EXPECT(c)                                                     \
  do {                                                                \
    if (!(c)) {                                                       \
      std::fprintf(stderr, "%s:%d: FAILED: %s\n", __FILE__, __LINE__, #c); \
      ++g_fail;                                                       \
    }                                                                 \
  } while (0)
static bool Same(int4 a, int4 b) { return std::memcmp(&a, &b, sizeof a) == 0; }
static int4 I4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return int4{(int)x, (int)y, (int)z, (int)w}; }
namespace ad = fss::prg::aes_detail;

int main() {
  // ---- the helpers' values (prg/aes128_mmo_soft.cuh:32-120)
  EXPECT(ad::ComputeTe0(0) == 0xc66363a5u && ad::ComputeTe0(1) == 0xf87c7c84u);
  EXPECT(ad::Sbox(0x53) == 0xed && ad::Sbox(0) == 0x63);
  EXPECT(ad::Rcon(0) == 0x8d && ad::Rcon(1) == 0x01 && ad::Rcon(9) == 0x1b && ad::Rcon(10) == 0x36);
  EXPECT(ad::RotWord8(0x11223344u) == 0x44112233u && ad::RotWord16(0x11223344u) == 0x33441122u &&
         ad::RotWord24(0x11223344u) == 0x22334411u);
  static_assert(ad::kRoundKeySize == 176 && ad::kNr == 10 && ad::kNb == 4 && ad::kNk == 4);
  uint32_t te0[256];
  uint8_t sbox[256];
  ad::InitTe0(te0);
  ad::InitSbox(sbox);
  for (int i = 0; i < 256; ++i) EXPECT(te0[i] == ad::ComputeTe0(static_cast<uint8_t>(i)) && sbox[i] == ad::Sbox(static_cast<uint8_t>(i)));
  {  // FIPS-197 appendix C.1 through KeyExpansion + Encrypt, and appendix A.1's last round key
    uint8_t key[16], pt[16], rk[ad::kRoundKeySize];
    for (int i = 0; i < 16; ++i) { key[i] = static_cast<uint8_t>(i); pt[i] = static_cast<uint8_t>(i * 0x11); }
    ad::KeyExpansion(rk, key, sbox);
    ad::Encrypt(pt, rk, te0, sbox);
    const uint8_t want[16] = {0x69, 0xc4, 0xe0, 0xd8, 0x6a, 0x7b, 0x04, 0x30, 0xd8, 0xcd, 0xb7, 0x80, 0x70, 0xb4, 0xc5, 0x5a};
    EXPECT(std::memcmp(pt, want, 16) == 0);
    EXPECT(rk[160] == 0x13 && rk[161] == 0x11 && rk[162] == 0x1d && rk[163] == 0x7f && rk[174] == 0x30 && rk[175] == 0xc5);
  }
  // ---- the PRG known answers, every mul, Gen and GenSide, plain and bank-replicated tables
  const uint8_t hkeys[4][16] = {{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16},
                                {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1},
                                {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8},
                                {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1}};
  const int4 seed = I4(0x11111111, 0x22222222, 0x33333333, 0x44444440);
  const int4 kat[4] = {I4(0x1423e6d2, 0x60533602, 0x813b3fbc, 0x412b31dc), I4(0x2d18cbe7, 0xa5eddcc9, 0xc691e4f6, 0x97831704),
                       I4(0x2c046c24, 0x8a033811, 0xcb0335b6, 0x3d799ddb), I4(0x2413b69f, 0xffed7ac4, 0x314804c9, 0x595d2580)};
  {
    fss::prg::Aes128Soft<4> prg(hkeys, te0, sbox);
    const auto o = prg.Gen(seed);
    for (int i = 0; i < 4; ++i) EXPECT(Same(o[i], kat[i]));
    const auto l = prg.GenSide(seed, false), r = prg.GenSide(seed, true);
    EXPECT(Same(l[0], kat[0]) && Same(l[1], kat[1]) && Same(r[0], kat[2]) && Same(r[1], kat[3]));
    fss::prg::Aes128Soft<2> prg2(hkeys, te0, sbox);
    const auto o2 = prg2.Gen(seed);
    EXPECT(Same(o2[0], kat[0]) && Same(o2[1], kat[1]));
    EXPECT(Same(prg2.GenSide(seed, true)[0], kat[1]) && Same(prg2.GenSide(seed, false)[0], kat[0]));
    fss::prg::Aes128Soft<1> prg1(hkeys, te0, sbox);
    EXPECT(Same(prg1.Gen(seed)[0], kat[0]));
  }
  {
    std::vector<uint32_t> banked(ad::kBankedTe0Words);
    ad::InitTe0Banked(banked.data(), 0, 1);
    for (int i = 0; i < 256; ++i) EXPECT(banked[32 * i] == te0[i] && banked[32 * i + 31] == te0[i]);
    fss::prg::Aes128SoftBanked<4> prg(hkeys, banked.data(), sbox);
    const auto o = prg.Gen(seed);
    for (int i = 0; i < 4; ++i) EXPECT(Same(o[i], kat[i]));
    const auto r = prg.GenSide(seed, true);
    EXPECT(Same(r[0], kat[2]) && Same(r[1], kat[3]));
    fss::prg::Aes128SoftBanked<2> prg2(hkeys, banked.data(), sbox);
    EXPECT(Same(prg2.Gen(seed)[1], kat[1]) && Same(prg2.GenSide(seed, false)[0], kat[0]));
  }
  {  // a second seed and the MMO feed-forward: AES_k(s) ^ s from the byte-level Encrypt equals Gen
    const int4 s2 = I4(0x01234567, 0x89abcdef, 0xfedcba98, 0x76543210);
    uint8_t rk[ad::kRoundKeySize], buf[16];
    fss::prg::Aes128Soft<2> prg2(hkeys, te0, sbox);
    const auto o = prg2.Gen(s2);
    for (int i = 0; i < 2; ++i) {
      ad::KeyExpansion(rk, hkeys[i], sbox);
      std::memcpy(buf, &s2, 16);
      ad::Encrypt(buf, rk, te0, sbox);
      int4 c;
      std::memcpy(&c, buf, 16);
      EXPECT(Same(o[i], int4{c.x ^ s2.x, c.y ^ s2.y, c.z ^ s2.z, c.w ^ s2.w}));
    }
  }
  if (g_fail) {
    std::fprintf(stderr, "%d check(s) FAILED\n", g_fail);
    return 1;
  }
  std::printf("OK\n");
  return 0;
}
