// The reference's own test list, case by case, run against this library.
//
// myl7/fss ships six gtest files (src/{group,dpf,dcf,half_tree_dpf,grotto_dcf,vdpf}_test.cu):
// group axioms over eleven group types, and per scheme the FSS property -- the two parties' shares reconstruct beta at
// alpha (below alpha for the comparison functions) and the identity elsewhere -- at alpha, at random other points and over
// the whole 2^16 domain, for ChaCha and AES PRGs, plus Grotto's parity-tree / EvalAll agreement and its alpha = 0 /
// alpha = N - 1 edges, Half-Tree's one-bit domain, and the verifiable DPF's proofs.  They hold no vectors: seeds are
// random per run (src/dpf_test.cu:49-56).
//
// This file is that list under the same suite and case names (`Suite.Case`, typed suites as `Suite/<index>.Case`), each
// case written here against the reference's spellings: the include lines <fss/...cuh>, the seven scheme templates, public
// aggregate initialisation `Scheme{prg}` / `{prg, hash_key}` / `{prg, xor_hash, hash}`, `Gen / Eval / EvalAll /
// Preprocess / Prove / Verify` on host arrays (cuda::std::array / span are std::array / span here).  The host forms stage
// onto the GPU (the C ABI of include/fss_amd.h), so a green run says: a user who swaps the reference for this library
// under their own test suite sees it pass.  gtest is not in the image: the thirty-line registry below stands in for it.
//
//   reference_cases_test [--seed S] [--filter substring] [--list]
// Exit code 0 and a final "OK" line when every case passed; the seed of the run is printed (random unless given).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include <fss/dcf.cuh>
#include <fss/dpf.cuh>
#include <fss/grotto_dcf.cuh>
#include <fss/group/bytes.cuh>
#include <fss/group/uint.cuh>
#include <fss/half_tree_dpf.cuh>
#include <fss/hash/blake3.cuh>
#include <fss/prg/aes128_mmo.cuh>
#include <fss/prg/aes128_mmo_soft.cuh>
#include <fss/prg/chacha.cuh>
#include <fss/vdpf.cuh>

// ---------------------------------------------------------------------------------------------- the registry
namespace {
struct Case {
  std::string name;
  std::function<void()> run;
};
std::vector<Case>& Cases() {
  static std::vector<Case> v;
  return v;
}
int g_checks = 0, g_failed_checks = 0;
std::mt19937 g_rng;

void Expect(bool ok, const char* what, long long at = -1) {
  ++g_checks;
  if (ok) return;
  if (++g_failed_checks <= 20) {
    if (at >= 0) std::fprintf(stderr, "    expectation failed: %s (x = %lld)\n", what, at);
    else std::fprintf(stderr, "    expectation failed: %s\n", what);
  }
}
struct Add {
  Add(std::string name, std::function<void()> fn) { Cases().push_back({std::move(name), std::move(fn)}); }
};

// ---------------------------------------------------------------------------------------------- what every suite shares
using BytesGroup = fss::group::Bytes;
using Uint64Group = fss::group::Uint<uint64_t>;
using Uint127Group = fss::group::Uint<__uint128_t, (static_cast<__uint128_t>(1) << 127)>;

constexpr uint16_t kAlpha = 107;  // src/dpf_test.cu:30-32 and the same three lines of every scheme's file
constexpr int kAlphaBits = 16;
constexpr __uint128_t kBeta = 604;
int g_nonce[2] = {0x12345678, static_cast<int>(0x9abcdef0u)};  // the ChaCha nonce of the reference's fixtures

int RandInt() { return static_cast<int>(g_rng()); }
int4 ClampedSeed() { return int4{RandInt(), RandInt(), RandInt(), RandInt() & ~1}; }
int4 BetaBuf() {
  return int4{static_cast<int>(kBeta & 0xffffffffu), static_cast<int>((kBeta >> 32) & 0xffffffffu), static_cast<int>((kBeta >> 64) & 0xffffffffu),
              static_cast<int>((kBeta >> 96) & 0xfffffffeu)};
}
uint16_t RandIn(uint16_t lo, uint16_t hi) { return std::uniform_int_distribution<uint16_t>(lo, hi)(g_rng); }

// how many bytes of a share buffer the group defines: sizeof(T) for Uint<T> below 128 bits, all sixteen otherwise
template <typename G>
struct Defined {
  static constexpr size_t bytes = sizeof(int4);
};
template <typename T, T mod>
struct Defined<fss::group::Uint<T, mod>> {
  static constexpr size_t bytes = sizeof(T) == 16 ? sizeof(int4) : sizeof(T);
};
template <typename G>
bool SameElement(int4 a, int4 b) {
  return std::memcmp(&a, &b, Defined<G>::bytes) == 0;
}
template <typename G>
int4 Reconstruct(int4 y0, int4 y1) {
  return (G::From(y0) + G::From(y1)).Into();
}
const int4 kZero = {0, 0, 0, 0};

// ---------------------------------------------------------------------------------------------- src/group_test.cu
template <typename G>
struct GroupAxioms {
  static constexpr int kTrials = 100;
  static G Random() { return G::From(ClampedSeed()); }
  static void Equal(const G& l, const G& r, const char* what) { Expect(SameElement<G>(l.Into(), r.Into()), what); }
  static void Associativity() {
    for (int i = 0; i < kTrials; ++i) {
      const G a = Random(), b = Random(), c = Random();
      Equal((a + b) + c, a + (b + c), "(a + b) + c == a + (b + c)");
    }
  }
  static void Identity() {
    const G zero;
    for (int i = 0; i < kTrials; ++i) {
      const G a = Random();
      Equal(a + zero, a, "a + 0 == a");
      Equal(zero + a, a, "0 + a == a");
    }
  }
  static void Inverses() {
    const G zero;
    for (int i = 0; i < kTrials; ++i) {
      const G a = Random();
      Equal(a + (-a), zero, "a + (-a) == 0");
      Equal((-a) + a, zero, "(-a) + a == 0");
    }
  }
};
template <typename G>
void AddGroupAxioms(int index) {
  const std::string suite = "GroupAxiomTest/" + std::to_string(index);
  Cases().push_back({suite + ".Associativity", &GroupAxioms<G>::Associativity});
  Cases().push_back({suite + ".Identity", &GroupAxioms<G>::Identity});
  Cases().push_back({suite + ".Inverses", &GroupAxioms<G>::Inverses});
}
const int group_axioms_added = [] {  // the eleven types of src/group_test.cu:9-19, in that order
  int i = 0;
  AddGroupAxioms<BytesGroup>(i++);
  AddGroupAxioms<fss::group::Uint<uint8_t>>(i++);
  AddGroupAxioms<fss::group::Uint<uint8_t, 131>>(i++);
  AddGroupAxioms<fss::group::Uint<uint16_t>>(i++);
  AddGroupAxioms<fss::group::Uint<uint16_t, 32771>>(i++);
  AddGroupAxioms<fss::group::Uint<uint32_t>>(i++);
  AddGroupAxioms<fss::group::Uint<uint32_t, 2147483659U>>(i++);
  AddGroupAxioms<Uint64Group>(i++);
  AddGroupAxioms<fss::group::Uint<uint64_t, 9223372036854775837ULL>>(i++);
  AddGroupAxioms<Uint127Group>(i++);
  AddGroupAxioms<fss::group::Uint<__uint128_t, (static_cast<__uint128_t>(1) << 126)>>(i++);
  return i;
}();

// ---------------------------------------------------------------------------------------------- src/dpf_test.cu
// One body per case, taking the PRG: the ChaCha suites construct it from the nonce, the AES suite from CreateCtxs.
template <typename Group, typename Prg>
struct DpfCases {
  using DpfType = fss::Dpf<kAlphaBits, Group, Prg, uint16_t>;
  int4 s0s[2] = {ClampedSeed(), ClampedSeed()};
  typename DpfType::Cw cws[kAlphaBits + 1];
  DpfType dpf;
  explicit DpfCases(Prg prg) : dpf{prg} { dpf.Gen(cws, s0s, kAlpha, BetaBuf()); }

  void EvalAtAlpha() {
    const int4 y0 = dpf.Eval(false, s0s[0], cws, kAlpha), y1 = dpf.Eval(true, s0s[1], cws, kAlpha);
    Expect(SameElement<Group>(Reconstruct<Group>(y0, y1), Group::From(BetaBuf()).Into()), "shares at alpha add up to beta");
  }
  void EvalAtNonAlpha() {
    for (int i = 0; i < 100; ++i) {
      const uint16_t x = RandIn(0, 0xffff);
      if (x == kAlpha) continue;
      const int4 y0 = dpf.Eval(false, s0s[0], cws, x), y1 = dpf.Eval(true, s0s[1], cws, x);
      Expect(SameElement<Group>(Reconstruct<Group>(y0, y1), kZero), "shares off alpha add up to the identity", x);
    }
  }
  void EvalAll() {
    constexpr size_t n = size_t(1) << kAlphaBits;
    std::vector<int4> ys0(n), ys1(n);
    dpf.EvalAll(false, s0s[0], cws, ys0.data());
    dpf.EvalAll(true, s0s[1], cws, ys1.data());
    const int4 beta = Group::From(BetaBuf()).Into();
    for (size_t x = 0; x < n; ++x)
      Expect(SameElement<Group>(Reconstruct<Group>(ys0[x], ys1[x]), x == kAlpha ? beta : kZero), "full-domain shares reconstruct the point function", (long long)x);
  }
};
template <typename Group>
void AddDpfChaCha(const char* suite) {
  using P = fss::prg::ChaCha<2>;
  using C = DpfCases<Group, P>;
  const std::string s = suite;
  Cases().push_back({s + ".EvalAtAlpha", [] { C(P(g_nonce)).EvalAtAlpha(); }});
  Cases().push_back({s + ".EvalAtNonAlpha", [] { C(P(g_nonce)).EvalAtNonAlpha(); }});
  Cases().push_back({s + ".EvalAll", [] { C(P(g_nonce)).EvalAll(); }});
}
unsigned char kAesKey0[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
unsigned char kAesKey1[16] = {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1};
unsigned char kAesKey2[16] = {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8};
unsigned char kAesKey3[16] = {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1};
// SetUp / TearDown of DpfBytesAesTest (src/dpf_test.cu:158-186): contexts from two keys, the PRG over them, freed afterwards
template <typename Fn>
void WithAes2(Fn fn) {
  using P = fss::prg::Aes128Mmo<2>;
  const unsigned char* keys[2] = {kAesKey0, kAesKey1};
  std::array<EVP_CIPHER_CTX*, 2> ctxs = P::CreateCtxs(keys);  // the fixture's member, spelled as src/dpf_test.cu:164 spells it
  {
    P prg(ctxs);
    fn(prg);
  }
  P::FreeCtxs(ctxs);
}
const int dpf_added = [] {
  AddDpfChaCha<BytesGroup>("DpfBytesChaChaTest");
  AddDpfChaCha<Uint127Group>("DpfUint128ChaChaTest");
  AddDpfChaCha<Uint64Group>("DpfUint64ChaChaTest");
  using C = DpfCases<BytesGroup, fss::prg::Aes128Mmo<2>>;
  Cases().push_back({"DpfBytesAesTest.EvalAtAlpha", [] { WithAes2([](auto prg) { C(prg).EvalAtAlpha(); }); }});
  Cases().push_back({"DpfBytesAesTest.EvalAtNonAlpha", [] { WithAes2([](auto prg) { C(prg).EvalAtNonAlpha(); }); }});
  return 0;
}();

// ---------------------------------------------------------------------------------------------- src/dcf_test.cu
template <typename Group, typename Prg>
struct DcfCases {
  using DcfType = fss::Dcf<kAlphaBits, Group, Prg, uint16_t>;  // the default predicate: x < alpha
  int4 s0s[2] = {ClampedSeed(), ClampedSeed()};
  typename DcfType::Cw cws[kAlphaBits + 1];
  DcfType dcf;
  explicit DcfCases(Prg prg) : dcf{prg} { dcf.Gen(cws, s0s, kAlpha, BetaBuf()); }

  void EvalLt() {
    const int4 beta = Group::From(BetaBuf()).Into();
    for (int i = 0; i < 50; ++i) {  // below alpha: beta
      const uint16_t x = RandIn(0, kAlpha - 1);
      const int4 y0 = dcf.Eval(false, s0s[0], cws, x), y1 = dcf.Eval(true, s0s[1], cws, x);
      Expect(SameElement<Group>(Reconstruct<Group>(y0, y1), beta), "shares below alpha add up to beta", x);
    }
    for (int i = 0; i < 50; ++i) {  // from alpha on: the identity
      const uint16_t x = RandIn(kAlpha, 0xffff);
      const int4 y0 = dcf.Eval(false, s0s[0], cws, x), y1 = dcf.Eval(true, s0s[1], cws, x);
      Expect(SameElement<Group>(Reconstruct<Group>(y0, y1), kZero), "shares from alpha on add up to the identity", x);
    }
  }
  void EvalAll() {
    constexpr size_t n = size_t(1) << kAlphaBits;
    std::vector<int4> ys0(n), ys1(n);
    dcf.EvalAll(false, s0s[0], cws, ys0.data());
    dcf.EvalAll(true, s0s[1], cws, ys1.data());
    const int4 beta = Group::From(BetaBuf()).Into();
    for (size_t x = 0; x < n; ++x)
      Expect(SameElement<Group>(Reconstruct<Group>(ys0[x], ys1[x]), x < kAlpha ? beta : kZero), "full-domain shares reconstruct the comparison function", (long long)x);
  }
};
template <typename Group>
void AddDcfChaCha(const char* suite) {
  using P = fss::prg::ChaCha<4>;
  using C = DcfCases<Group, P>;
  const std::string s = suite;
  Cases().push_back({s + ".EvalLt", [] { C(P(g_nonce)).EvalLt(); }});
  Cases().push_back({s + ".EvalAll", [] { C(P(g_nonce)).EvalAll(); }});
}
const int dcf_added = [] {
  AddDcfChaCha<BytesGroup>("DcfBytesChaChaTest");
  AddDcfChaCha<Uint127Group>("DcfUint128ChaChaTest");
  AddDcfChaCha<Uint64Group>("DcfUint64ChaChaTest");
  Cases().push_back({"DcfBytesAesTest.EvalLt", [] {  // four keys: src/dcf_test.cu:166-171
                       using P = fss::prg::Aes128Mmo<4>;
                       const unsigned char* keys[4] = {kAesKey0, kAesKey1, kAesKey2, kAesKey3};
                       std::array<EVP_CIPHER_CTX*, 4> ctxs = P::CreateCtxs(keys);
                       {
                         P prg(ctxs);
                         DcfCases<BytesGroup, P>(prg).EvalLt();
                       }
                       P::FreeCtxs(ctxs);
                     }});
  return 0;
}();

// ---------------------------------------------------------------------------------------------- src/half_tree_dpf_test.cu
// Three fixtures x three groups x three cases: ChaCha<1> on 16 bits, Aes128Soft<1> on 16 bits (tables and key built by the
// caller, src/half_tree_dpf_test.cu:205-211), ChaCha<1> on the one-bit domain with alpha = 1.
template <int in_bits, typename Group, typename Prg, typename In>
struct HalfTreeCases {
  using DpfType = fss::HalfTreeDpf<in_bits, Group, Prg, In>;
  static constexpr In kA = in_bits == 1 ? In(1) : In(kAlpha);
  int4 s0s[2] = {ClampedSeed(), ClampedSeed()};
  int4 hash_key = {RandInt(), RandInt(), RandInt(), RandInt()};
  typename DpfType::Cw cws[in_bits];
  int4 ocw;
  DpfType dpf;
  explicit HalfTreeCases(Prg prg) : dpf{prg, hash_key} { dpf.Gen(cws, ocw, s0s, kA, BetaBuf()); }

  void EvalAtAlpha() {
    const int4 y0 = dpf.Eval(false, s0s[0], cws, ocw, kA), y1 = dpf.Eval(true, s0s[1], cws, ocw, kA);
    Expect(SameElement<Group>(Reconstruct<Group>(y0, y1), Group::From(BetaBuf()).Into()), "shares at alpha add up to beta");
  }
  void EvalAtNonAlpha() {
    if constexpr (in_bits == 1) {
      const int4 y0 = dpf.Eval(false, s0s[0], cws, ocw, In(0)), y1 = dpf.Eval(true, s0s[1], cws, ocw, In(0));
      Expect(SameElement<Group>(Reconstruct<Group>(y0, y1), kZero), "the other point of the one-bit domain reconstructs the identity");
    } else {
      for (int i = 0; i < 100; ++i) {
        const In x = RandIn(0, 0xffff);
        if (x == kA) continue;
        const int4 y0 = dpf.Eval(false, s0s[0], cws, ocw, x), y1 = dpf.Eval(true, s0s[1], cws, ocw, x);
        Expect(SameElement<Group>(Reconstruct<Group>(y0, y1), kZero), "shares off alpha add up to the identity", x);
      }
    }
  }
  void EvalAll() {
    constexpr size_t n = size_t(1) << in_bits;
    std::vector<int4> ys0(n), ys1(n);
    dpf.EvalAll(false, s0s[0], cws, ocw, ys0.data());
    dpf.EvalAll(true, s0s[1], cws, ocw, ys1.data());
    const int4 beta = Group::From(BetaBuf()).Into();
    for (size_t x = 0; x < n; ++x)
      Expect(SameElement<Group>(Reconstruct<Group>(ys0[x], ys1[x]), x == kA ? beta : kZero), "full-domain shares reconstruct the point function", (long long)x);
  }
};
template <int in_bits, typename Group, typename In>
void AddHalfTreeChaCha(const char* suite) {
  using P = fss::prg::ChaCha<1>;
  using C = HalfTreeCases<in_bits, Group, P, In>;
  const std::string s = suite;
  Cases().push_back({s + ".EvalAtAlpha", [] { C(P(g_nonce)).EvalAtAlpha(); }});
  Cases().push_back({s + ".EvalAtNonAlpha", [] { C(P(g_nonce)).EvalAtNonAlpha(); }});
  Cases().push_back({s + ".EvalAll", [] { C(P(g_nonce)).EvalAll(); }});
}
template <typename Fn>
void WithAesSoft1(Fn fn) {
  static uint8_t aes_key[1][16] = {{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16}};
  uint32_t te0[256];
  uint8_t sbox[256];
  fss::prg::aes_detail::InitTe0(te0);
  fss::prg::aes_detail::InitSbox(sbox);
  fss::prg::Aes128Soft<1> prg(aes_key, te0, sbox);
  fn(prg);
}
template <typename Group>
void AddHalfTreeAesSoft(const char* suite) {
  using C = HalfTreeCases<kAlphaBits, Group, fss::prg::Aes128Soft<1>, uint16_t>;
  const std::string s = suite;
  Cases().push_back({s + ".EvalAtAlpha", [] { WithAesSoft1([](auto prg) { C(prg).EvalAtAlpha(); }); }});
  Cases().push_back({s + ".EvalAtNonAlpha", [] { WithAesSoft1([](auto prg) { C(prg).EvalAtNonAlpha(); }); }});
  Cases().push_back({s + ".EvalAll", [] { WithAesSoft1([](auto prg) { C(prg).EvalAll(); }); }});
}
const int half_tree_added = [] {
  AddHalfTreeChaCha<kAlphaBits, BytesGroup, uint16_t>("HalfTreeDpfBytesChaChaTest");
  AddHalfTreeChaCha<kAlphaBits, Uint127Group, uint16_t>("HalfTreeDpfUint128ChaChaTest");
  AddHalfTreeChaCha<kAlphaBits, Uint64Group, uint16_t>("HalfTreeDpfUint64ChaChaTest");
  AddHalfTreeAesSoft<BytesGroup>("HalfTreeDpfBytesAesSoftTest");
  AddHalfTreeAesSoft<Uint127Group>("HalfTreeDpfUint128AesSoftTest");
  AddHalfTreeAesSoft<Uint64Group>("HalfTreeDpfUint64AesSoftTest");
  AddHalfTreeChaCha<1, BytesGroup, uint8_t>("HalfTreeDpfBytesOneBitTest");
  AddHalfTreeChaCha<1, Uint127Group, uint8_t>("HalfTreeDpfUint128OneBitTest");
  AddHalfTreeChaCha<1, Uint64Group, uint8_t>("HalfTreeDpfUint64OneBitTest");
  return 0;
}();

// ---------------------------------------------------------------------------------------------- src/grotto_dcf_test.cu
// Shares are single bits; the function is 1[alpha <= x].
struct GrottoCases {
  using GrottoDcfType = fss::GrottoDcf<kAlphaBits, fss::prg::ChaCha<2>, uint16_t>;
  static constexpr size_t N = size_t(1) << kAlphaBits;
  int4 s0s[2] = {ClampedSeed(), ClampedSeed()};
  GrottoDcfType::Cw cws[kAlphaBits + 1];
  fss::prg::ChaCha<2> prg{g_nonce};
  GrottoDcfType dcf{prg};
  GrottoCases() { dcf.Gen(cws, s0s, kAlpha); }

  void PreprocessEvalAll() {
    auto p0 = std::make_unique<bool[]>(2 * N - 1), p1 = std::make_unique<bool[]>(2 * N - 1);
    GrottoDcfType::ParityTree pt0{p0.get(), false}, pt1{p1.get(), true};
    dcf.Preprocess(pt0, s0s[0], cws);
    dcf.Preprocess(pt1, s0s[1], cws);
    for (size_t x = 0; x < N; ++x) {
      const bool got = GrottoDcfType::Eval(pt0, static_cast<uint16_t>(x)) ^ GrottoDcfType::Eval(pt1, static_cast<uint16_t>(x));
      Expect(got == (kAlpha <= x), "parity-tree shares XOR to 1[alpha <= x]", (long long)x);
    }
  }
  void XorIs(const GrottoDcfType::Cw* k, std::function<bool(size_t)> want, const char* what) {
    auto ys0 = std::make_unique<bool[]>(N), ys1 = std::make_unique<bool[]>(N);
    dcf.EvalAll(false, s0s[0], k, ys0.get());
    dcf.EvalAll(true, s0s[1], k, ys1.get());
    for (size_t x = 0; x < N; ++x) Expect((ys0[x] ^ ys1[x]) == want(x), what, (long long)x);
  }
  void EvalAll() { XorIs(cws, [](size_t x) { return kAlpha <= x; }, "full-domain shares XOR to 1[alpha <= x]"); }
  void PreprocessEvalMatchesEvalAll() {
    auto p0 = std::make_unique<bool[]>(2 * N - 1);
    GrottoDcfType::ParityTree pt0{p0.get(), false};
    dcf.Preprocess(pt0, s0s[0], cws);
    auto ys0 = std::make_unique<bool[]>(N);
    dcf.EvalAll(false, s0s[0], cws, ys0.get());
    for (size_t x = 0; x < N; ++x)
      Expect(GrottoDcfType::Eval(pt0, static_cast<uint16_t>(x)) == ys0[x], "party 0: parity-tree share == full-domain share", (long long)x);
  }
  void EdgeAlphaZero() {
    GrottoDcfType::Cw edge[kAlphaBits + 1];
    dcf.Gen(edge, s0s, static_cast<uint16_t>(0));
    XorIs(edge, [](size_t) { return true; }, "alpha = 0: every x compares true");
  }
  void EdgeAlphaMax() {
    GrottoDcfType::Cw edge[kAlphaBits + 1];
    dcf.Gen(edge, s0s, static_cast<uint16_t>(N - 1));
    XorIs(edge, [](size_t x) { return x == N - 1; }, "alpha = N - 1: only the last x compares true");
  }
};
const int grotto_added = [] {
  Cases().push_back({"GrottoDcfChaChaTest.PreprocessEvalAll", [] { GrottoCases().PreprocessEvalAll(); }});
  Cases().push_back({"GrottoDcfChaChaTest.EvalAll", [] { GrottoCases().EvalAll(); }});
  Cases().push_back({"GrottoDcfChaChaTest.PreprocessEvalMatchesEvalAll", [] { GrottoCases().PreprocessEvalMatchesEvalAll(); }});
  Cases().push_back({"GrottoDcfChaChaTest.EdgeAlphaZero", [] { GrottoCases().EdgeAlphaZero(); }});
  Cases().push_back({"GrottoDcfChaChaTest.EdgeAlphaMax", [] { GrottoCases().EdgeAlphaMax(); }});
  return 0;
}();

// ---------------------------------------------------------------------------------------------- src/vdpf_test.cu
// Key generation may ask for fresh seeds (a non-zero return: the two parties' hashes of alpha collided on the control
// bits, vdpf.cuh:103-193); the fixture draws again, as the reference's does.  Shares compare over all sixteen bytes.
template <typename Group>
struct VdpfCases {
  using Prg = fss::prg::ChaCha<2>;
  using H = fss::hash::Blake3;
  using VdpfType = fss::Vdpf<kAlphaBits, Group, Prg, H, H, uint16_t>;
  using Cws = std::span<const typename VdpfType::Cw>;
  using Cs = std::span<const int4, 4>;
  int4 s0s[2];
  typename VdpfType::Cw cws[kAlphaBits];
  std::array<int4, 4> cs;
  int4 ocw;
  int4 hash_iv[2] = {{0x11111111, 0x22222222, 0x33333333, 0x44444444}, {0x55555555, 0x66666666, 0x77777777, static_cast<int>(0x88888888u)}};
  VdpfType vdpf{Prg(g_nonce), H(std::span<const int4, 2>(hash_iv, 2)), H(std::span<const int4, 2>(hash_iv, 2))};
  VdpfCases() {
    int ret;
    do {
      s0s[0] = ClampedSeed();
      s0s[1] = ClampedSeed();
      ret = vdpf.Gen(cws, cs, ocw, std::span<const int4, 2>(s0s, 2), kAlpha, BetaBuf());
    } while (ret != 0);
  }
  static bool Same16(int4 a, int4 b) { return std::memcmp(&a, &b, sizeof a) == 0; }

  void EvalAtAlpha() {
    int4 y0, y1;
    vdpf.Eval(false, s0s[0], Cws(cws, kAlphaBits), Cs(cs), ocw, kAlpha, y0);
    vdpf.Eval(true, s0s[1], Cws(cws, kAlphaBits), Cs(cs), ocw, kAlpha, y1);
    Expect(Same16(Reconstruct<Group>(y0, y1), Group::From(BetaBuf()).Into()), "shares at alpha add up to beta");
  }
  void EvalAtNonAlpha() {
    for (int i = 0; i < 100; ++i) {
      const uint16_t x = RandIn(0, 0xffff);
      if (x == kAlpha) continue;
      int4 y0, y1;
      vdpf.Eval(false, s0s[0], Cws(cws, kAlphaBits), Cs(cs), ocw, x, y0);
      vdpf.Eval(true, s0s[1], Cws(cws, kAlphaBits), Cs(cs), ocw, x, y1);
      Expect(Same16(Reconstruct<Group>(y0, y1), kZero), "shares off alpha add up to the identity", x);
    }
  }
  void VerifyEval() {
    constexpr int L = 50;
    std::vector<std::array<int4, 4>> pt0(L), pt1(L);
    for (int i = 0; i < L; ++i) {
      const uint16_t x = RandIn(0, 0xffff);
      int4 y0, y1;
      pt0[i] = vdpf.Eval(false, s0s[0], Cws(cws, kAlphaBits), Cs(cs), ocw, x, y0);
      pt1[i] = vdpf.Eval(true, s0s[1], Cws(cws, kAlphaBits), Cs(cs), ocw, x, y1);
    }
    std::array<int4, 4> pi0, pi1;
    vdpf.Prove(std::span<const std::array<int4, 4>>(pt0), Cs(cs), pi0);
    vdpf.Prove(std::span<const std::array<int4, 4>>(pt1), Cs(cs), pi1);
    Expect(VdpfType::Verify(Cs(pi0), Cs(pi1)), "the two parties' proofs over fifty points verify");
  }
  void EvalAll() {
    constexpr size_t n = size_t(1) << kAlphaBits;
    std::vector<int4> ys0(n), ys1(n);
    std::array<int4, 4> pi0, pi1;
    vdpf.EvalAll(false, s0s[0], Cws(cws, kAlphaBits), Cs(cs), ocw, std::span<int4>(ys0), pi0);
    vdpf.EvalAll(true, s0s[1], Cws(cws, kAlphaBits), Cs(cs), ocw, std::span<int4>(ys1), pi1);
    const int4 beta = Group::From(BetaBuf()).Into();
    for (size_t x = 0; x < n; ++x)
      Expect(Same16(Reconstruct<Group>(ys0[x], ys1[x]), x == kAlpha ? beta : kZero), "full-domain shares reconstruct the point function", (long long)x);
    Expect(VdpfType::Verify(Cs(pi0), Cs(pi1)), "the full-domain proofs verify");
  }
};
template <typename Group>
void AddVdpf(const char* suite) {
  using C = VdpfCases<Group>;
  const std::string s = suite;
  Cases().push_back({s + ".EvalAtAlpha", [] { C().EvalAtAlpha(); }});
  Cases().push_back({s + ".EvalAtNonAlpha", [] { C().EvalAtNonAlpha(); }});
  Cases().push_back({s + ".VerifyEval", [] { C().VerifyEval(); }});
  Cases().push_back({s + ".EvalAll", [] { C().EvalAll(); }});
}
const int vdpf_added = [] {
  AddVdpf<BytesGroup>("VdpfBytesChaChaTest");
  AddVdpf<Uint127Group>("VdpfUint128ChaChaTest");
  return 0;
}();
}  // namespace

int main(int argc, char** argv) {
  uint32_t seed = std::random_device{}();
  std::string filter;
  bool list = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--seed" && i + 1 < argc) seed = static_cast<uint32_t>(std::strtoul(argv[++i], nullptr, 0));
    else if (a == "--filter" && i + 1 < argc) filter = argv[++i];
    else if (a == "--list") list = true;
    else {
      std::fprintf(stderr, "usage: %s [--seed S] [--filter substring] [--list]\n", argv[0]);
      return 2;
    }
  }
  if (list) {
    for (const Case& c : Cases()) std::printf("%s\n", c.name.c_str());
    return 0;
  }
  std::printf("seed %u\n", seed);
  int ran = 0, failed = 0;
  for (const Case& c : Cases()) {
    if (!filter.empty() && c.name.find(filter) == std::string::npos) continue;
    g_rng.seed(seed ^ static_cast<uint32_t>(std::hash<std::string>{}(c.name)));  // a case's inputs do not depend on which others ran
    const int before = g_failed_checks;
    std::string thrown;
    try {
      c.run();
    } catch (const std::exception& e) {
      thrown = e.what();
    }
    ++ran;
    const bool ok = thrown.empty() && g_failed_checks == before;
    if (!ok) ++failed;
    if (!thrown.empty()) std::fprintf(stderr, "    exception: %s\n", thrown.c_str());
    std::printf("[ %s ] %s\n", ok ? "      OK" : "  FAILED", c.name.c_str());
    std::fflush(stdout);
  }
  std::printf("%d cases, %d failed, %d expectations checked\n", ran, failed, g_checks);
  if (failed || ran == 0) return 1;
  std::printf("OK\n");
  return 0;
}
