// Drop-in check of include/fss_amd.hpp: caller code written the way myl7/fss's own callers write it must compile
// unchanged (apart from cuda* -> hip* runtime names and cuda::std -> std) and give the reference's results.
//   part 1  the CPU samples' flow (samples/dpf_dcf_cpu.cu:35-91, half_tree_dpf_cpu.cu, grotto_dcf_cpu.cu:29-82):
//           CreateCtxs -> Prg(ctxs) -> Scheme{prg} -> Gen / Eval / EvalAll on host arrays -> FreeCtxs, checked
//           against the reference's known answers (PRG, DPF n=8 Bytes, DCF n=8, DPF n=32 Uint<u64>,
//           Half-Tree n=64) and the FSS property;
//   part 2  one Eval per key with the call of samples/dpf_dcf_gpu.cu:58 -- dpf.Eval(party, seeds[i],
//           cws + i * (kInBits + 1), xs[i]) -- here from a host loop over device-resident keys;
//   part 3  the GPU launchers with the call syntax of src/bench_gpu.cu:578 (DpfEvalAllGpuBatch<17, 9, 256>),
//           :618 (DpfRelayoutGpu<in_bits, Group, Prg, uint> with uint32_t extra), :625 (DpfEvalPointGpu<1, ...>
//           with uint32_t xs), :653-661 (Dcf), :688-697 (HalfTree), plus DpfEvalAllGpu<16> as
//           check_evalall_gpu.cu:64-91 uses it, each compared with the host-signature methods and reconstructed.
//   part 4  the verifiable DPF: fss::Vdpf<in_bits, Group, Prg, XorHash, Hash, In> {prg, xor_hash, hash} with Gen / Eval /
//           Prove / Verify / EvalAll as src/vdpf_test.cu and samples/vdpf_cpu.cu call them, the per-thread device forms and
//           VdpfRelayoutGpu / VdpfEvalPointGpu as src/bench_gpu.cu:170-205,704-735; fss::util::ResolveParDepth.
// Exit code 0 and a final "OK" line on success.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "fss_amd.hpp"

#define HIP_OK(e)                                                                                        \
  do {                                                                                                   \
    hipError_t r_ = (e);                                                                                 \
    if (r_ != hipSuccess) {                                                                              \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(r_), __FILE__, __LINE__);        \
      std::exit(2);                                                                                      \
    }                                                                                                    \
  } while (0)

static int g_fail = 0;
#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) {                                                            \
      ++g_fail;                                                               \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
    }                                                                         \
  } while (0)

static bool Same(int4 a, int4 b) { return std::memcmp(&a, &b, sizeof a) == 0; }
static int4 I4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return int4{(int)x, (int)y, (int)z, (int)w}; }
template <typename G>
static int4 Recon(int4 y0, int4 y1) {
  return (G::From(y0) + G::From(y1)).Into();
}

static unsigned char k0[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static unsigned char k1[16] = {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1};
static unsigned char k2[16] = {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8};
static unsigned char k3[16] = {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1};
static const int4 kSeeds[2] = {{0x11111111, 0x22222222, 0x33333333, 0x44444440},
                               {0x55555555, 0x66666666, 0x77777777, static_cast<int>(0x88888880u)}};
static const int4 kZero = {0, 0, 0, 0};

// ---------------------------------------------------------------------------------------------- part 1
static void HostSignatureDpf() {
  constexpr int kInBits = 8;
  using In = uint8_t;
  using Group = fss::group::Bytes;
  using DpfPrg = fss::prg::Aes128Mmo<2>;
  using Dpf = fss::Dpf<kInBits, Group, DpfPrg, In>;

  const unsigned char* keys[2] = {k0, k1};
  auto ctxs = DpfPrg::CreateCtxs(keys);
  DpfPrg prg(ctxs);
  Dpf dpf{prg};

  In alpha = 42;
  int4 beta = {7, 0, 0, 0};
  int4 seeds[2] = {kSeeds[0], kSeeds[1]};
  Dpf::Cw cws[kInBits + 1];
  dpf.Gen(cws, seeds, alpha, beta);
  // reference known answers for exactly these inputs
  EXPECT(Same(cws[0].s, I4(0x84863059, 0xb57c0062, 0xc0a8015f, 0x3520c9e8)) && cws[0].tr == true);
  EXPECT(Same(cws[8].s, I4(0x0b17cf59, 0xcd54e225, 0xef2cd79f, 0x30ad9cf8)));

  int4 y0 = dpf.Eval(false, seeds[0], cws, alpha);
  int4 y1 = dpf.Eval(true, seeds[1], cws, alpha);
  EXPECT(Same(y0, I4(0xa81892ad, 0xad2e583e, 0x272d1483, 0xd2bfa094)));
  EXPECT(Same(y1, I4(0xa81892aa, 0xad2e583e, 0x272d1483, 0xd2bfa094)));
  EXPECT(Same(Recon<Group>(y0, y1), beta));
  In x = 100;
  y0 = dpf.Eval(false, seeds[0], cws, x);
  y1 = dpf.Eval(true, seeds[1], cws, x);
  EXPECT(Same(y0, I4(0x7e2be32f, 0x4508eb0c, 0x537f8e89, 0x39b1bd96)));
  EXPECT(Same(Recon<Group>(y0, y1), kZero));

  int4 ys0[1 << kInBits];
  int4 ys1[1 << kInBits];
  dpf.EvalAll(false, seeds[0], cws, ys0);
  dpf.EvalAll(true, seeds[1], cws, ys1);
  int bad = 0;
  for (int i = 0; i < (1 << kInBits); ++i) {
    if (!Same(Recon<Group>(ys0[i], ys1[i]), i == alpha ? beta : kZero)) ++bad;
    if (!Same(ys0[i], dpf.Eval(false, seeds[0], cws, (In)i))) ++bad;  // EvalAll == Eval, per party
  }
  EXPECT(bad == 0);

  // the PRG hook itself: Gen(int4) -> array<int4, mul>, the reference's outputs for this seed and these keys
  auto out = prg.Gen(kSeeds[0]);
  EXPECT(Same(out[0], I4(0x1423e6d2, 0x60533602, 0x813b3fbc, 0x412b31dc)));
  EXPECT(Same(out[1], I4(0x2d18cbe7, 0xa5eddcc9, 0xc691e4f6, 0x97831704)));
  {  // what one host-signature call costs (a batch of one on the GPU; the batched forms are the fast path)
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0));
    int4 acc = kZero;
    for (int i = 0; i < 200; ++i) acc = fss::util::Xor(acc, dpf.Eval(false, seeds[0], cws, (In)i));
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("     (host-signature Eval: %.1f us per call, checksum %08x)\n", ms * 1000.0f / 200, (unsigned)acc.x);
  }
  {  // the reference's other two spellings of the same PRG, host side: Aes128MmoRaw(keys) (aes128_mmo_raw.cuh:76) and
     // Aes128Soft(keys, te0, sbox) with host tables (aes128_mmo_soft.cuh:197-207) -- same outputs, same keys, same shares
    const uint8_t kk[2][16] = {{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16}, {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1}};
    fss::prg::Aes128MmoRaw<2> raw(kk);
    auto o = raw.Gen(kSeeds[0]);
    EXPECT(Same(o[0], out[0]) && Same(o[1], out[1]));
    fss::Dpf<kInBits, Group, fss::prg::Aes128MmoRaw<2>, In> rdpf{raw};
    EXPECT(Same(rdpf.Eval(false, seeds[0], reinterpret_cast<const decltype(rdpf)::Cw*>(cws), alpha), I4(0xa81892ad, 0xad2e583e, 0x272d1483, 0xd2bfa094)));
    uint32_t te0[256];
    uint8_t sbox[256];
    fss::prg::aes_detail::InitTe0(te0);
    fss::prg::aes_detail::InitSbox(sbox);
    EXPECT(te0[0] == 0xc66363a5u && te0[1] == 0xf87c7c84u);
    fss::prg::Aes128Soft<2> soft(kk, te0, sbox);
    auto so = soft.Gen(kSeeds[0]);
    EXPECT(Same(so[0], out[0]) && Same(so[1], out[1]));
    fss::Dpf<kInBits, Group, fss::prg::Aes128Soft<2>, In> sdpf{soft};
    decltype(sdpf)::Cw scws[kInBits + 1];
    sdpf.Gen(scws, seeds, alpha, beta);  // the engine, keyed from the Aes128Soft object
    EXPECT(std::memcmp(scws, cws, sizeof scws) == 0 || (Same(scws[0].s, cws[0].s) && Same(scws[8].s, cws[8].s)));
    EXPECT(Same(sdpf.Eval(true, seeds[1], scws, alpha), I4(0xa81892aa, 0xad2e583e, 0x272d1483, 0xd2bfa094)));
  }
  DpfPrg::FreeCtxs(ctxs);
  std::printf("DPF  n=8  Bytes      host-signature Gen/Eval/EvalAll + PRG KAT (Aes128Mmo, Aes128MmoRaw, Aes128Soft): %s\n", g_fail ? "FAILED" : "ok");
}

static void HostSignatureDcf() {
  constexpr int kInBits = 8;
  using In = uint8_t;
  using Group = fss::group::Bytes;
  using DcfPrg = fss::prg::Aes128Mmo<4>;
  using Dcf = fss::Dcf<kInBits, Group, DcfPrg, In>;
  const unsigned char* keys[4] = {k0, k1, k2, k3};
  auto ctxs = DcfPrg::CreateCtxs(keys);
  DcfPrg prg(ctxs);
  Dcf dcf{prg};
  int before = g_fail;

  In alpha = 42;
  int4 beta = {7, 0, 0, 0};
  int4 seeds[2] = {kSeeds[0], kSeeds[1]};
  Dcf::Cw cws[kInBits + 1];
  dcf.Gen(cws, seeds, alpha, beta);
  EXPECT(Same(cws[0].s, I4(0x79f28305, 0x7b1a2718, 0xc9874166, 0xfe3afd02)));
  EXPECT(Same(cws[0].v, I4(0x0b706acf, 0xe72e35b7, 0x1984be09, 0x3d5a09d1)));
  int4 y0 = dcf.Eval(false, seeds[0], cws, 10);
  int4 y1 = dcf.Eval(true, seeds[1], cws, 10);
  EXPECT(Same(y0, I4(0x3a28e8cb, 0x5f11990b, 0xf2fee90f, 0xa7604478)));
  EXPECT(Same(y1, I4(0x3a28e8cc, 0x5f11990b, 0xf2fee90f, 0xa7604478)));
  std::vector<int4> ys0(1 << kInBits), ys1(1 << kInBits);
  dcf.EvalAll(false, seeds[0], cws, ys0.data());
  dcf.EvalAll(true, seeds[1], cws, ys1.data());
  int bad = 0;
  for (int i = 0; i < (1 << kInBits); ++i)
    if (!Same(Recon<Group>(ys0[i], ys1[i]), i < alpha ? beta : kZero)) ++bad;
  EXPECT(bad == 0);
  auto out = prg.Gen(kSeeds[0]);
  EXPECT(Same(out[2], I4(0x2c046c24, 0x8a033811, 0xcb0335b6, 0x3d799ddb)));
  EXPECT(Same(out[3], I4(0x2413b69f, 0xffed7ac4, 0x314804c9, 0x595d2580)));

  // kGt and a wrapping integer group
  using DcfGt = fss::Dcf<16, fss::group::Uint<uint32_t>, DcfPrg, uint16_t, fss::DcfPred::kGt>;
  DcfGt gt{prg};
  DcfGt::Cw gcws[17];
  int4 b32 = {1234567, 0, 0, 0};
  gt.Gen(gcws, seeds, (uint16_t)30000, b32);
  for (uint16_t xx : {(uint16_t)0, (uint16_t)29999, (uint16_t)30000, (uint16_t)30001, (uint16_t)65535}) {
    int4 s = Recon<fss::group::Uint<uint32_t>>(gt.Eval(false, seeds[0], gcws, xx), gt.Eval(true, seeds[1], gcws, xx));
    EXPECT(Same(s, xx > 30000 ? b32 : kZero));
  }
  {  // the handles are spelled EVP_CIPHER_CTX* as in the reference; a pointer that CreateCtxs did not hand out is refused
    std::array<EVP_CIPHER_CTX*, 4> same_type = ctxs;
    static unsigned char not_ours[64] = {};
    same_type[2] = reinterpret_cast<EVP_CIPHER_CTX*>(not_ours);
    bool refused = false;
    try {
      DcfPrg bad(same_type);
    } catch (const fss::Error& e) {
      refused = e.code == FSS_AMD_EINVAL;
    }
    EXPECT(refused);
  }
  DcfPrg::FreeCtxs(ctxs);
  std::printf("DCF  n=8  Bytes, n=16 Uint<u32> kGt   host-signature: %s\n", g_fail == before ? "ok" : "FAILED");
}

// Uint<__uint128_t, mod> with a modulus other than 2^127 (group/uint.cuh:28-30 allows 0 < mod <= 2^127; src/group_test.cu:18
// uses 2^126): a value type without an engine name, so the scheme's host methods take the generic per-thread kernels.  The
// correction words of levels 0 .. n - 1 do not depend on the group: they must equal those of the 2^127 group, which the engine computes.
static void HostSignatureOddModulus() {
  int before = g_fail;
  using G126 = fss::group::Uint<__uint128_t, (static_cast<__uint128_t>(1) << 126)>;
  using G127 = fss::group::Uint<__uint128_t, (static_cast<__uint128_t>(1) << 127)>;
  using Prg = fss::prg::ChaCha<2>;
  using D126 = fss::Dpf<16, G126, Prg, uint16_t>;
  using D127 = fss::Dpf<16, G127, Prg, uint16_t>;
  static_assert(!D126::kEngine && D127::kEngine);
  static int nonce[2] = {0x12345678, static_cast<int>(0x9abcdef0u)};
  Prg prg(nonce);
  D126 d126{prg};
  D127 d127{prg};
  int4 seeds[2] = {kSeeds[0], kSeeds[1]};
  const int4 beta = I4(0xdeadbeef, 0x01234567, 0x89abcdef, 0xfffffffe);  // bit 126 set: reduced modulo 2^126 by From
  D126::Cw c126[17];
  D127::Cw c127[17];
  d126.Gen(c126, seeds, (uint16_t)4242, beta);
  d127.Gen(c127, seeds, (uint16_t)4242, beta);
  int same = 0;
  for (int i = 0; i < 16; ++i) same += Same(c126[i].s, c127[i].s) && c126[i].tr == c127[i].tr;
  EXPECT(same == 16);
  const int4 want = G126::From(beta).Into();
  EXPECT(!Same(want, G127::From(beta).Into()));
  for (uint16_t xx : {(uint16_t)0, (uint16_t)4241, (uint16_t)4242, (uint16_t)4243, (uint16_t)65535})
    EXPECT(Same(Recon<G126>(d126.Eval(false, seeds[0], c126, xx), d126.Eval(true, seeds[1], c126, xx)), xx == 4242 ? want : kZero));
  std::vector<int4> ys0(1 << 16), ys1(1 << 16);
  d126.EvalAll(false, seeds[0], c126, ys0.data());
  d126.EvalAll(true, seeds[1], c126, ys1.data());
  int bad = 0;
  for (int i = 0; i < (1 << 16); ++i) bad += !Same(Recon<G126>(ys0[i], ys1[i]), i == 4242 ? want : kZero);
  EXPECT(bad == 0);
  std::printf("DPF  n=16 Uint<u128, 2^126> (no engine name: generic kernels)   host-signature: %s\n", g_fail == before ? "ok" : "FAILED");
}

static void HostSignatureWide() {
  int before = g_fail;
  {  // DPF n = 32 over Uint<uint64_t> (the bench group, src/bench_gpu.cu:23)
    using Group = fss::group::Uint<uint64_t>;
    using DpfPrg = fss::prg::Aes128Mmo<2>;
    using Dpf = fss::Dpf<32, Group, DpfPrg, uint32_t>;
    const unsigned char* keys[2] = {k0, k1};
    auto ctxs = DpfPrg::CreateCtxs(keys);
    Dpf dpf{DpfPrg(ctxs)};
    int4 seeds[2] = {kSeeds[0], kSeeds[1]};
    Dpf::Cw cws[33];
    dpf.Gen(cws, seeds, 0x12345678u, int4{7, 0, 0, 0});
    EXPECT(Same(cws[32].s, I4(0xd316e94a, 0x28e6b72d, 0, 0)));
    EXPECT(Same(dpf.Eval(false, seeds[0], cws, 0x12345678u), I4(0x3ffa817c, 0xf3ec1fb3, 0, 0)));
    EXPECT(Same(dpf.Eval(true, seeds[1], cws, 0x12345678u), I4(0xc0057e8b, 0x0c13e04c, 0, 0)));
    EXPECT(Same(dpf.Eval(false, seeds[0], cws, 0xdeadbeefu), I4(0xa840848f, 0x144d9ccf, 0, 0)));
    DpfPrg::FreeCtxs(ctxs);
  }
  {  // Half-Tree n = 64 (samples/half_tree_dpf_cpu.cu flow): aggregate {prg, hash_key}, ocw by reference
    using Group = fss::group::Uint<uint64_t>;
    using HtPrg = fss::prg::Aes128Mmo<1>;
    using HtDpf = fss::HalfTreeDpf<64, Group, HtPrg, uint64_t>;
    const unsigned char* keys[1] = {k0};
    auto ctxs = HtPrg::CreateCtxs(keys);
    HtPrg prg(ctxs);
    int4 hash_key = I4(0x12345678, 0x9abcdef0, 0x13572468, 0x2468ace0);
    HtDpf dpf{prg, hash_key};
    int4 seeds[2] = {kSeeds[0], kSeeds[1]};
    auto cws = std::make_unique<HtDpf::Cw[]>(64);
    int4 ocw;
    const uint64_t alpha = 0x0123456789abcdefull;
    dpf.Gen(cws.get(), ocw, seeds, alpha, int4{7, 0, 0, 0});
    EXPECT(Same(cws[0].s, I4(0x02fc4a65, 0xf8fecae6, 0xf249f101, 0x66111f80)) && cws[63].extra == true);
    EXPECT(Same(ocw, I4(0x7c3c2870, 0x598af3a9, 0, 0)));
    EXPECT(Same(dpf.Eval(false, seeds[0], cws.get(), ocw, alpha), I4(0x98317401, 0xbf535b1d, 0, 0)));
    EXPECT(Same(dpf.Eval(true, seeds[1], cws.get(), ocw, alpha), I4(0x67ce8c06, 0x40aca4e2, 0, 0)));
    EXPECT(Same(dpf.Eval(false, seeds[0], cws.get(), ocw, 0xdeadbeefcafef00dull), I4(0xe7bf5193, 0x17c707b4, 0, 0)));
    HtPrg::FreeCtxs(ctxs);

    // a small Half-Tree domain: EvalAll on host arrays, and the 1-level edge case (half_tree_dpf.cuh:259-266)
    using Ht10 = fss::HalfTreeDpf<10, Group, HtPrg, uint16_t>;
    Ht10 h10{HtPrg(keys), hash_key};
    Ht10::Cw c10[10];
    int4 o10;
    h10.Gen(c10, o10, seeds, (uint16_t)777, int4{99, 5, 0, 0});
    std::vector<int4> a0(1024), a1(1024);
    h10.EvalAll(false, seeds[0], c10, o10, a0.data());
    h10.EvalAll(true, seeds[1], c10, o10, a1.data());
    int bad = 0;
    for (int i = 0; i < 1024; ++i)
      if (!Same(Recon<Group>(a0[i], a1[i]), i == 777 ? int4{99, 5, 0, 0} : kZero)) ++bad;
    EXPECT(bad == 0);
    using Ht1 = fss::HalfTreeDpf<1, Group, HtPrg, uint8_t>;
    Ht1 h1{HtPrg(keys), hash_key};
    Ht1::Cw c1[1];
    int4 o1;
    h1.Gen(c1, o1, seeds, (uint8_t)1, int4{3, 0, 0, 0});
    EXPECT(Same(Recon<Group>(h1.Eval(false, seeds[0], c1, o1, 1), h1.Eval(true, seeds[1], c1, o1, 1)), int4{3, 0, 0, 0}));
    EXPECT(Same(Recon<Group>(h1.Eval(false, seeds[0], c1, o1, 0), h1.Eval(true, seeds[1], c1, o1, 0)), kZero));
  }
  {  // Grotto (samples/grotto_dcf_cpu.cu:29-82): ParityTree on the host, static Eval, EvalAll into bools
    constexpr int kInBits = 8;
    using In = uint8_t;
    using GdcfPrg = fss::prg::Aes128Mmo<2>;
    using Gdcf = fss::GrottoDcf<kInBits, GdcfPrg, In>;
    const unsigned char* keys[2] = {k0, k1};
    auto ctxs = GdcfPrg::CreateCtxs(keys);
    GdcfPrg prg(ctxs);
    Gdcf dcf{prg};
    In alpha = 42;
    int4 seeds[2] = {kSeeds[0], kSeeds[1]};
    Gdcf::Cw cws[kInBits + 1];
    dcf.Gen(cws, seeds, alpha);
    constexpr size_t N = size_t{1} << kInBits;
    auto p0 = std::make_unique<bool[]>(2 * N - 1);
    auto p1 = std::make_unique<bool[]>(2 * N - 1);
    Gdcf::ParityTree pt0{p0.get(), false};
    Gdcf::ParityTree pt1{p1.get(), true};
    dcf.Preprocess(pt0, seeds[0], cws);
    dcf.Preprocess(pt1, seeds[1], cws);
    auto ys0 = std::make_unique<bool[]>(N);
    auto ys1 = std::make_unique<bool[]>(N);
    dcf.EvalAll(false, seeds[0], cws, ys0.get());
    dcf.EvalAll(true, seeds[1], cws, ys1.get());
    int bad = 0;
    for (size_t i = 0; i < N; ++i) {
      bool e0 = Gdcf::Eval(pt0, (In)i), e1 = Gdcf::Eval(pt1, (In)i);
      if ((e0 ^ e1) != (alpha <= i)) ++bad;
      if (e0 != ys0[i] || e1 != ys1[i]) ++bad;  // Preprocess + Eval == EvalAll, per party (src/grotto_dcf_test.cu:79-96)
    }
    EXPECT(bad == 0);
    GdcfPrg::FreeCtxs(ctxs);
  }
  {  // Uint<__uint128_t, 2^127> and a modulus, through the reference's group arithmetic for the reconstruction
    using G128 = fss::group::Uint<unsigned __int128, static_cast<unsigned __int128>(1) << 127>;
    using GMod = fss::group::Uint<uint32_t, 1000003u>;
    using DpfPrg = fss::prg::Aes128Mmo<2>;
    const unsigned char* keys[2] = {k0, k1};
    int4 seeds[2] = {kSeeds[0], kSeeds[1]};
    fss::Dpf<20, G128, DpfPrg, uint32_t> d128{DpfPrg(keys)};
    fss::Dpf<20, G128, DpfPrg, uint32_t>::Cw c128[21];
    int4 b128 = I4(0xdeadbeef, 0x01234567, 0x89abcdef, 0x7ffffffe);
    d128.Gen(c128, seeds, 0xabcdeu, b128);
    EXPECT(Same(Recon<G128>(d128.Eval(false, seeds[0], c128, 0xabcdeu), d128.Eval(true, seeds[1], c128, 0xabcdeu)),
                G128::From(b128).Into()));
    EXPECT(Same(Recon<G128>(d128.Eval(false, seeds[0], c128, 0xabcdfu), d128.Eval(true, seeds[1], c128, 0xabcdfu)), kZero));
    fss::Dpf<12, GMod, DpfPrg, uint16_t> dm{DpfPrg(keys)};
    fss::Dpf<12, GMod, DpfPrg, uint16_t>::Cw cm[13];
    int4 bm = I4(0xfffffff0, 0, 0, 0);
    dm.Gen(cm, seeds, (uint16_t)3000, bm);
    EXPECT(Same(Recon<GMod>(dm.Eval(false, seeds[0], cm, 3000), dm.Eval(true, seeds[1], cm, 3000)), GMod::From(bm).Into()));
    EXPECT(Same(Recon<GMod>(dm.Eval(false, seeds[0], cm, 3001), dm.Eval(true, seeds[1], cm, 3001)), kZero));
  }
  std::printf("DPF n=32 Uint<u64>, Half-Tree n=64/10/1, Grotto n=8, Uint<u128>/mod  host-signature: %s\n",
              g_fail == before ? "ok" : "FAILED");
}

// ---------------------------------------------------------------------------------------------- parts 2 and 3
template <typename T>
static T* ToDevice(const T* host, size_t n) {
  T* dev;
  HIP_OK(hipMalloc(&dev, sizeof(T) * n));
  HIP_OK(hipMemcpy(dev, host, sizeof(T) * n, hipMemcpyHostToDevice));
  return dev;
}
template <typename T>
static void ToHost(T* host, const T* dev, size_t n) {
  HIP_OK(hipMemcpy(host, dev, sizeof(T) * n, hipMemcpyDeviceToHost));
}

constexpr int kN = 1 << 12;
using uint = unsigned int;
using UintGroup = fss::group::Uint<uint64_t>;
static const int kNonce[2] = {0x12345678, static_cast<int>(0x9abcdef0u)};

struct GpuData {  // the reference benches' synthetic inputs (src/bench_gpu.cu:239-290), srand(42)
  std::vector<int4> h_seeds, h_seeds0, h_seeds1, h_betas;
  std::vector<uint> h_alphas, h_xs;
  int4 *d_seeds, *d_seeds0, *d_seeds1, *d_betas, *d_ys;
  uint *d_alphas, *d_xs;
  explicit GpuData(int in_bits) : h_seeds(2 * kN), h_seeds0(kN), h_seeds1(kN), h_betas(kN), h_alphas(kN), h_xs(kN) {
    const uint mask = in_bits >= 32 ? 0xffffffffu : ((1u << in_bits) - 1);
    srand(42);
    for (int i = 0; i < kN; i++) {
      h_seeds[i * 2] = {rand(), rand(), rand(), rand() & ~1};
      h_seeds[i * 2 + 1] = {rand(), rand(), rand(), rand() & ~1};
      h_seeds0[i] = h_seeds[i * 2];
      h_seeds1[i] = h_seeds[i * 2 + 1];
      h_alphas[i] = (uint)rand() & mask;
      h_betas[i] = {rand(), rand(), rand(), rand() & ~1};
      h_xs[i] = (i % 2 == 0) ? h_alphas[i] : ((uint)rand() & mask);
    }
    d_seeds = ToDevice(h_seeds.data(), 2 * kN);
    d_seeds0 = ToDevice(h_seeds0.data(), kN);
    d_seeds1 = ToDevice(h_seeds1.data(), kN);
    d_alphas = ToDevice(h_alphas.data(), kN);
    d_betas = ToDevice(h_betas.data(), kN);
    d_xs = ToDevice(h_xs.data(), kN);
    HIP_OK(hipMalloc(&d_ys, sizeof(int4) * kN));
  }
  ~GpuData() {
    (void)hipFree(d_seeds); (void)hipFree(d_seeds0); (void)hipFree(d_seeds1); (void)hipFree(d_alphas);
    (void)hipFree(d_betas); (void)hipFree(d_xs); (void)hipFree(d_ys);
  }
};

template <int in_bits, typename Group, typename Prg>
static void DpfLaunchers(const Prg& prg, const char* tag) {
  using DpfType = fss::Dpf<in_bits, Group, Prg, uint>;
  int before = g_fail;
  GpuData data(in_bits);
  DpfType dpf{prg};

  typename DpfType::Cw* d_cws;
  int4 *d_cw_s, *d_out_cw;
  uint32_t* d_extra;
  HIP_OK(hipMalloc(&d_cws, sizeof(typename DpfType::Cw) * (in_bits + 1) * kN));
  HIP_OK(hipMalloc(&d_cw_s, sizeof(int4) * in_bits * kN));
  HIP_OK(hipMalloc(&d_extra, sizeof(uint32_t) * kN));
  HIP_OK(hipMalloc(&d_out_cw, sizeof(int4) * kN));
  dpf.GenBatch(d_cws, data.d_seeds, data.d_alphas, data.d_betas, kN);  // the reference runs a Gen kernel here
  HIP_OK(hipDeviceSynchronize());

  // part 2: per-key Eval with the reference's call expression, keys resident on the device
  std::vector<int4> ref0(kN), ref1(kN);
  for (int tid = 0; tid < 64; ++tid) {
    ref0[tid] = dpf.Eval(false, data.h_seeds0[tid], d_cws + tid * (in_bits + 1), data.h_xs[tid]);
    ref1[tid] = dpf.Eval(true, data.h_seeds1[tid], d_cws + tid * (in_bits + 1), data.h_xs[tid]);
  }
  // ... and a key generated by the host-signature Gen equals the batched Gen's
  {
    std::vector<typename DpfType::Cw> one(in_bits + 1), dev(in_bits + 1);
    int4 s[2] = {data.h_seeds[10], data.h_seeds[11]};
    dpf.Gen(one.data(), s, data.h_alphas[5], data.h_betas[5]);
    ToHost(dev.data(), d_cws + 5 * (in_bits + 1), in_bits + 1);
    bool eq = true;
    for (int i = 0; i <= in_bits; ++i) eq = eq && Same(one[i].s, dev[i].s) && (i == in_bits || one[i].tr == dev[i].tr);
    EXPECT(eq);
  }

  // part 3: src/bench_gpu.cu:618 and :625
  fss::gpu::DpfRelayoutGpu<in_bits, Group, Prg, uint>(d_cws, kN, d_cw_s, d_extra, d_out_cw);
  HIP_OK(hipDeviceSynchronize());
  std::vector<int4> y0(kN), y1(kN);
  fss::gpu::DpfEvalPointGpu<1, in_bits, Group, Prg, uint>(
      false, data.d_seeds0, d_cw_s, d_extra, d_out_cw, data.d_xs, data.d_ys, kN, dpf);
  HIP_OK(hipPeekAtLastError());
  ToHost(y0.data(), data.d_ys, kN);
  fss::gpu::DpfEvalPointGpu(true, data.d_seeds1, d_cw_s, d_extra, d_out_cw, data.d_xs, data.d_ys, kN, dpf);
  ToHost(y1.data(), data.d_ys, kN);
  int bad = 0;
  for (int i = 0; i < kN; ++i) {
    int4 want = data.h_xs[i] == data.h_alphas[i] ? Group::From(fss::util::SetLsb(data.h_betas[i], false)).Into() : kZero;
    if (!Same(Recon<Group>(y0[i], y1[i]), want)) ++bad;
    if (i < 64 && (!Same(y0[i], ref0[i]) || !Same(y1[i], ref1[i]))) ++bad;
  }
  EXPECT(bad == 0);

  // src/bench_gpu.cu:578 shape on a small domain: every key's full domain in one launch, seeds party-major
  if constexpr (in_bits <= 12) {
    constexpr int kBatch = 16;
    int4* d_ys;
    HIP_OK(hipMalloc(&d_ys, sizeof(int4) * (1 << in_bits) * kBatch));
    for (int party = 0; party < 2; ++party) {
      const int4* d_s0s = party ? data.d_seeds1 : data.d_seeds0;
      int off = 32;
      fss::gpu::DpfEvalAllGpuBatch<17, 9, 256>(party, d_s0s + off, d_cws + off * (in_bits + 1), kBatch, d_ys, dpf);
      std::vector<int4> all((size_t)kBatch << in_bits);
      ToHost(all.data(), d_ys, all.size());
      // key 32 + 3 of the batch against the host-signature EvalAll on a host copy of that key
      std::vector<typename DpfType::Cw> key(in_bits + 1);
      ToHost(key.data(), d_cws + (off + 3) * (in_bits + 1), in_bits + 1);
      std::vector<int4> one(size_t(1) << in_bits);
      dpf.EvalAll(party, (party ? data.h_seeds1 : data.h_seeds0)[off + 3], key.data(), one.data());
      EXPECT(std::memcmp(one.data(), all.data() + (size_t(3) << in_bits), one.size() * sizeof(int4)) == 0);
      // check_evalall_gpu.cu:64-91: DpfEvalAllGpu of one key (seed by value, no workspace) == the same
      fss::gpu::DpfEvalAllGpu<(in_bits < 16 ? in_bits : 16), (in_bits < 16 ? in_bits - 8 : 8)>(
          party, (party ? data.h_seeds1 : data.h_seeds0)[off + 3], d_cws + (off + 3) * (in_bits + 1), d_ys, dpf);
      std::vector<int4> single(size_t(1) << in_bits);
      ToHost(single.data(), d_ys, single.size());
      EXPECT(std::memcmp(one.data(), single.data(), one.size() * sizeof(int4)) == 0);
    }
    HIP_OK(hipFree(d_ys));
  }
  HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_cw_s)); HIP_OK(hipFree(d_extra)); HIP_OK(hipFree(d_out_cw));
  std::printf("DPF  n=%-2d %s launchers (Relayout uint32 extra / EvalPoint<1,...> / EvalAllGpu[Batch]): %s\n", in_bits, tag,
              g_fail == before ? "ok" : "FAILED");
}

template <int in_bits, typename Group, typename Prg>
static void DcfLaunchers(const Prg& prg, const char* tag) {
  using DcfType = fss::Dcf<in_bits, Group, Prg, uint>;
  int before = g_fail;
  GpuData data(in_bits);
  typename DcfType::Cw* d_cws;
  int4 *d_cw_s, *d_cw_v, *d_out_cw;
  HIP_OK(hipMalloc(&d_cws, sizeof(typename DcfType::Cw) * (in_bits + 1) * kN));
  HIP_OK(hipMalloc(&d_cw_s, sizeof(int4) * in_bits * kN));
  HIP_OK(hipMalloc(&d_cw_v, sizeof(int4) * in_bits * kN));
  HIP_OK(hipMalloc(&d_out_cw, sizeof(int4) * kN));
  DcfType dcf{prg};
  dcf.GenBatch(d_cws, data.d_seeds, data.d_alphas, data.d_betas, kN);
  fss::gpu::DcfRelayoutGpu<in_bits, Group, Prg, uint>(d_cws, kN, d_cw_s, d_cw_v, d_out_cw);
  HIP_OK(hipDeviceSynchronize());
  std::vector<int4> y0(kN), y1(kN);
  fss::gpu::DcfEvalPointGpu<4, in_bits, Group, Prg, uint>(
      false, data.d_seeds0, d_cw_s, d_cw_v, d_out_cw, data.d_xs, data.d_ys, kN, dcf);
  ToHost(y0.data(), data.d_ys, kN);
  fss::gpu::DcfEvalPointGpu<4, in_bits, Group, Prg, uint>(
      true, data.d_seeds1, d_cw_s, d_cw_v, d_out_cw, data.d_xs, data.d_ys, kN, dcf);
  ToHost(y1.data(), data.d_ys, kN);
  int bad = 0;
  for (int i = 0; i < kN; ++i) {
    int4 want = data.h_xs[i] < data.h_alphas[i] ? Group::From(fss::util::SetLsb(data.h_betas[i], false)).Into() : kZero;
    if (!Same(Recon<Group>(y0[i], y1[i]), want)) ++bad;
    if (i < 32 && !Same(y1[i], dcf.Eval(true, data.h_seeds1[i], d_cws + i * (in_bits + 1), data.h_xs[i]))) ++bad;
  }
  EXPECT(bad == 0);
  HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_cw_s)); HIP_OK(hipFree(d_cw_v)); HIP_OK(hipFree(d_out_cw));
  std::printf("DCF  n=%-2d %s launchers: %s\n", in_bits, tag, g_fail == before ? "ok" : "FAILED");
}

template <int in_bits, typename Group, typename Prg>
static void HalfTreeLaunchers(const Prg& prg, const char* tag) {
  using HtDpf = fss::HalfTreeDpf<in_bits, Group, Prg, uint>;
  const int4 kHalfTreeHashKeyHost = {0x12345678, static_cast<int>(0x9abcdef0u), 0x0fedcba9, static_cast<int>(0x87654321u)};
  int before = g_fail;
  GpuData data(in_bits);
  typename HtDpf::Cw* d_cws;
  int4 *d_cw_s, *d_ocws;
  uint32_t* d_extra;
  HIP_OK(hipMalloc(&d_cws, sizeof(typename HtDpf::Cw) * in_bits * kN));
  HIP_OK(hipMalloc(&d_cw_s, sizeof(int4) * in_bits * kN));
  HIP_OK(hipMalloc(&d_extra, sizeof(uint32_t) * kN));
  HIP_OK(hipMalloc(&d_ocws, sizeof(int4) * kN));
  HtDpf dpf{prg, kHalfTreeHashKeyHost};
  dpf.GenBatch(d_cws, d_ocws, data.d_seeds, data.d_alphas, data.d_betas, kN);
  fss::gpu::HalfTreeDpfRelayoutGpu<in_bits, Group, Prg, uint>(d_cws, kN, d_cw_s, d_extra);
  HIP_OK(hipDeviceSynchronize());
  std::vector<int4> y0(kN), y1(kN), ocws(kN);
  ToHost(ocws.data(), d_ocws, kN);
  fss::gpu::HalfTreeDpfEvalPointGpu<4, in_bits, Group, Prg, uint>(
      false, data.d_seeds0, d_cw_s, d_extra, d_ocws, data.d_xs, data.d_ys, kN, dpf);
  ToHost(y0.data(), data.d_ys, kN);
  fss::gpu::HalfTreeDpfEvalPointGpu<4, in_bits, Group, Prg, uint>(
      true, data.d_seeds1, d_cw_s, d_extra, d_ocws, data.d_xs, data.d_ys, kN, dpf);
  ToHost(y1.data(), data.d_ys, kN);
  int bad = 0;
  for (int i = 0; i < kN; ++i) {
    int4 want = data.h_xs[i] == data.h_alphas[i] ? Group::From(fss::util::SetLsb(data.h_betas[i], false)).Into() : kZero;
    if (!Same(Recon<Group>(y0[i], y1[i]), want)) ++bad;
    if (i < 32 && !Same(y0[i], dpf.Eval(false, data.h_seeds0[i], d_cws + i * in_bits, ocws[i], data.h_xs[i]))) ++bad;
  }
  EXPECT(bad == 0);
  if constexpr (in_bits <= 12) {
    constexpr int kBatch = 8;
    int4* d_ys;
    HIP_OK(hipMalloc(&d_ys, sizeof(int4) * (1 << in_bits) * kBatch));
    int off = 64;
    fss::gpu::HalfTreeDpfEvalAllGpuBatch<17, 9, 256>(
        false, data.d_seeds0 + off, d_cws + off * in_bits, d_ocws + off, kBatch, d_ys, dpf);
    std::vector<int4> all((size_t)kBatch << in_bits), single(size_t(1) << in_bits);
    ToHost(all.data(), d_ys, all.size());
    fss::gpu::HalfTreeDpfEvalAllGpu(false, data.h_seeds0[off + 1], d_cws + (off + 1) * in_bits, ocws[off + 1], d_ys, dpf);
    ToHost(single.data(), d_ys, single.size());
    EXPECT(std::memcmp(single.data(), all.data() + (size_t(1) << in_bits), single.size() * sizeof(int4)) == 0);
    EXPECT(Same(single[data.h_xs[off + 1]], y0[off + 1]));
    HIP_OK(hipFree(d_ys));
  }
  HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_cw_s)); HIP_OK(hipFree(d_extra)); HIP_OK(hipFree(d_ocws));
  std::printf("HT   n=%-2d %s launchers: %s\n", in_bits, tag, g_fail == before ? "ok" : "FAILED");
}

// a stream other than the null stream: the launchers' pool is per stream, the seed upload is stream-ordered
static void OnAStream() {
  int before = g_fail;
  using DpfPrg = fss::prg::Aes128Mmo<2>;
  using DpfType = fss::Dpf<12, UintGroup, DpfPrg, uint>;
  const unsigned char* keys[2] = {k0, k1};
  DpfType dpf{DpfPrg(keys)};
  int4 seeds[2] = {kSeeds[0], kSeeds[1]};
  DpfType::Cw cws[13];
  dpf.Gen(cws, seeds, 1000u, int4{5, 6, 0, 0});
  DpfType::Cw* d_cws = ToDevice(cws, 13);
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  int4 *d_a, *d_b;
  HIP_OK(hipMalloc(&d_a, sizeof(int4) << 12));
  HIP_OK(hipMalloc(&d_b, sizeof(int4) << 12));
  fss::gpu::DpfEvalAllGpu(false, seeds[0], d_cws, d_a, dpf, st);
  fss::gpu::DpfEvalAllGpu(true, seeds[1], d_cws, d_b, dpf, st);  // reuses the pool behind the first launch
  HIP_OK(hipStreamSynchronize(st));
  std::vector<int4> a(4096), b(4096);
  ToHost(a.data(), d_a, 4096);
  ToHost(b.data(), d_b, 4096);
  int bad = 0;
  for (int i = 0; i < 4096; ++i)
    if (!Same(Recon<UintGroup>(a[i], b[i]), i == 1000 ? int4{5, 6, 0, 0} : kZero)) ++bad;
  EXPECT(bad == 0);
  HIP_OK(hipStreamDestroy(st));
  HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_a)); HIP_OK(hipFree(d_b));
  std::printf("DpfEvalAllGpu on a side stream: %s\n", g_fail == before ? "ok" : "FAILED");
}

// buffers in PINNED host memory: EvalAll writes its leaves into them from the kernels (no staging buffer behind the
// host-signature call), and EvalPointsHost walks a batch whose x and y both live on the host -- the same outputs as
// through pageable memory (staged) and through device memory
static void PinnedHostBuffers() {
  int before = g_fail;
  using DpfPrg = fss::prg::Aes128Mmo<2>;
  using DpfType = fss::Dpf<16, UintGroup, DpfPrg, uint>;
  const unsigned char* keys[2] = {k0, k1};
  DpfType dpf{DpfPrg(keys)};
  int4 seeds[2] = {kSeeds[0], kSeeds[1]};
  DpfType::Cw cws[17];
  dpf.Gen(cws, seeds, 4242u, int4{9, 1, 0, 0});
  const size_t n = size_t(1) << 16;
  std::vector<int4> pageable(n);
  dpf.EvalAll(false, seeds[0], cws, pageable.data());
  int4* pinned = nullptr;
  HIP_OK(hipHostMalloc(&pinned, n * sizeof(int4)));
  std::memset(pinned, 0xff, n * sizeof(int4));
  dpf.EvalAll(false, seeds[0], cws, pinned);
  EXPECT(std::memcmp(pinned, pageable.data(), n * sizeof(int4)) == 0);
  // a registered (hipHostRegister) buffer is pinned memory too
  std::vector<int4> reg(n + 1);
  int4* reg_p = reinterpret_cast<int4*>((reinterpret_cast<uintptr_t>(reg.data()) + 15) & ~uintptr_t(15));
  HIP_OK(hipHostRegister(reg_p, n * sizeof(int4), hipHostRegisterDefault));
  dpf.EvalAll(false, seeds[0], cws, reg_p);
  EXPECT(std::memcmp(reg_p, pageable.data(), n * sizeof(int4)) == 0);
  HIP_OK(hipHostUnregister(reg_p));
  // a batch of points on the host: every x of the domain in a scrambled order, ragged length
  const int64_t m = (int64_t)n - 37;
  uint* xs = nullptr;
  HIP_OK(hipHostMalloc(&xs, m * sizeof(uint)));
  for (int64_t i = 0; i < m; ++i) xs[i] = (uint)((i * 40503u + 7u) & 0xffffu);
  std::memset(pinned, 0xff, n * sizeof(int4));
  DpfType::Cw* d_cws = ToDevice(cws, 17);
  int4* d_s0 = ToDevice(&seeds[0], 1);
  dpf.EvalPointsHost(false, d_s0, d_cws, xs, pinned, m);
  HIP_OK(hipDeviceSynchronize());
  int bad = 0;
  for (int64_t i = 0; i < m; ++i)
    if (!Same(pinned[i], pageable[xs[i]])) ++bad;
  EXPECT(bad == 0);
  EXPECT(pinned[m].x == -1);  // nothing written past the batch
  // pageable memory cannot be walked in place: refused, not read
  std::vector<uint> xs_pg(64, 1u);
  EXPECT(fss_amd_eval_points_host(dpf.handle(), FSS_AMD_DPF, 16, UintGroup::kId, 0, d_s0, d_cws, nullptr, xs_pg.data(), 4, 64,
                                  pinned, nullptr) == FSS_AMD_EINVAL);
  HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_s0));
  HIP_OK(hipHostFree(xs)); HIP_OK(hipHostFree(pinned));
  {  // Grotto: EvalAll's bools and Preprocess's table are updated in place by later kernels of the call, so pinned
     // buffers are staged like pageable ones -- same contents either way
    using Gdcf = fss::GrottoDcf<12, DpfPrg, uint16_t>;
    Gdcf gd{DpfPrg(keys)};
    Gdcf::Cw gcws[13];
    gd.Gen(gcws, seeds, (uint16_t)1234);
    constexpr size_t N = size_t(1) << 12;
    std::vector<uint8_t> ya(N), ta(2 * N - 1);
    bool *yp = nullptr, *tp = nullptr;
    HIP_OK(hipHostMalloc(&yp, N));
    HIP_OK(hipHostMalloc(&tp, 2 * N - 1));
    gd.EvalAll(true, seeds[1], gcws, reinterpret_cast<bool*>(ya.data()));
    gd.EvalAll(true, seeds[1], gcws, yp);
    EXPECT(std::memcmp(yp, ya.data(), N) == 0);
    Gdcf::ParityTree pa{reinterpret_cast<bool*>(ta.data()), true}, pp{tp, true};
    gd.Preprocess(pa, seeds[1], gcws);
    gd.Preprocess(pp, seeds[1], gcws);
    EXPECT(std::memcmp(tp, ta.data(), 2 * N - 1) == 0);
    HIP_OK(hipHostFree(yp)); HIP_OK(hipHostFree(tp));
  }
  std::printf("pinned host buffers (EvalAll in place, EvalPointsHost, Grotto staged): %s\n", g_fail == before ? "ok" : "FAILED");
}

// One thread launches on a stream while another releases that stream's scratch (round-4 advisor finding: the release
// held the registry lock while taking the launch lock, the launchers the other way round -- a deadlock).  The run must
// end, and a launch after the last release must still be right.
static void LaunchVersusRelease() {
  int before = g_fail;
  using DpfPrg = fss::prg::Aes128Mmo<2>;
  using DpfType = fss::Dpf<10, UintGroup, DpfPrg, uint>;
  const unsigned char* keys[2] = {k0, k1};
  DpfType dpf{DpfPrg(keys)};
  int4 seeds[2] = {kSeeds[0], kSeeds[1]};
  DpfType::Cw cws[11];
  dpf.Gen(cws, seeds, 77u, int4{3, 0, 0, 0});
  DpfType::Cw* d_cws = ToDevice(cws, 11);
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  int4 *d_a, *d_b;
  HIP_OK(hipMalloc(&d_a, sizeof(int4) << 10));
  HIP_OK(hipMalloc(&d_b, sizeof(int4) << 10));
  std::atomic<int> done{0};
  std::thread launcher([&] {
    for (int i = 0; i < 300; ++i) fss::gpu::DpfEvalAllGpu(false, seeds[0], d_cws, d_a, dpf, st);
    ++done;
  });
  std::thread releaser([&] {
    for (int i = 0; i < 60; ++i) fss::ReleaseStreamScratch(st);
    ++done;
  });
  for (int waited = 0; done.load() < 2; ++waited) {
    if (waited > 1200) {  // two minutes: the two threads are stuck on each other's locks
      std::fprintf(stderr, "FAILED: launch loop and ReleaseStreamScratch deadlocked\n");
      std::_Exit(4);
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
  }
  launcher.join();
  releaser.join();
  fss::gpu::DpfEvalAllGpu(false, seeds[0], d_cws, d_a, dpf, st);
  fss::gpu::DpfEvalAllGpu(true, seeds[1], d_cws, d_b, dpf, st);
  HIP_OK(hipStreamSynchronize(st));
  std::vector<int4> a(1024), b(1024);
  ToHost(a.data(), d_a, 1024);
  ToHost(b.data(), d_b, 1024);
  int bad = 0;
  for (int i = 0; i < 1024; ++i)
    if (!Same(Recon<UintGroup>(a[i], b[i]), i == 77 ? int4{3, 0, 0, 0} : kZero)) ++bad;
  EXPECT(bad == 0);
  fss::ReleaseStreamScratch(st);
  HIP_OK(hipStreamDestroy(st));
  HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_a)); HIP_OK(hipFree(d_b));
  std::printf("launch loop against ReleaseStreamScratch on one stream: %s\n", g_fail == before ? "ok" : "FAILED");
}


// ---------------------------------------------------------------------------------------------- part 4: the verifiable DPF
// Caller code in the reference's own syntax: the class with its seven template parameters and three public members,
// Gen / Eval / Prove / Verify / EvalAll as src/vdpf_test.cu:28-110 and samples/vdpf_cpu.cu:22-80 call them, the per-thread
// device forms as src/bench_gpu.cu:170-205 (VdpfGenKernel / VdpfEvalKernel), the launchers with their six explicit template
// arguments, `uint32_t* extra`, `const uint32_t* xs` and `const array<int4, 4>* cs` as src/bench_gpu.cu:704-735.
template <int in_bits, typename Group>
using VdpfType = fss::Vdpf<in_bits, Group, fss::prg::ChaCha<2>, fss::hash::Blake3, fss::hash::Blake3, uint>;
static const int4 kBlake3Iv[2] = {{0x11111111, 0x22222222, 0x33333333, 0x44444444},
                                  {0x55555555, 0x66666666, 0x77777777, static_cast<int>(0x88888888u)}};
__constant__ int kNonceDev[2] = {0x12345678, static_cast<int>(0x9abcdef0u)};
__constant__ int4 kBlake3IvDev[2] = {{0x11111111, 0x22222222, 0x33333333, 0x44444444},
                                     {0x55555555, 0x66666666, 0x77777777, static_cast<int>(0x88888888u)}};

template <int in_bits, typename Group>
__global__ void VdpfGenKernel(typename VdpfType<in_bits, Group>::Cw* cws, std::array<int4, 4>* cs, int4* ocws, const int4* seeds,
                              const uint* alphas, const int4* betas, int* rets) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= kN) return;
  fss::prg::ChaCha<2> prg(kNonceDev);
  fss::hash::Blake3 xor_hash{std::span<const int4, 2>(kBlake3IvDev)};
  fss::hash::Blake3 hash{std::span<const int4, 2>(kBlake3IvDev)};
  VdpfType<in_bits, Group> vdpf{prg, xor_hash, hash};
  int4 s0s_arr[2] = {seeds[tid * 2], seeds[tid * 2 + 1]};
  std::span<const int4, 2> s0s(s0s_arr);
  rets[tid] = vdpf.Gen(cws + tid * in_bits, cs[tid], ocws[tid], s0s, alphas[tid], betas[tid]);
}
template <int in_bits, typename Group>
__global__ void VdpfEvalKernel(int4* ys, std::array<int4, 4>* pis, bool party, const int4* seeds, const typename VdpfType<in_bits, Group>::Cw* cws,
                               const std::array<int4, 4>* cs, const int4* ocws, const uint* xs) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= kN) return;
  fss::prg::ChaCha<2> prg(kNonceDev);
  fss::hash::Blake3 xor_hash{std::span<const int4, 2>(kBlake3IvDev)};
  fss::hash::Blake3 hash{std::span<const int4, 2>(kBlake3IvDev)};
  VdpfType<in_bits, Group> vdpf{prg, xor_hash, hash};
  int4 y;
  std::span<const typename VdpfType<in_bits, Group>::Cw> cws_span(cws + tid * in_bits, in_bits);
  std::span<const int4, 4> cs_span(cs[tid].data(), 4);
  pis[tid] = vdpf.Eval(party, seeds[tid], cws_span, cs_span, ocws[tid], xs[tid], y);
  ys[tid] = y;
}

template <typename V, typename In>
static void VdpfHostFlow(V vdpf, In alpha, const char* tag) {  // src/vdpf_test.cu:59-110, samples/vdpf_cpu.cu:55-80
  const int before = g_fail;
  using Group = typename V::GroupType;
  constexpr int n = V::kSlots;
  int4 s0s[2];
  typename V::Cw cws[n];
  std::array<int4, 4> cs;
  int4 ocw = kZero;
  const int4 b_buf = I4(604, 0, 0, 0);
  int ret, tries = 0;
  srand(7);
  do {
    s0s[0] = {rand(), rand(), rand(), rand() & ~1};
    s0s[1] = {rand(), rand(), rand(), rand() & ~1};
    ret = vdpf.Gen(cws, cs, ocw, std::span<const int4, 2>(s0s, 2), alpha, b_buf);
  } while (ret != 0 && ++tries < 64);
  EXPECT(ret == 0);
  std::vector<std::array<int4, 4>> pt0, pt1;
  const In mask = static_cast<In>(n >= (int)sizeof(In) * 8 ? ~In(0) : (In(1) << n) - 1);
  for (int i = 0; i < 24; ++i) {
    const In x = i == 0 ? alpha : static_cast<In>((alpha + In(i) * 2654435761u) & mask);
    int4 y0, y1;
    pt0.push_back(vdpf.Eval(false, s0s[0], std::span<const typename V::Cw>(cws, n), std::span<const int4, 4>(cs), ocw, x, y0));
    pt1.push_back(vdpf.Eval(true, s0s[1], std::span<const typename V::Cw>(cws, n), std::span<const int4, 4>(cs), ocw, x, y1));
    EXPECT(Same(Recon<Group>(y0, y1), x == alpha ? Group::From(b_buf).Into() : kZero));
    EXPECT(std::memcmp(pt0.back().data(), pt1.back().data(), 64) == 0);  // an honest key: the corrected hashes agree point by point
  }
  std::array<int4, 4> pi0, pi1;
  vdpf.Prove(std::span<const std::array<int4, 4>>(pt0), std::span<const int4, 4>(cs), pi0);
  vdpf.Prove(std::span<const std::array<int4, 4>>(pt1), std::span<const int4, 4>(cs), pi1);
  EXPECT(V::Verify(std::span<const int4, 4>(pi0), std::span<const int4, 4>(pi1)));
  if constexpr (V::kEngine) {  // the same fold on the device, the hashes left where a batched evaluation writes them
    std::array<int4, 4>* d_pt = ToDevice(pt0.data(), pt0.size());
    std::array<int4, 4> pid;
    vdpf.Prove(std::span<const std::array<int4, 4>>(d_pt, pt0.size()), std::span<const int4, 4>(cs), pid);
    EXPECT(std::memcmp(pid.data(), pi0.data(), 64) == 0);  // header hash on the host == engine hash on the device
    HIP_OK(hipFree(d_pt));
  }
  {  // a forged key (one bit of a correction word): party 1's proof no longer matches
    typename V::Cw bad[n];
    std::memcpy(bad, cws, sizeof bad);
    bad[n / 2].s.y ^= 0x100;
    std::vector<std::array<int4, 4>> ptb;
    for (int i = 0; i < 24; ++i) {
      const In x = i == 0 ? alpha : static_cast<In>((alpha + In(i) * 2654435761u) & mask);
      int4 y1;
      ptb.push_back(vdpf.Eval(true, s0s[1], std::span<const typename V::Cw>(bad, n), std::span<const int4, 4>(cs), ocw, x, y1));
    }
    std::array<int4, 4> pib;
    vdpf.Prove(std::span<const std::array<int4, 4>>(ptb), std::span<const int4, 4>(cs), pib);
    EXPECT(!V::Verify(std::span<const int4, 4>(pi0), std::span<const int4, 4>(pib)));
  }
  if constexpr (n <= 12) {  // EvalAll (vdpf.cuh:313-361): every x in index order, and the proof over all of them
    std::vector<int4> ys0(size_t(1) << n), ys1(size_t(1) << n);
    std::array<int4, 4> pa0, pa1;
    vdpf.EvalAll(false, s0s[0], std::span<const typename V::Cw>(cws, n), std::span<const int4, 4>(cs), ocw, std::span<int4>(ys0), pa0);
    vdpf.EvalAll(true, s0s[1], std::span<const typename V::Cw>(cws, n), std::span<const int4, 4>(cs), ocw, std::span<int4>(ys1), pa1);
    int bad = 0;
    for (size_t x = 0; x < ys0.size(); ++x) bad += !Same(Recon<Group>(ys0[x], ys1[x]), x == alpha ? Group::From(b_buf).Into() : kZero);
    EXPECT(bad == 0);
    EXPECT(V::Verify(std::span<const int4, 4>(pa0), std::span<const int4, 4>(pa1)));
    std::vector<std::array<int4, 4>> all;  // == Eval at every x, then Prove
    for (size_t x = 0; x < ys0.size(); ++x) {
      int4 y;
      all.push_back(vdpf.Eval(false, s0s[0], std::span<const typename V::Cw>(cws, n), std::span<const int4, 4>(cs), ocw, static_cast<In>(x), y));
      bad += !Same(y, ys0[x]);
    }
    std::array<int4, 4> pe;
    vdpf.Prove(std::span<const std::array<int4, 4>>(all), std::span<const int4, 4>(cs), pe);
    EXPECT(bad == 0 && std::memcmp(pe.data(), pa0.data(), 64) == 0);
  }
  std::printf("VDPF %s  Gen / Eval / Prove / Verify%s, host signatures: %s\n", tag, n <= 12 ? " / EvalAll" : "", g_fail == before ? "ok" : "FAILED");
}

template <int in_bits, typename Group>
static void VdpfLaunchers() {  // src/bench_gpu.cu:704-735
  const int before = g_fail;
  using Vdpf = VdpfType<in_bits, Group>;
  GpuData data(in_bits);
  typename Vdpf::Cw* d_cws;
  int4 *d_cw_s, *d_pi, *d_ys1;
  uint32_t* d_extra;
  std::array<int4, 4>* d_cs;
  std::array<int4, 4>* d_pit;
  int4* d_ocws;
  int* d_rets;
  HIP_OK(hipMalloc(&d_cws, sizeof(typename Vdpf::Cw) * in_bits * kN));
  HIP_OK(hipMalloc(&d_cw_s, sizeof(int4) * in_bits * kN));
  HIP_OK(hipMalloc(&d_extra, sizeof(uint32_t) * kN));
  HIP_OK(hipMalloc(&d_pi, sizeof(int4) * kN * 4));
  HIP_OK(hipMalloc(&d_pit, sizeof(int4) * kN * 4));
  HIP_OK(hipMalloc(&d_cs, sizeof(std::array<int4, 4>) * kN));
  HIP_OK(hipMalloc(&d_ocws, sizeof(int4) * kN));
  HIP_OK(hipMalloc(&d_ys1, sizeof(int4) * kN));
  HIP_OK(hipMalloc(&d_rets, sizeof(int) * kN));
  HIP_OK(hipMemset(d_ocws, 0, sizeof(int4) * kN));
  VdpfGenKernel<in_bits, Group><<<kN / 256, 256>>>(d_cws, d_cs, d_ocws, data.d_seeds, data.d_alphas, data.d_betas, d_rets);
  HIP_OK(hipDeviceSynchronize());
  fss::gpu::VdpfRelayoutGpu<in_bits, Group, fss::prg::ChaCha<2>, fss::hash::Blake3, fss::hash::Blake3, uint>(d_cws, kN, d_cw_s, d_extra);
  HIP_OK(hipDeviceSynchronize());

  fss::prg::ChaCha<2> prg(kNonce);
  fss::hash::Blake3 xor_hash{std::span<const int4, 2>(kBlake3Iv)};
  Vdpf vdpf{prg, xor_hash, xor_hash};
  fss::gpu::VdpfEvalPointGpu<1, in_bits, Group, fss::prg::ChaCha<2>, fss::hash::Blake3, fss::hash::Blake3, uint>(
      false, data.d_seeds0, d_cw_s, d_extra, d_cs, d_ocws, data.d_xs, data.d_ys, d_pi, kN, vdpf);
  HIP_OK(hipDeviceSynchronize());
  std::vector<int4> y0(kN), y1(kN), yk(kN), pi0(4 * kN), pi1(4 * kN), pik(4 * kN), ocws(kN);
  std::vector<int> rets(kN);
  ToHost(y0.data(), data.d_ys, kN);
  ToHost(pi0.data(), d_pi, 4 * kN);
  ToHost(rets.data(), d_rets, kN);
  ToHost(ocws.data(), d_ocws, kN);
  fss::gpu::VdpfEvalPointGpu(true, data.d_seeds1, d_cw_s, d_extra, d_cs, d_ocws, data.d_xs, d_ys1, d_pi, kN, vdpf);  // arguments deduced
  HIP_OK(hipDeviceSynchronize());
  ToHost(y1.data(), d_ys1, kN);
  ToHost(pi1.data(), d_pi, 4 * kN);
  // the caller's own kernel with the per-thread Eval (src/bench_gpu.cu:188-205) on the key-major keys
  VdpfEvalKernel<in_bits, Group><<<kN / 256, 256>>>(d_ys1, d_pit, false, data.d_seeds0, d_cws, d_cs, d_ocws, data.d_xs);
  HIP_OK(hipDeviceSynchronize());
  ToHost(yk.data(), d_ys1, kN);
  ToHost(pik.data(), reinterpret_cast<const int4*>(d_pit), 4 * kN);
  int bad = 0, retries = 0, bad_thread = 0;
  for (int i = 0; i < kN; ++i) {
    if (rets[i]) { ++retries; continue; }  // (t0 == t1 at the leaf: the caller would resample; 2^-1 of random keys never: Gen forces t0 != t1 off alpha only)
    const int4 want = data.h_xs[i] == data.h_alphas[i] ? Group::From(data.h_betas[i]).Into() : kZero;
    int4 wb = want;
    wb.w &= ~1;
    bad += !Same(Recon<Group>(y0[i], y1[i]), Group::From(wb).Into());
    bad += std::memcmp(&pi0[4 * i], &pi1[4 * i], 64) != 0;
    bad_thread += !Same(yk[i], y0[i]) || std::memcmp(&pik[4 * i], &pi0[4 * i], 64) != 0;
  }
  EXPECT(bad == 0);
  EXPECT(bad_thread == 0);  // header BLAKE3 + per-thread walk on the device == the engine's kernel
  EXPECT(retries < kN / 2);
  {  // the host-signature Gen / Eval on key 0's inputs give the device's bytes
    typename Vdpf::Cw cws[in_bits];
    std::array<int4, 4> cs;
    int4 ocw = kZero;
    const int ret = vdpf.Gen(cws, cs, ocw, std::span<const int4, 2>(&data.h_seeds[0], 2), data.h_alphas[0], data.h_betas[0]);
    std::vector<typename Vdpf::Cw> dev_cws(in_bits);
    ToHost(dev_cws.data(), d_cws, in_bits);
    std::array<int4, 4> dev_cs;
    ToHost(&dev_cs, d_cs, 1);
    EXPECT(ret == rets[0] && std::memcmp(cws, dev_cws.data(), sizeof cws) == 0 && std::memcmp(&cs, &dev_cs, 64) == 0);
    if (ret == 0) {
      EXPECT(Same(ocw, ocws[0]));
      int4 y;
      const std::array<int4, 4> pi = vdpf.Eval(false, data.h_seeds0[0], std::span<const typename Vdpf::Cw>(cws, in_bits), std::span<const int4, 4>(cs), ocw,
                                               data.h_xs[0], y);
      EXPECT(Same(y, y0[0]) && std::memcmp(pi.data(), &pi0[0], 64) == 0);
    }
  }
  HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_cw_s)); HIP_OK(hipFree(d_extra)); HIP_OK(hipFree(d_pi)); HIP_OK(hipFree(d_pit)); HIP_OK(hipFree(d_cs));
  HIP_OK(hipFree(d_ocws)); HIP_OK(hipFree(d_ys1)); HIP_OK(hipFree(d_rets));
  std::printf("VDPF n=%d ChaCha<2> + Blake3  VdpfRelayoutGpu / VdpfEvalPointGpu in the reference's syntax, per-thread Gen / Eval kernels (%d retries): %s\n",
              in_bits, retries, g_fail == before ? "ok" : "FAILED");
}

static void VdpfAll() {
  EXPECT(fss::util::ResolveParDepth(3) == 3 && fss::util::ResolveParDepth(0) == 0);  // util.cuh:40-45
  const int d = fss::util::ResolveParDepth(-1);
  EXPECT(d >= 0 && d <= 12 && (1u << d) >= std::max(1u, std::thread::hardware_concurrency()));
  {  // src/vdpf_test.cu:24-44: ChaCha<2> + Blake3 for both hashes, uint16_t inputs, the 127-bit group
    using Uint127Group = fss::group::Uint<__uint128_t, (static_cast<__uint128_t>(1) << 127)>;
    fss::prg::ChaCha<2> prg(kNonce);
    fss::hash::Blake3 xor_hash(std::span<const int4, 2>(kBlake3Iv, 2)), hash_(std::span<const int4, 2>(kBlake3Iv, 2));
    using V16 = fss::Vdpf<16, fss::group::Bytes, fss::prg::ChaCha<2>, fss::hash::Blake3, fss::hash::Blake3, uint16_t>;
    VdpfHostFlow(V16{prg, xor_hash, hash_}, (uint16_t)107, "n=16 ChaCha<2> + Blake3, Bytes      ");
    using V16u = fss::Vdpf<16, Uint127Group, fss::prg::ChaCha<2>, fss::hash::Blake3, fss::hash::Blake3, uint16_t>;
    VdpfHostFlow(V16u{prg, xor_hash, hash_}, (uint16_t)107, "n=16 ChaCha<2> + Blake3, Uint<u128> ");
    using V10 = fss::Vdpf<10, UintGroup, fss::prg::ChaCha<2>, fss::hash::Blake3, fss::hash::Blake3, uint16_t>;
    VdpfHostFlow(V10{prg, xor_hash, hash_}, (uint16_t)599, "n=10 ChaCha<2> + Blake3, Uint<u64>  ");
    // two different keys for H and H': one engine context cannot hold both -- refused, not silently hashed with one of them
    const int4 other[2] = {{1, 2, 3, 4}, {5, 6, 7, 8}};
    V10 two{prg, xor_hash, fss::hash::Blake3(std::span<const int4, 2>(other, 2))};
    bool refused = false;
    try { (void)two.handle(); } catch (const fss::Error& e) { refused = e.code == FSS_AMD_EINVAL; }
    EXPECT(refused);
  }
  {  // samples/vdpf_cpu.cu:22: AES-128-MMO + Sha256 for both hashes
    const unsigned char* keys2[2] = {k0, k1};
    fss::prg::Aes128Mmo<2> prg(keys2);
    const int4 hkey = I4(0x01234567, 0x89abcdef, 0x0f1e2d3c, 0x4b5a6978);
    fss::hash::Sha256 xor_hash(hkey), hash_(hkey);
    using VS = fss::Vdpf<12, fss::group::Bytes, fss::prg::Aes128Mmo<2>, fss::hash::Sha256, fss::hash::Sha256, uint16_t>;
    VdpfHostFlow(VS{prg, xor_hash, hash_}, (uint16_t)2049, "n=12 Aes128Mmo<2> + Sha256, Bytes   ");
  }
  VdpfLaunchers<20, UintGroup>();
  VdpfLaunchers<32, fss::group::Bytes>();
}
int main() {
  try {
    HostSignatureDpf();
    HostSignatureDcf();
    HostSignatureOddModulus();
    HostSignatureWide();
    const unsigned char* keys2[2] = {k0, k1};
    const unsigned char* keys4[4] = {k0, k1, k2, k3};
    const unsigned char* keys1[1] = {k0};
    DpfLaunchers<20, UintGroup>(fss::prg::Aes128Mmo<2>(keys2), "Aes128Mmo<2>");
    DpfLaunchers<12, fss::group::Bytes>(fss::prg::Aes128Mmo<2>(keys2), "Aes128Mmo<2>");
    DpfLaunchers<20, UintGroup>(fss::prg::ChaCha<2>(kNonce), "ChaCha<2>   ");
    DpfLaunchers<32, UintGroup>(fss::prg::Aes128Mmo<2>(keys2), "Aes128Mmo<2>");
    DcfLaunchers<20, UintGroup>(fss::prg::Aes128Mmo<4>(keys4), "Aes128Mmo<4>");
    DcfLaunchers<20, UintGroup>(fss::prg::ChaCha<4>(kNonce), "ChaCha<4>   ");
    HalfTreeLaunchers<20, UintGroup>(fss::prg::Aes128Mmo<1>(keys1), "Aes128Mmo<1>");
    HalfTreeLaunchers<12, UintGroup>(fss::prg::ChaCha<1>(kNonce), "ChaCha<1>   ");
    OnAStream();
    PinnedHostBuffers();
    LaunchVersusRelease();
    VdpfAll();
    fss::ReleaseEngines();
  } catch (const fss::Error& e) {
    std::fprintf(stderr, "fss::Error: %s\n", e.what());
    return 3;
  }
  std::printf(g_fail ? "FAILED (%d)\n" : "OK\n", g_fail);
  return g_fail ? 1 : 0;
}
