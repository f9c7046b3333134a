// Drop-in check of the DEVICE side of include/fss_amd.hpp: Gen / Eval called per thread from inside the caller's own
// kernels, the way myl7/fss's GPU sample and benches call them (samples/dpf_dcf_gpu.cu:39-80: `DpfPrg prg(kNonce);
// Dpf dpf{prg}; dpf.Gen(cws + tid * (kInBits + 1), s, alphas[tid], betas[tid]);` and `ys[tid] = dpf.Eval(party,
// seeds[tid], cws + tid * (kInBits + 1), xs[tid]);`; src/bench_gpu.cu:85-95).
//   part 1  ChaCha, the sample's flow: 1024 keys generated and evaluated one per thread; reconstructed, and compared
//           bit for bit with the engine's batched forms on the same inputs (GenBatch, RelayoutGpu + EvalPointGpu);
//   part 2  AES-128-MMO on the device (the reference's OpenSSL PRG traps there): scheme objects passed to kernels by
//           value and a PRG constructed inside the kernel from key bytes in __constant__ memory, against the
//           reference's known answers (DPF n=8 Bytes, DCF n=8, DPF n=32 Uint<u64>, Half-Tree n=64) and
//           against the engine on random DCF (kGt) and Half-Tree n=64 batches;
//   part 3  a caller-defined Prgable and Groupable: per thread in a kernel, and through the host signatures
//           (Gen / Eval / EvalAll / GenBatch / EvalPoints -> the generic per-thread kernels), reconstructed;
//   part 4  fss::prg::Aes128Soft / aes_detail in the reference's GPU kernel shape (src/bench_gpu.cu:100-138): tables in
//           shared memory, PRG per thread, 2^20 keys of n = 20 -- against the known answers, the engine, and timed.
// Exit code 0 and a final "OK" line on success.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
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
static const int4 kZero = {0, 0, 0, 0};

template <typename T>
static T* ToDevice(const T* host, size_t n) {
  T* dev;
  HIP_OK(hipMalloc(&dev, sizeof(T) * n));
  HIP_OK(hipMemcpy(dev, host, sizeof(T) * n, hipMemcpyHostToDevice));
  return dev;
}
template <typename T>
static T* DeviceBuf(size_t n) {
  T* dev;
  HIP_OK(hipMalloc(&dev, sizeof(T) * n));
  HIP_OK(hipMemset(dev, 0, sizeof(T) * n));
  return dev;
}
template <typename T>
static std::vector<T> ToHost(const T* dev, size_t n) {
  std::vector<T> h(n);
  HIP_OK(hipMemcpy(h.data(), dev, sizeof(T) * n, hipMemcpyDeviceToHost));
  return h;
}
static uint64_t g_rng = 0x5EED0000u + 77;
static uint64_t Next() {  // SplitMix64
  uint64_t z = (g_rng += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static int4 RandSeed() { return int4{(int)Next(), (int)Next(), (int)Next(), (int)(Next() & ~1ull)}; }

// ---------------------------------------------------------------------------------------------- part 1: ChaCha
namespace p1 {
constexpr int kN = 1024;
constexpr int kInBits = 8;
constexpr int kThreadsPerBlock = 256;
using In = uint8_t;
using Group = fss::group::Bytes;
using DpfPrg = fss::prg::ChaCha<2>;
using DcfPrg = fss::prg::ChaCha<4>;
using Dpf = fss::Dpf<kInBits, Group, DpfPrg, In>;
using Dcf = fss::Dcf<kInBits, Group, DcfPrg, In>;

__constant__ int kNonce[2] = {0x12345678, static_cast<int>(0x9abcdef0u)};
static const int kHostNonce[2] = {0x12345678, static_cast<int>(0x9abcdef0u)};

__global__ void DpfGenKernel(Dpf::Cw* cws, const int4* seeds, const In* alphas, const int4* betas) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= kN) return;
  DpfPrg prg(kNonce);
  Dpf dpf{prg};
  int4 s[2] = {seeds[tid * 2], seeds[tid * 2 + 1]};
  dpf.Gen(cws + tid * (kInBits + 1), s, alphas[tid], betas[tid]);
}
__global__ void DpfEvalKernel(int4* ys, bool party, const int4* seeds, const Dpf::Cw* cws, const In* xs) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= kN) return;
  DpfPrg prg(kNonce);
  Dpf dpf{prg};
  ys[tid] = dpf.Eval(party, seeds[tid], cws + tid * (kInBits + 1), xs[tid]);
}
__global__ void DcfGenKernel(Dcf::Cw* cws, const int4* seeds, const In* alphas, const int4* betas) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= kN) return;
  DcfPrg prg(kNonce);
  Dcf dcf{prg};
  int4 s[2] = {seeds[tid * 2], seeds[tid * 2 + 1]};
  dcf.Gen(cws + tid * (kInBits + 1), s, alphas[tid], betas[tid]);
}
__global__ void DcfEvalKernel(int4* ys, bool party, const int4* seeds, const Dcf::Cw* cws, const In* xs) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= kN) return;
  DcfPrg prg(kNonce);
  Dcf dcf{prg};
  ys[tid] = dcf.Eval(party, seeds[tid], cws + tid * (kInBits + 1), xs[tid]);
}

struct Inputs {
  std::vector<int4> seeds, seeds0, seeds1, betas;
  std::vector<In> alphas, xs;
  Inputs() : seeds(2 * kN), seeds0(kN), seeds1(kN), betas(kN), alphas(kN), xs(kN) {
    for (int i = 0; i < kN; ++i) {
      seeds0[i] = seeds[2 * i] = RandSeed();
      seeds1[i] = seeds[2 * i + 1] = RandSeed();
      alphas[i] = static_cast<In>(Next());
      betas[i] = RandSeed();
      xs[i] = i % 2 == 0 ? alphas[i] : static_cast<In>(Next());  // at alpha for even keys, random for odd ones
    }
  }
};

static void Run() {
  const int before = g_fail;
  Inputs in;
  int4* d_seeds = ToDevice(in.seeds.data(), 2 * kN);
  int4* d_seeds0 = ToDevice(in.seeds0.data(), kN);
  int4* d_seeds1 = ToDevice(in.seeds1.data(), kN);
  In* d_alphas = ToDevice(in.alphas.data(), kN);
  int4* d_betas = ToDevice(in.betas.data(), kN);
  In* d_xs = ToDevice(in.xs.data(), kN);
  int4* d_ys0 = DeviceBuf<int4>(kN);
  int4* d_ys1 = DeviceBuf<int4>(kN);
  const int blocks = (kN + kThreadsPerBlock - 1) / kThreadsPerBlock;
  std::vector<uint32_t> xs32(in.xs.begin(), in.xs.end());  // the launchers take uint32_t points
  uint32_t* d_xs32 = ToDevice(xs32.data(), kN);
  int4* d_cw_s = DeviceBuf<int4>(size_t(kInBits) * kN);
  int4* d_cw_v = DeviceBuf<int4>(size_t(kInBits) * kN);
  uint32_t* d_extra = DeviceBuf<uint32_t>(kN);
  int4* d_out_cw = DeviceBuf<int4>(kN);
  int4* d_ys_engine = DeviceBuf<int4>(kN);

  {  // DPF
    Dpf::Cw* d_cws = DeviceBuf<Dpf::Cw>(size_t(kInBits + 1) * kN);
    DpfGenKernel<<<blocks, kThreadsPerBlock>>>(d_cws, d_seeds, d_alphas, d_betas);
    HIP_OK(hipGetLastError());
    DpfEvalKernel<<<blocks, kThreadsPerBlock>>>(d_ys0, false, d_seeds0, d_cws, d_xs);
    DpfEvalKernel<<<blocks, kThreadsPerBlock>>>(d_ys1, true, d_seeds1, d_cws, d_xs);
    HIP_OK(hipDeviceSynchronize());
    auto ys0 = ToHost(d_ys0, kN), ys1 = ToHost(d_ys1, kN);
    int bad = 0;
    for (int i = 0; i < kN; ++i) {
      int4 want = in.xs[i] == in.alphas[i] ? in.betas[i] : kZero;
      if (!Same(Recon<Group>(ys0[i], ys1[i]), want)) ++bad;
    }
    EXPECT(bad == 0);
    // the engine on the same inputs: keys, then the level-major launchers
    DpfPrg prg(kHostNonce);
    Dpf dpf{prg};
    Dpf::Cw* d_cws_engine = DeviceBuf<Dpf::Cw>(size_t(kInBits + 1) * kN);
    dpf.GenBatch(d_cws_engine, d_seeds, d_alphas, d_betas, kN);
    auto a = ToHost(d_cws, size_t(kInBits + 1) * kN), e = ToHost(d_cws_engine, size_t(kInBits + 1) * kN);
    bad = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      const bool last = i % (kInBits + 1) == kInBits;
      if (!Same(a[i].s, e[i].s) || (!last && a[i].tr != e[i].tr)) ++bad;
    }
    EXPECT(bad == 0);
    fss::gpu::DpfRelayoutGpu<kInBits, Group, DpfPrg, In>(d_cws, kN, d_cw_s, d_extra, d_out_cw);
    fss::gpu::DpfEvalPointGpu<4>(false, d_seeds0, d_cw_s, d_extra, d_out_cw, d_xs32, d_ys_engine, kN, dpf);
    HIP_OK(hipDeviceSynchronize());
    auto ye = ToHost(d_ys_engine, kN);
    bad = 0;
    for (int i = 0; i < kN; ++i) bad += !Same(ye[i], ys0[i]);
    EXPECT(bad == 0);
    HIP_OK(hipFree(d_cws));
    HIP_OK(hipFree(d_cws_engine));
  }
  {  // DCF
    Dcf::Cw* d_cws = DeviceBuf<Dcf::Cw>(size_t(kInBits + 1) * kN);
    DcfGenKernel<<<blocks, kThreadsPerBlock>>>(d_cws, d_seeds, d_alphas, d_betas);
    HIP_OK(hipGetLastError());
    DcfEvalKernel<<<blocks, kThreadsPerBlock>>>(d_ys0, false, d_seeds0, d_cws, d_xs);
    DcfEvalKernel<<<blocks, kThreadsPerBlock>>>(d_ys1, true, d_seeds1, d_cws, d_xs);
    HIP_OK(hipDeviceSynchronize());
    auto ys0 = ToHost(d_ys0, kN), ys1 = ToHost(d_ys1, kN);
    int bad = 0;
    for (int i = 0; i < kN; ++i) {
      int4 want = in.xs[i] < in.alphas[i] ? in.betas[i] : kZero;
      if (!Same(Recon<Group>(ys0[i], ys1[i]), want)) ++bad;
    }
    EXPECT(bad == 0);
    DcfPrg prg(kHostNonce);
    Dcf dcf{prg};
    Dcf::Cw* d_cws_engine = DeviceBuf<Dcf::Cw>(size_t(kInBits + 1) * kN);
    dcf.GenBatch(d_cws_engine, d_seeds, d_alphas, d_betas, kN);
    auto a = ToHost(d_cws, size_t(kInBits + 1) * kN), e = ToHost(d_cws_engine, size_t(kInBits + 1) * kN);
    bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += !Same(a[i].s, e[i].s) || !Same(a[i].v, e[i].v);
    EXPECT(bad == 0);
    fss::gpu::DcfRelayoutGpu<kInBits, Group, DcfPrg, In>(d_cws, kN, d_cw_s, d_cw_v, d_out_cw);
    fss::gpu::DcfEvalPointGpu<4>(true, d_seeds1, d_cw_s, d_cw_v, d_out_cw, d_xs32, d_ys_engine, kN, dcf);
    HIP_OK(hipDeviceSynchronize());
    auto ye = ToHost(d_ys_engine, kN);
    bad = 0;
    for (int i = 0; i < kN; ++i) bad += !Same(ye[i], ys1[i]);
    EXPECT(bad == 0);
    HIP_OK(hipFree(d_cws));
    HIP_OK(hipFree(d_cws_engine));
  }
  std::printf("ChaCha  n=8  Bytes   per-thread Gen/Eval in caller kernels (1024 keys) == engine, reconstructs: %s\n",
              g_fail == before ? "ok" : "FAILED");
}
}  // namespace p1

// ---------------------------------------------------------------------------------------------- part 2: AES-128-MMO
namespace p2 {
static unsigned char k0[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static unsigned char k1[16] = {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1};
static unsigned char k2[16] = {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8};
static unsigned char k3[16] = {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1};
__constant__ unsigned char kDevKeys[2][16] = {{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16},
                                              {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1}};
static const int4 kSeeds[2] = {{0x11111111, 0x22222222, 0x33333333, 0x44444440},
                               {0x55555555, 0x66666666, 0x77777777, static_cast<int>(0x88888880u)}};

// one key, generated and evaluated by thread 0 with the scheme object handed over by value
template <typename S, typename In>
__global__ void GenEvalKernel(S sch, typename S::Cw* cws, const int4* s0s, In alpha, int4 beta, const In* xs, int nx,
                              int4* ys /* [2][nx] */) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  sch.Gen(cws, s0s, alpha, beta);
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < nx; ++i) ys[b * nx + i] = sch.Eval(b, s0s[b], cws, xs[i]);
}
template <typename S, typename In>
__global__ void HtGenEvalKernel(S sch, typename S::Cw* cws, int4* ocw, const int4* s0s, In alpha, int4 beta, const In* xs,
                                int nx, int4* ys) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  sch.Gen(cws, *ocw, s0s, alpha, beta);
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < nx; ++i) ys[b * nx + i] = sch.Eval(b, s0s[b], cws, *ocw, xs[i]);
}
// the PRG built inside the kernel from key bytes in constant memory
using Dpf8 = fss::Dpf<8, fss::group::Bytes, fss::prg::Aes128Mmo<2>, uint8_t>;
__global__ void InKernelPrgKernel(int4* prg_out, int4* y, const int4* s0s, const Dpf8::Cw* cws) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const unsigned char* keys[2] = {kDevKeys[0], kDevKeys[1]};
  fss::prg::Aes128Mmo<2> prg(keys);
  auto out = prg.Gen(s0s[0]);
  prg_out[0] = out[0];
  prg_out[1] = out[1];
  Dpf8 dpf{prg};
  *y = dpf.Eval(false, s0s[0], cws, static_cast<uint8_t>(42));
}

static void Kats() {
  const int before = g_fail;
  int4* d_s0s = ToDevice(kSeeds, 2);
  {  // C1: Dpf<8, Bytes, Aes128Mmo<2>, uint8_t>, alpha 42, beta {7,0,0,0}
    const unsigned char* keys[2] = {k0, k1};
    auto ctxs = fss::prg::Aes128Mmo<2>::CreateCtxs(keys);
    fss::prg::Aes128Mmo<2> prg(ctxs);
    Dpf8 dpf{prg};
    const uint8_t xs[2] = {42, 100};
    uint8_t* d_xs = ToDevice(xs, 2);
    Dpf8::Cw* d_cws = DeviceBuf<Dpf8::Cw>(9);
    int4* d_ys = DeviceBuf<int4>(4);
    GenEvalKernel<Dpf8, uint8_t><<<1, 64>>>(dpf, d_cws, d_s0s, uint8_t(42), int4{7, 0, 0, 0}, d_xs, 2, d_ys);
    HIP_OK(hipDeviceSynchronize());
    auto cws = ToHost(d_cws, 9);
    auto ys = ToHost(d_ys, 4);
    EXPECT(Same(cws[0].s, I4(0x84863059, 0xb57c0062, 0xc0a8015f, 0x3520c9e8)) && cws[0].tr == true);
    EXPECT(Same(cws[8].s, I4(0x0b17cf59, 0xcd54e225, 0xef2cd79f, 0x30ad9cf8)));
    EXPECT(Same(ys[0], I4(0xa81892ad, 0xad2e583e, 0x272d1483, 0xd2bfa094)));
    EXPECT(Same(ys[2], I4(0xa81892aa, 0xad2e583e, 0x272d1483, 0xd2bfa094)));
    EXPECT(Same(ys[1], I4(0x7e2be32f, 0x4508eb0c, 0x537f8e89, 0x39b1bd96)));
    // the host-side (engine) members on the key the device thread generated, and the other way round
    Dpf8::Cw hcws[9];
    int4 seeds[2] = {kSeeds[0], kSeeds[1]};
    dpf.Gen(hcws, seeds, uint8_t(42), int4{7, 0, 0, 0});
    int bad = 0;
    for (int i = 0; i < 9; ++i) bad += !Same(hcws[i].s, cws[i].s) || (i < 8 && hcws[i].tr != cws[i].tr);
    EXPECT(bad == 0);
    EXPECT(Same(dpf.Eval(false, kSeeds[0], cws.data(), uint8_t(100)), ys[1]));
    // PRG constructed inside a kernel
    int4* d_prg = DeviceBuf<int4>(2);
    int4* d_y = DeviceBuf<int4>(1);
    InKernelPrgKernel<<<1, 64>>>(d_prg, d_y, d_s0s, d_cws);
    HIP_OK(hipDeviceSynchronize());
    auto po = ToHost(d_prg, 2);
    EXPECT(Same(po[0], I4(0x1423e6d2, 0x60533602, 0x813b3fbc, 0x412b31dc)));
    EXPECT(Same(po[1], I4(0x2d18cbe7, 0xa5eddcc9, 0xc691e4f6, 0x97831704)));
    EXPECT(Same(ToHost(d_y, 1)[0], ys[0]));
    fss::prg::Aes128Mmo<2>::FreeCtxs(ctxs);
  }
  {  // Dcf<8, Bytes, Aes128Mmo<4>, uint8_t, kLt>
    using Dcf = fss::Dcf<8, fss::group::Bytes, fss::prg::Aes128Mmo<4>, uint8_t>;
    const unsigned char* keys[4] = {k0, k1, k2, k3};
    fss::prg::Aes128Mmo<4> prg(keys);
    Dcf dcf{prg};
    const uint8_t xs[1] = {10};
    uint8_t* d_xs = ToDevice(xs, 1);
    Dcf::Cw* d_cws = DeviceBuf<Dcf::Cw>(9);
    int4* d_ys = DeviceBuf<int4>(2);
    GenEvalKernel<Dcf, uint8_t><<<1, 64>>>(dcf, d_cws, d_s0s, uint8_t(42), int4{7, 0, 0, 0}, d_xs, 1, d_ys);
    HIP_OK(hipDeviceSynchronize());
    auto cws = ToHost(d_cws, 9);
    auto ys = ToHost(d_ys, 2);
    EXPECT(Same(cws[0].s, I4(0x79f28305, 0x7b1a2718, 0xc9874166, 0xfe3afd02)));
    EXPECT(Same(cws[0].v, I4(0x0b706acf, 0xe72e35b7, 0x1984be09, 0x3d5a09d1)));
    EXPECT(Same(ys[0], I4(0x3a28e8cb, 0x5f11990b, 0xf2fee90f, 0xa7604478)));
    EXPECT(Same(ys[1], I4(0x3a28e8cc, 0x5f11990b, 0xf2fee90f, 0xa7604478)));
  }
  {  // Dpf<32, Uint<u64>, Aes128Mmo<2>, uint32_t>, alpha 0x12345678
    using Dpf = fss::Dpf<32, fss::group::Uint<uint64_t>, fss::prg::Aes128Mmo<2>, uint32_t>;
    const unsigned char* keys[2] = {k0, k1};
    fss::prg::Aes128Mmo<2> prg(keys);
    Dpf dpf{prg};
    const uint32_t xs[2] = {0x12345678u, 0xdeadbeefu};
    uint32_t* d_xs = ToDevice(xs, 2);
    Dpf::Cw* d_cws = DeviceBuf<Dpf::Cw>(33);
    int4* d_ys = DeviceBuf<int4>(4);
    GenEvalKernel<Dpf, uint32_t><<<1, 64>>>(dpf, d_cws, d_s0s, 0x12345678u, int4{7, 0, 0, 0}, d_xs, 2, d_ys);
    HIP_OK(hipDeviceSynchronize());
    auto cws = ToHost(d_cws, 33);
    auto ys = ToHost(d_ys, 4);
    EXPECT(Same(cws[32].s, I4(0xd316e94a, 0x28e6b72d, 0, 0)));
    EXPECT(Same(ys[0], I4(0x3ffa817c, 0xf3ec1fb3, 0, 0)));
    EXPECT(Same(ys[2], I4(0xc0057e8b, 0x0c13e04c, 0, 0)));
    EXPECT(Same(ys[1], I4(0xa840848f, 0x144d9ccf, 0, 0)));
  }
  {  // HalfTreeDpf<64, Uint<u64>, Aes128Mmo<1>, uint64_t>, alpha 0x0123456789abcdef
    using Ht = fss::HalfTreeDpf<64, fss::group::Uint<uint64_t>, fss::prg::Aes128Mmo<1>, uint64_t>;
    const unsigned char* keys[1] = {k0};
    fss::prg::Aes128Mmo<1> prg(keys);
    Ht ht{prg, int4{0x12345678, static_cast<int>(0x9abcdef0u), 0x13572468, 0x2468ace0}};
    const uint64_t xs[2] = {0x0123456789abcdefull, 0xdeadbeefcafef00dull};
    uint64_t* d_xs = ToDevice(xs, 2);
    Ht::Cw* d_cws = DeviceBuf<Ht::Cw>(64);
    int4* d_ocw = DeviceBuf<int4>(1);
    int4* d_ys = DeviceBuf<int4>(4);
    HtGenEvalKernel<Ht, uint64_t><<<1, 64>>>(ht, d_cws, d_ocw, d_s0s, xs[0], int4{7, 0, 0, 0}, d_xs, 2, d_ys);
    HIP_OK(hipDeviceSynchronize());
    auto cws = ToHost(d_cws, 64);
    auto ys = ToHost(d_ys, 4);
    EXPECT(Same(cws[0].s, I4(0x02fc4a65, 0xf8fecae6, 0xf249f101, 0x66111f80)));
    EXPECT(cws[63].extra == true);
    EXPECT(Same(ToHost(d_ocw, 1)[0], I4(0x7c3c2870, 0x598af3a9, 0, 0)));
    EXPECT(Same(ys[0], I4(0x98317401, 0xbf535b1d, 0, 0)));
    EXPECT(Same(ys[2], I4(0x67ce8c06, 0x40aca4e2, 0, 0)));
    EXPECT(Same(ys[1], I4(0xe7bf5193, 0x17c707b4, 0, 0)));
  }
  std::printf("AES-128-MMO  per-thread Gen/Eval on the device == the reference's known answers (4 schemes): %s\n",
              g_fail == before ? "ok" : "FAILED");
}

// random batches: one key per thread vs the engine's level-major launchers
template <typename S, typename In>
__global__ void GenKeysKernel(S sch, typename S::Cw* cws, const int4* seeds, const In* alphas, const int4* betas, int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  int4 s[2] = {seeds[tid * 2], seeds[tid * 2 + 1]};
  sch.Gen(cws + size_t(tid) * S::kSlots, s, alphas[tid], betas[tid]);
}
template <typename S, typename In>
__global__ void EvalKeysKernel(S sch, int4* ys, bool party, const int4* seeds, const typename S::Cw* cws, const In* xs,
                               int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  ys[tid] = sch.Eval(party, seeds[tid], cws + size_t(tid) * S::kSlots, xs[tid]);
}
template <typename S, typename In>
__global__ void HtGenKeysKernel(S sch, typename S::Cw* cws, int4* ocws, const int4* seeds, const In* alphas,
                                const int4* betas, int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  int4 s[2] = {seeds[tid * 2], seeds[tid * 2 + 1]};
  sch.Gen(cws + size_t(tid) * S::kSlots, ocws[tid], s, alphas[tid], betas[tid]);
}
template <typename S, typename In>
__global__ void HtEvalKeysKernel(S sch, int4* ys, bool party, const int4* seeds, const typename S::Cw* cws,
                                 const int4* ocws, const In* xs, int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  ys[tid] = sch.Eval(party, seeds[tid], cws + size_t(tid) * S::kSlots, ocws[tid], xs[tid]);
}

// the scheme reached through a pointer that all waves of the workgroup share (a __shared__ copy made by thread 0): the
// per-wave LDS copies of the AES tables and of a four-key PRG's round keys must be found missing -- and filled -- by
// every wave that arrives, whoever constructed or first used the object
template <typename S, typename In>
__global__ void EvalThroughSharedPointerKernel(S sch, int4* ys, bool party, const int4* seeds, const typename S::Cw* cws,
                                               const In* xs, int n) {
  __shared__ alignas(16) unsigned char raw[sizeof(S)];
  if (threadIdx.x == 0) new (raw) S(sch);
  __syncthreads();
  const S* shared = reinterpret_cast<const S*>(raw);
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  ys[tid] = shared->Eval(party, seeds[tid], cws + size_t(tid) * S::kSlots, xs[tid]);
}

// the first use of the scheme inside divergent code: the odd lanes of every wave evaluate (and fill their wave's LDS
// copies of the tables and round keys) while the even ones wait, then the even lanes arrive at their own call site
template <typename S, typename In>
__global__ void EvalDivergentKernel(S sch, int4* ys, bool party, const int4* seeds, const typename S::Cw* cws, const In* xs,
                                    int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  if (tid & 1) {
    ys[tid] = sch.Eval(party, seeds[tid], cws + size_t(tid) * S::kSlots, xs[tid]);
  } else {
    if (tid % 6 == 0) return;  // some lanes never call at all
    ys[tid] = sch.Eval(party, seeds[tid], cws + size_t(tid) * S::kSlots, xs[tid]);
  }
}

// two key sets inside one wave, the cases the wave's LDS key slot (two / four schedules, fss_amd_thread.hpp) has to get
// right: (a) lanes that construct their PRG from DIFFERENT keys (odd lanes set B, even lanes set A: no wave-uniform
// schedule exists, every lane expands its own), (b) one thread using two PRG objects in turn (A, B, A: the slot's tag
// names the other set each time and the schedules are re-derived).
__constant__ unsigned char kSetA[4][16] = {{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16},
                                           {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1},
                                           {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8},
                                           {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1}};
__constant__ unsigned char kSetB[4][16] = {{8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1},
                                           {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8},
                                           {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1},
                                           {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16}};
template <typename S, typename Prg, typename In>
__global__ void PerLaneKeySetKernel(int4* ys, bool party, const int4* seeds, const typename S::Cw* cws_a,
                                    const typename S::Cw* cws_b, const In* xs, int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  const bool odd = tid & 1;
  const unsigned char* keys[Prg::kMul];
  for (int i = 0; i < Prg::kMul; ++i) keys[i] = odd ? kSetB[i] : kSetA[i];
  S sch{Prg(keys)};
  ys[tid] = sch.Eval(party, seeds[tid], (odd ? cws_b : cws_a) + size_t(tid) * S::kSlots, xs[tid]);
}
template <typename S, typename In>
__global__ void AlternatingKeySetKernel(S sch_a, S sch_b, int4* ys /* [3][n] */, bool party, const int4* seeds,
                                        const typename S::Cw* cws_a, const typename S::Cw* cws_b, const In* xs, int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  ys[tid] = sch_a.Eval(party, seeds[tid], cws_a + size_t(tid) * S::kSlots, xs[tid]);
  ys[n + tid] = sch_b.Eval(party, seeds[tid], cws_b + size_t(tid) * S::kSlots, xs[tid]);
  ys[2 * n + tid] = sch_a.Eval(party, seeds[tid], cws_a + size_t(tid) * S::kSlots, xs[tid]);
}
// Two HOST-constructed schemes (each `uniform` by construction) and a per-lane choice between them: the lanes of one wave
// then hold different key sets although every object says its keys are wave-uniform -- one LDS key slot cannot serve both,
// the cipher has to notice (the tag is not the same in all active lanes) and expand per lane.  Also: the objects built in
// divergent branches on the device and used after the branches have met again.
template <typename S, typename In>
__global__ void ChooseSchemePerLaneKernel(S sch_a, S sch_b, int4* ys, bool party, const int4* seeds, const typename S::Cw* cws_a,
                                          const typename S::Cw* cws_b, const In* xs, int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  const bool odd = tid & 1;
  const S& sch = odd ? sch_b : sch_a;
  ys[tid] = sch.Eval(party, seeds[tid], (odd ? cws_b : cws_a) + size_t(tid) * S::kSlots, xs[tid]);
}
template <typename S, typename Prg, typename In>
__global__ void BuiltInBranchesKernel(int4* ys, bool party, const int4* seeds, const typename S::Cw* cws_a, const typename S::Cw* cws_b,
                                      const In* xs, int n) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  const bool odd = tid & 1;
  const unsigned char* keys[Prg::kMul];
  for (int i = 0; i < Prg::kMul; ++i) keys[i] = kSetA[i];
  Prg prg(keys);
  if (odd) {  // each branch sees only lanes with ONE key set: both constructions find their keys "uniform"
    for (int i = 0; i < Prg::kMul; ++i) keys[i] = kSetB[i];
    prg = Prg(keys);
  }
  S sch{prg};
  ys[tid] = sch.Eval(party, seeds[tid], (odd ? cws_b : cws_a) + size_t(tid) * S::kSlots, xs[tid]);
}
template <typename S, typename Prg>
static void KeySetCase(const char* name) {
  const int before = g_fail;
  constexpr int kN = 1024;
  const unsigned char* ka[4] = {k0, k1, k2, k3};
  const unsigned char* kb[4] = {k3, k2, k1, k0};
  S sch_a{Prg(ka)}, sch_b{Prg(kb)};
  std::vector<int4> seeds(2 * kN), seeds0(kN), betas(kN);
  std::vector<uint16_t> alphas(kN), xs(kN);
  for (int i = 0; i < kN; ++i) {
    seeds0[i] = seeds[2 * i] = RandSeed();
    seeds[2 * i + 1] = RandSeed();
    betas[i] = RandSeed();
    alphas[i] = static_cast<uint16_t>(Next());
    xs[i] = i % 3 == 0 ? alphas[i] : static_cast<uint16_t>(Next());
  }
  int4* d_seeds = ToDevice(seeds.data(), 2 * kN);
  int4* d_seeds0 = ToDevice(seeds0.data(), kN);
  int4* d_betas = ToDevice(betas.data(), kN);
  uint16_t* d_alphas = ToDevice(alphas.data(), kN);
  uint16_t* d_xs = ToDevice(xs.data(), kN);
  typename S::Cw* d_ca = DeviceBuf<typename S::Cw>(size_t(S::kSlots) * kN);
  typename S::Cw* d_cb = DeviceBuf<typename S::Cw>(size_t(S::kSlots) * kN);
  int4* d_ya = DeviceBuf<int4>(kN);
  int4* d_yb = DeviceBuf<int4>(kN);
  int4* d_mixed = DeviceBuf<int4>(kN);
  int4* d_turns = DeviceBuf<int4>(3 * kN);
  int4* d_chosen = DeviceBuf<int4>(kN);
  int4* d_branch = DeviceBuf<int4>(kN);
  sch_a.GenBatch(d_ca, d_seeds, d_alphas, d_betas, kN);  // the engine's keys and (uniform-key kernels) reference shares
  sch_b.GenBatch(d_cb, d_seeds, d_alphas, d_betas, kN);
  EvalKeysKernel<S, uint16_t><<<kN / 128, 128>>>(sch_a, d_ya, false, d_seeds0, d_ca, d_xs, kN);
  EvalKeysKernel<S, uint16_t><<<kN / 128, 128>>>(sch_b, d_yb, false, d_seeds0, d_cb, d_xs, kN);
  PerLaneKeySetKernel<S, Prg, uint16_t><<<kN / 128, 128>>>(d_mixed, false, d_seeds0, d_ca, d_cb, d_xs, kN);
  AlternatingKeySetKernel<S, uint16_t><<<kN / 128, 128>>>(sch_a, sch_b, d_turns, false, d_seeds0, d_ca, d_cb, d_xs, kN);
  ChooseSchemePerLaneKernel<S, uint16_t><<<kN / 128, 128>>>(sch_a, sch_b, d_chosen, false, d_seeds0, d_ca, d_cb, d_xs, kN);
  BuiltInBranchesKernel<S, Prg, uint16_t><<<kN / 128, 128>>>(d_branch, false, d_seeds0, d_ca, d_cb, d_xs, kN);
  HIP_OK(hipDeviceSynchronize());
  auto ya = ToHost(d_ya, kN), yb = ToHost(d_yb, kN), mixed = ToHost(d_mixed, kN), turns = ToHost(d_turns, 3 * kN);
  auto chosen = ToHost(d_chosen, kN), branch = ToHost(d_branch, kN);
  int differ = 0, bad_mixed = 0, bad_turns = 0, bad_chosen = 0, bad_branch = 0;
  for (int i = 0; i < kN; ++i) {
    differ += !Same(ya[i], yb[i]);
    bad_chosen += !Same(chosen[i], i & 1 ? yb[i] : ya[i]);
    bad_branch += !Same(branch[i], i & 1 ? yb[i] : ya[i]);
    bad_mixed += !Same(mixed[i], i & 1 ? yb[i] : ya[i]);
    bad_turns += !Same(turns[i], ya[i]) || !Same(turns[kN + i], yb[i]) || !Same(turns[2 * kN + i], ya[i]);
  }
  EXPECT(differ > kN / 2);  // the two key sets really give different shares
  EXPECT(bad_mixed == 0);
  EXPECT(bad_turns == 0);
  EXPECT(bad_chosen == 0);  // two host-built schemes, one picked per lane
  EXPECT(bad_branch == 0);  // PRGs built in divergent branches, used after they met again
  for (void* q : {(void*)d_seeds, (void*)d_seeds0, (void*)d_betas, (void*)d_alphas, (void*)d_xs, (void*)d_ca, (void*)d_cb, (void*)d_ya,
                  (void*)d_yb, (void*)d_mixed, (void*)d_turns, (void*)d_chosen, (void*)d_branch})
    HIP_OK(hipFree(q));
  std::printf("AES-128-MMO  %s: key sets differing per lane, and alternating in one thread == the uniform-key kernels: %s\n", name,
              g_fail == before ? "ok" : "FAILED");
}
static void KeySets() {
  using G = fss::group::Uint<uint64_t>;
  KeySetCase<fss::Dpf<16, G, fss::prg::Aes128Mmo<2>, uint16_t>, fss::prg::Aes128Mmo<2>>("DPF n=16 (two schedules)");
  KeySetCase<fss::Dcf<16, G, fss::prg::Aes128Mmo<4>, uint16_t>, fss::prg::Aes128Mmo<4>>("DCF n=16 (four schedules)");
}

static void Batches() {
  const int before = g_fail;
  constexpr int kN = 4096;
  std::vector<int4> seeds(2 * kN), seeds0(kN), seeds1(kN), betas(kN);
  for (int i = 0; i < kN; ++i) {
    seeds0[i] = seeds[2 * i] = RandSeed();
    seeds1[i] = seeds[2 * i + 1] = RandSeed();
    betas[i] = RandSeed();
  }
  int4* d_seeds = ToDevice(seeds.data(), 2 * kN);
  int4* d_seeds0 = ToDevice(seeds0.data(), kN);
  int4* d_seeds1 = ToDevice(seeds1.data(), kN);
  int4* d_betas = ToDevice(betas.data(), kN);
  int4* d_ys0 = DeviceBuf<int4>(kN);
  int4* d_ys1 = DeviceBuf<int4>(kN);
  int4* d_ye = DeviceBuf<int4>(kN);
  {  // DCF n=32, Uint<u64>, kGt
    using G = fss::group::Uint<uint64_t>;
    using Prg = fss::prg::Aes128Mmo<4>;
    using Dcf = fss::Dcf<32, G, Prg, uint32_t, fss::DcfPred::kGt>;
    const unsigned char* keys[4] = {k0, k1, k2, k3};
    Prg prg(keys);
    Dcf dcf{prg};
    std::vector<uint32_t> alphas(kN), xs(kN);
    for (int i = 0; i < kN; ++i) {
      alphas[i] = static_cast<uint32_t>(Next());
      xs[i] = i % 3 == 0 ? alphas[i] : i % 3 == 1 ? alphas[i] + 1 + static_cast<uint32_t>(Next() % 1000) : static_cast<uint32_t>(Next());
    }
    uint32_t* d_alphas = ToDevice(alphas.data(), kN);
    uint32_t* d_xs = ToDevice(xs.data(), kN);
    Dcf::Cw* d_cws = DeviceBuf<Dcf::Cw>(size_t(33) * kN);
    GenKeysKernel<Dcf, uint32_t><<<kN / 128, 128>>>(dcf, d_cws, d_seeds, d_alphas, d_betas, kN);
    EvalKeysKernel<Dcf, uint32_t><<<kN / 128, 128>>>(dcf, d_ys0, false, d_seeds0, d_cws, d_xs, kN);
    EvalKeysKernel<Dcf, uint32_t><<<kN / 128, 128>>>(dcf, d_ys1, true, d_seeds1, d_cws, d_xs, kN);
    HIP_OK(hipDeviceSynchronize());
    auto ys0 = ToHost(d_ys0, kN), ys1 = ToHost(d_ys1, kN);
    int bad = 0;
    for (int i = 0; i < kN; ++i) {
      int4 want = xs[i] > alphas[i] ? G::From(betas[i]).Into() : kZero;
      bad += !Same(Recon<G>(ys0[i], ys1[i]), want);
    }
    EXPECT(bad == 0);
    Dcf::Cw* d_cws_e = DeviceBuf<Dcf::Cw>(size_t(33) * kN);
    dcf.GenBatch(d_cws_e, d_seeds, d_alphas, d_betas, kN);
    auto a = ToHost(d_cws, size_t(33) * kN), e = ToHost(d_cws_e, size_t(33) * kN);
    bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += !Same(a[i].s, e[i].s) || !Same(a[i].v, e[i].v);
    EXPECT(bad == 0);
    int4* cw_s = DeviceBuf<int4>(size_t(32) * kN);
    int4* cw_v = DeviceBuf<int4>(size_t(32) * kN);
    int4* out_cw = DeviceBuf<int4>(kN);
    fss::gpu::DcfRelayoutGpu<32, G, Prg, uint32_t, fss::DcfPred::kGt>(d_cws, kN, cw_s, cw_v, out_cw);
    fss::gpu::DcfEvalPointGpu<4>(false, d_seeds0, cw_s, cw_v, out_cw, d_xs, d_ye, kN, dcf);
    HIP_OK(hipDeviceSynchronize());
    auto ye = ToHost(d_ye, kN);
    bad = 0;
    for (int i = 0; i < kN; ++i) bad += !Same(ye[i], ys0[i]);
    EXPECT(bad == 0);
    // the same evaluations with the scheme shared between the four waves of a workgroup through a pointer
    HIP_OK(hipMemset(d_ye, 0xff, sizeof(int4) * kN));
    EvalThroughSharedPointerKernel<Dcf, uint32_t><<<kN / 256, 256>>>(dcf, d_ye, false, d_seeds0, d_cws, d_xs, kN);
    HIP_OK(hipDeviceSynchronize());
    ye = ToHost(d_ye, kN);
    bad = 0;
    for (int i = 0; i < kN; ++i) bad += !Same(ye[i], ys0[i]);
    EXPECT(bad == 0);
    // ... and with the first call made from divergent code (lanes that never call keep their 0xff fill)
    HIP_OK(hipMemset(d_ye, 0xff, sizeof(int4) * kN));
    EvalDivergentKernel<Dcf, uint32_t><<<kN / 192, 192>>>(dcf, d_ye, false, d_seeds0, d_cws, d_xs, kN - 64);
    HIP_OK(hipDeviceSynchronize());
    ye = ToHost(d_ye, kN);
    bad = 0;
    const int4 untouched = int4{-1, -1, -1, -1};
    for (int i = 0; i < kN; ++i) {
      const bool called = i < kN - 64 && ((i & 1) || i % 6 != 0);
      bad += !Same(ye[i], called ? ys0[i] : untouched);
    }
    EXPECT(bad == 0);
  }
  {  // Half-Tree n=64, Uint<u64>
    using G = fss::group::Uint<uint64_t>;
    using Prg = fss::prg::Aes128Mmo<1>;
    using Ht = fss::HalfTreeDpf<64, G, Prg, uint64_t>;
    const unsigned char* keys[1] = {k0};
    Prg prg(keys);
    Ht ht{prg, int4{0x12345678, static_cast<int>(0x9abcdef0u), 0x13572468, 0x2468ace0}};
    std::vector<uint64_t> alphas(kN), xs(kN);
    for (int i = 0; i < kN; ++i) {
      alphas[i] = Next();
      xs[i] = i % 2 == 0 ? alphas[i] : (i % 4 == 1 ? alphas[i] ^ 1 : Next());
    }
    uint64_t* d_alphas = ToDevice(alphas.data(), kN);
    uint64_t* d_xs = ToDevice(xs.data(), kN);
    Ht::Cw* d_cws = DeviceBuf<Ht::Cw>(size_t(64) * kN);
    int4* d_ocws = DeviceBuf<int4>(kN);
    HtGenKeysKernel<Ht, uint64_t><<<kN / 128, 128>>>(ht, d_cws, d_ocws, d_seeds, d_alphas, d_betas, kN);
    HtEvalKeysKernel<Ht, uint64_t><<<kN / 128, 128>>>(ht, d_ys0, false, d_seeds0, d_cws, d_ocws, d_xs, kN);
    HtEvalKeysKernel<Ht, uint64_t><<<kN / 128, 128>>>(ht, d_ys1, true, d_seeds1, d_cws, d_ocws, d_xs, kN);
    HIP_OK(hipDeviceSynchronize());
    auto ys0 = ToHost(d_ys0, kN), ys1 = ToHost(d_ys1, kN);
    int bad = 0;
    for (int i = 0; i < kN; ++i) {
      int4 want = xs[i] == alphas[i] ? G::From(betas[i]).Into() : kZero;
      bad += !Same(Recon<G>(ys0[i], ys1[i]), want);
    }
    EXPECT(bad == 0);
    Ht::Cw* d_cws_e = DeviceBuf<Ht::Cw>(size_t(64) * kN);
    int4* d_ocws_e = DeviceBuf<int4>(kN);
    ht.GenBatch(d_cws_e, d_ocws_e, d_seeds, d_alphas, d_betas, kN);
    auto a = ToHost(d_cws, size_t(64) * kN), e = ToHost(d_cws_e, size_t(64) * kN);
    bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += !Same(a[i].s, e[i].s) || a[i].extra != e[i].extra;
    auto oa = ToHost(d_ocws, kN), oe = ToHost(d_ocws_e, kN);
    for (int i = 0; i < kN; ++i) bad += !Same(oa[i], oe[i]);
    EXPECT(bad == 0);
    int4* cw_s = DeviceBuf<int4>(size_t(64) * kN);
    uint32_t* extra = DeviceBuf<uint32_t>(kN);
    fss::gpu::HalfTreeDpfRelayoutGpu<64, G, Prg, uint64_t>(d_cws, kN, cw_s, extra);
    fss::gpu::HalfTreeDpfEvalPointGpu<4>(true, d_seeds1, cw_s, extra, d_ocws, d_xs, d_ye, kN, ht);
    HIP_OK(hipDeviceSynchronize());
    auto ye = ToHost(d_ye, kN);
    bad = 0;
    for (int i = 0; i < kN; ++i) bad += !Same(ye[i], ys1[i]);
    EXPECT(bad == 0);
  }
  std::printf("AES-128-MMO  DCF n=32 kGt, Half-Tree n=64: 4096 keys per thread == engine (keys and shares): %s\n",
              g_fail == before ? "ok" : "FAILED");
}

// every built-in group through the per-thread walk: 1024 DPF keys, n = 16, against the engine's keys and shares
template <typename G>
static void GroupCase(const char* name) {
  const int before = g_fail;
  constexpr int kN = 1024, kBits = 16;
  using Prg = fss::prg::Aes128Mmo<2>;
  using Dpf = fss::Dpf<kBits, G, Prg, uint16_t>;
  const unsigned char* keys[2] = {k0, k1};
  Prg prg(keys);
  Dpf dpf{prg};
  std::vector<int4> seeds(2 * kN), seeds0(kN), seeds1(kN), betas(kN);
  std::vector<uint16_t> alphas(kN), xs(kN);
  std::vector<uint32_t> xs32(kN);
  for (int i = 0; i < kN; ++i) {
    seeds0[i] = seeds[2 * i] = RandSeed();
    seeds1[i] = seeds[2 * i + 1] = RandSeed();
    betas[i] = RandSeed();
    alphas[i] = static_cast<uint16_t>(Next());
    xs32[i] = xs[i] = i % 2 ? alphas[i] : static_cast<uint16_t>(Next());
  }
  int4* d_seeds = ToDevice(seeds.data(), 2 * kN);
  int4* d_seeds0 = ToDevice(seeds0.data(), kN);
  int4* d_seeds1 = ToDevice(seeds1.data(), kN);
  int4* d_betas = ToDevice(betas.data(), kN);
  uint16_t* d_alphas = ToDevice(alphas.data(), kN);
  uint16_t* d_xs = ToDevice(xs.data(), kN);
  uint32_t* d_xs32 = ToDevice(xs32.data(), kN);
  typename Dpf::Cw* d_cws = DeviceBuf<typename Dpf::Cw>(size_t(kBits + 1) * kN);
  typename Dpf::Cw* d_cws_e = DeviceBuf<typename Dpf::Cw>(size_t(kBits + 1) * kN);
  int4* d_y0 = DeviceBuf<int4>(kN);
  int4* d_y1 = DeviceBuf<int4>(kN);
  int4* d_ye = DeviceBuf<int4>(kN);
  GenKeysKernel<Dpf, uint16_t><<<kN / 128, 128>>>(dpf, d_cws, d_seeds, d_alphas, d_betas, kN);
  EvalKeysKernel<Dpf, uint16_t><<<kN / 128, 128>>>(dpf, d_y0, false, d_seeds0, d_cws, d_xs, kN);
  EvalKeysKernel<Dpf, uint16_t><<<kN / 128, 128>>>(dpf, d_y1, true, d_seeds1, d_cws, d_xs, kN);
  HIP_OK(hipDeviceSynchronize());
  auto y0 = ToHost(d_y0, kN), y1 = ToHost(d_y1, kN);
  int bad = 0;
  for (int i = 0; i < kN; ++i) {
    int4 want = xs[i] == alphas[i] ? G::From(fss::util::SetLsb(betas[i], false)).Into() : kZero;
    bad += !Same(Recon<G>(y0[i], y1[i]), want);
  }
  EXPECT(bad == 0);
  dpf.GenBatch(d_cws_e, d_seeds, d_alphas, d_betas, kN);
  auto a = ToHost(d_cws, size_t(kBits + 1) * kN), e = ToHost(d_cws_e, size_t(kBits + 1) * kN);
  bad = 0;
  for (size_t i = 0; i < a.size(); ++i) bad += !Same(a[i].s, e[i].s) || (i % (kBits + 1) != kBits && a[i].tr != e[i].tr);
  EXPECT(bad == 0);
  int4* cw_s = DeviceBuf<int4>(size_t(kBits) * kN);
  uint32_t* extra = DeviceBuf<uint32_t>(kN);
  int4* out_cw = DeviceBuf<int4>(kN);
  fss::gpu::DpfRelayoutGpu<kBits, G, Prg, uint16_t>(d_cws, kN, cw_s, extra, out_cw);
  fss::gpu::DpfEvalPointGpu<4>(true, d_seeds1, cw_s, extra, out_cw, d_xs32, d_ye, kN, dpf);
  HIP_OK(hipDeviceSynchronize());
  auto ye = ToHost(d_ye, kN);
  bad = 0;
  for (int i = 0; i < kN; ++i) bad += !Same(ye[i], y1[i]);
  EXPECT(bad == 0);
  for (void* p : {(void*)d_seeds, (void*)d_seeds0, (void*)d_seeds1, (void*)d_betas, (void*)d_alphas, (void*)d_xs, (void*)d_xs32,
                  (void*)d_cws, (void*)d_cws_e, (void*)d_y0, (void*)d_y1, (void*)d_ye, (void*)cw_s, (void*)extra, (void*)out_cw})
    HIP_OK(hipFree(p));
  std::printf("AES-128-MMO  DPF n=16  %-22s 1024 keys per thread == engine, reconstructs: %s\n", name,
              g_fail == before ? "ok" : "FAILED");
}
static void Groups() {
  GroupCase<fss::group::Bytes>("Bytes");
  GroupCase<fss::group::Uint<uint8_t>>("Uint<u8>");
  GroupCase<fss::group::Uint<uint32_t>>("Uint<u32>");
  GroupCase<fss::group::Uint<unsigned __int128, static_cast<unsigned __int128>(1) << 127>>("Uint<u128, 2^127>");
  GroupCase<fss::group::Uint<uint32_t, 4294967291u>>("Uint<u32, 2^32 - 5>");
  GroupCase<fss::group::Uint<uint64_t, 18446744073709551557ull>>("Uint<u64, 2^64 - 59>");
}


// Milliseconds per launch at steady clocks: the chip needs tens of milliseconds of continuous work to reach the clock it
// sustains (a 1 ms kernel timed cold runs 10 - 15 % below it), so `launch` runs back to back for `warm_ms` first, then the
// median of `reps` event-timed launches is taken.  quick: one warm-up, one timed launch (the correctness-only passes).
template <typename F>
static float SteadyMs(F&& launch, bool quick = false, float warm_ms = 60.f, int reps = 21) {
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  auto once = [&] {
    float ms = 0;
    HIP_OK(hipEventRecord(e0));
    launch();
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
  };
  float est = once();
  float out = est;
  if (!quick) {
    est = std::max(once(), 1e-3f);
    const int warm = std::min(2000, static_cast<int>(warm_ms / est) + 1);
    for (int i = 0; i < warm; ++i) launch();
    std::vector<float> t(static_cast<size_t>(reps));
    for (auto& v : t) v = once();
    std::sort(t.begin(), t.end());
    out = t[t.size() / 2];
  }
  HIP_OK(hipEventDestroy(e0));
  HIP_OK(hipEventDestroy(e1));
  return out;
}

// what the compatibility form costs next to the engine: DPF / DCF n=32, 2^kLog independent keys, one evaluation each
// (2^18: 1024 workgroups, ONE residency of the chip -- start-up and the ragged end are in the figure; 2^21: eight)
template <int kLog>
static void Timing() {
  constexpr int kN = 1 << kLog;
  using G = fss::group::Uint<uint64_t>;
  using Prg = fss::prg::Aes128Mmo<2>;
  using Dpf = fss::Dpf<32, G, Prg, uint32_t>;
  const unsigned char* keys[2] = {k0, k1};
  Prg prg(keys);
  Dpf dpf{prg};
  std::vector<int4> seeds(2 * kN), seeds0(kN), betas(kN);
  std::vector<uint32_t> alphas(kN), xs(kN);
  for (int i = 0; i < kN; ++i) {
    seeds0[i] = seeds[2 * i] = RandSeed();
    seeds[2 * i + 1] = RandSeed();
    betas[i] = RandSeed();
    alphas[i] = static_cast<uint32_t>(Next());
    xs[i] = i % 2 ? alphas[i] : static_cast<uint32_t>(Next());
  }
  int4* d_seeds = ToDevice(seeds.data(), 2 * kN);
  int4* d_seeds0 = ToDevice(seeds0.data(), kN);
  int4* d_betas = ToDevice(betas.data(), kN);
  uint32_t* d_alphas = ToDevice(alphas.data(), kN);
  uint32_t* d_xs = ToDevice(xs.data(), kN);
  Dpf::Cw* d_cws = DeviceBuf<Dpf::Cw>(size_t(33) * kN);
  int4* d_yt = DeviceBuf<int4>(kN);
  int4* d_ye = DeviceBuf<int4>(kN);
  int4* cw_s = DeviceBuf<int4>(size_t(32) * kN);
  uint32_t* extra = DeviceBuf<uint32_t>(kN);
  int4* out_cw = DeviceBuf<int4>(kN);
  dpf.GenBatch(d_cws, d_seeds, d_alphas, d_betas, kN);
  fss::gpu::DpfRelayoutGpu<32, G, Prg, uint32_t>(d_cws, kN, cw_s, extra, out_cw);
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  // 2^18 keys: one cold-ish launch each (one residency of the chip: what a single small call costs); 2^21: steady clocks
  constexpr bool kQuick = kLog <= 18;
  const float ms_thread = SteadyMs([&] { EvalKeysKernel<Dpf, uint32_t><<<kN / 256, 256>>>(dpf, d_yt, false, d_seeds0, d_cws, d_xs, kN); }, kQuick);
  const float ms_engine = SteadyMs([&] { fss::gpu::DpfEvalPointGpu<4>(false, d_seeds0, cw_s, extra, out_cw, d_xs, d_ye, kN, dpf); }, kQuick);
  auto yt = ToHost(d_yt, kN), ye = ToHost(d_ye, kN);
  int bad = 0;
  for (int i = 0; i < kN; ++i) bad += !Same(yt[i], ye[i]);
  EXPECT(bad == 0);
  std::printf("     (DPF n=32, 2^%d keys: per-thread Eval in a caller kernel %.3f ms = %.1f M evals/s; "
              "DpfEvalPointGpu %.3f ms = %.1f M evals/s; same shares: %s)\n",
              kLog, ms_thread, kN / ms_thread / 1e3, ms_engine, kN / ms_engine / 1e3, bad ? "NO" : "yes");
  // the DPF scheme through a pointer shared by the waves of a workgroup (per-wave table copies found missing and filled)
  HIP_OK(hipMemset(d_yt, 0xff, sizeof(int4) * kN));
  EvalThroughSharedPointerKernel<Dpf, uint32_t><<<kN / 256, 256>>>(dpf, d_yt, false, d_seeds0, d_cws, d_xs, kN);
  HIP_OK(hipDeviceSynchronize());
  yt = ToHost(d_yt, kN);
  bad = 0;
  for (int i = 0; i < kN; ++i) bad += !Same(yt[i], ye[i]);
  EXPECT(bad == 0);
  {  // DCF n=32 (four AES keys: the round keys live in the wave's LDS slot, no scratch), same batch size
    using PrgD = fss::prg::Aes128Mmo<4>;
    using Dcf = fss::Dcf<32, G, PrgD, uint32_t>;
    const unsigned char* keys4[4] = {k0, k1, k2, k3};
    Dcf dcf{PrgD(keys4)};
    Dcf::Cw* d_dc = DeviceBuf<Dcf::Cw>(size_t(33) * kN);
    int4* cw_v = DeviceBuf<int4>(size_t(32) * kN);
    dcf.GenBatch(d_dc, d_seeds, d_alphas, d_betas, kN);
    fss::gpu::DcfRelayoutGpu<32, G, PrgD, uint32_t>(d_dc, kN, cw_s, cw_v, out_cw);
    const float ms_t = SteadyMs([&] { EvalKeysKernel<Dcf, uint32_t><<<kN / 256, 256>>>(dcf, d_yt, false, d_seeds0, d_dc, d_xs, kN); }, kQuick);
    const float ms_e = SteadyMs([&] { fss::gpu::DcfEvalPointGpu<4>(false, d_seeds0, cw_s, cw_v, out_cw, d_xs, d_ye, kN, dcf); }, kQuick);
    auto dt = ToHost(d_yt, kN), de = ToHost(d_ye, kN);
    int badd = 0;
    for (int i = 0; i < kN; ++i) badd += !Same(dt[i], de[i]);
    EXPECT(badd == 0);
    std::printf("     (DCF n=32, 2^%d keys: per-thread Eval in a caller kernel %.3f ms = %.1f M evals/s; "
                "DcfEvalPointGpu %.3f ms = %.1f M evals/s; same shares: %s)\n",
                kLog, ms_t, kN / ms_t / 1e3, ms_e, kN / ms_e / 1e3, badd ? "NO" : "yes");
    HIP_OK(hipFree(d_dc)); HIP_OK(hipFree(cw_v));
  }
  if constexpr (kLog <= 18) {  // Half-Tree n=64 (one AES key: the schedule in scalar registers), 2^18 keys
    using PrgH = fss::prg::Aes128Mmo<1>;
    using Ht = fss::HalfTreeDpf<64, G, PrgH, uint64_t>;
    const unsigned char* keys1[1] = {k0};
    Ht ht{PrgH(keys1), int4{0x12345678, static_cast<int>(0x9abcdef0u), 0x13572468, 0x2468ace0}};
    std::vector<uint64_t> ha(kN), hx(kN);
    for (int i = 0; i < kN; ++i) {
      ha[i] = Next();
      hx[i] = i % 2 ? ha[i] : Next();
    }
    uint64_t* d_ha = ToDevice(ha.data(), kN);
    uint64_t* d_hx = ToDevice(hx.data(), kN);
    typename Ht::Cw* d_hc = DeviceBuf<typename Ht::Cw>(size_t(64) * kN);
    int4* d_ho = DeviceBuf<int4>(kN);
    int4* h_s = DeviceBuf<int4>(size_t(64) * kN);
    ht.GenBatch(d_hc, d_ho, d_seeds, d_ha, d_betas, kN);
    fss::gpu::HalfTreeDpfRelayoutGpu<64, G, PrgH, uint64_t>(d_hc, kN, h_s, extra);
    const float ms_t = SteadyMs([&] { HtEvalKeysKernel<Ht, uint64_t><<<kN / 256, 256>>>(ht, d_yt, false, d_seeds0, d_hc, d_ho, d_hx, kN); }, kQuick);
    const float ms_e = SteadyMs([&] { fss::gpu::HalfTreeDpfEvalPointGpu<4>(false, d_seeds0, h_s, extra, d_ho, d_hx, d_ye, kN, ht); }, kQuick);
    auto ht_t = ToHost(d_yt, kN), ht_e = ToHost(d_ye, kN);
    int badh = 0;
    for (int i = 0; i < kN; ++i) badh += !Same(ht_t[i], ht_e[i]);
    EXPECT(badh == 0);
    std::printf("     (Half-Tree n=64, 2^%d keys: per-thread Eval in a caller kernel %.3f ms = %.1f M evals/s; "
                "HalfTreeDpfEvalPointGpu %.3f ms = %.1f M evals/s; same shares: %s)\n",
                kLog, ms_t, kN / ms_t / 1e3, ms_e, kN / ms_e / 1e3, badh ? "NO" : "yes");
    for (void* q : {(void*)d_ha, (void*)d_hx, (void*)d_hc, (void*)d_ho, (void*)h_s}) HIP_OK(hipFree(q));
  }
  for (void* q : {(void*)d_seeds, (void*)d_seeds0, (void*)d_betas, (void*)d_alphas, (void*)d_xs, (void*)d_cws, (void*)d_yt, (void*)d_ye,
                  (void*)cw_s, (void*)extra, (void*)out_cw})
    HIP_OK(hipFree(q));
}
// ---- fss::prg::Aes128MmoWide: the opt-in wide form (the engine's four-table layout, 128 KiB of LDS) in the caller's own
// kernels -- workgroups of 1024, InitWorkgroup() by every thread first, the scheme used in place as the kernel's argument.
// Shape of src/bench_gpu.cu:100-138: the block prepares the tables, then one Eval per thread.
template <typename S, typename In>
__global__ __launch_bounds__(1024) void WideEvalKeysKernel(S sch, int4* ys, bool party, const int4* seeds, const typename S::Cw* cws,
                                                           const In* xs, int n) {
  decltype(sch.prg)::InitWorkgroup();  // every thread, before any of them leaves
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  ys[tid] = sch.Eval(party, seeds[tid], cws + size_t(tid) * S::kSlots, xs[tid]);
}
// ... and the same with a grid-stride loop: as many workgroups as the chip has CUs, every thread walks key after key.  The
// tables are filled once per workgroup and a CU is never left waiting for the last wave of one workgroup before the next
// may start (with 128 KiB of LDS per workgroup, workgroups cannot overlap on a CU).
template <typename S, typename In>
__global__ __launch_bounds__(1024) void WideEvalKeysStrideKernel(S sch, int4* ys, bool party, const int4* seeds, const typename S::Cw* cws,
                                                                 const In* xs, int n) {
  decltype(sch.prg)::InitWorkgroup();
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
    ys[k] = sch.Eval(party, seeds[k], cws + size_t(k) * S::kSlots, xs[k]);
}
template <typename S, typename In>
__global__ __launch_bounds__(1024) void WideHtEvalKeysKernel(S sch, int4* ys, bool party, const int4* seeds, const typename S::Cw* cws,
                                                             const int4* ocws, const In* xs, int n) {
  decltype(sch.prg)::InitWorkgroup();
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  ys[tid] = sch.Eval(party, seeds[tid], cws + size_t(tid) * S::kSlots, ocws[tid], xs[tid]);
}
template <typename S, typename In>
__global__ __launch_bounds__(1024) void WideGenKeysKernel(S sch, typename S::Cw* cws, const int4* seeds, const In* alphas, const int4* betas,
                                                          int n) {
  decltype(sch.prg)::InitWorkgroup();
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  int4 s[2] = {seeds[tid * 2], seeds[tid * 2 + 1]};
  sch.Gen(cws + size_t(tid) * S::kSlots, s, alphas[tid], betas[tid]);
}
template <int mul>
__global__ __launch_bounds__(1024) void WidePrgKernel(fss::prg::Aes128MmoWide<mul> prg, const int4* seeds, int4* out, int n) {
  fss::prg::Aes128MmoWide<mul>::InitWorkgroup();
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  const auto g = prg.Gen(seeds[tid]);
  for (int i = 0; i < mul; ++i) out[size_t(tid) * mul + i] = g[i];
}

// ... and fss::prg::Aes128SoftWide: the PRG constructed PER THREAD inside the kernel from key bytes in constant memory (the
// reference's AES kernel shape, src/bench_gpu.cu:100-138), on the same wide tables.
__constant__ uint8_t kWideKeys[4][16] = {{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16},
                                         {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1},
                                         {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8},
                                         {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1}};
template <typename S, typename Prg, typename In>
__global__ __launch_bounds__(1024) void SoftWideEvalKeysKernel(int4* ys, bool party, const int4* seeds, const typename S::Cw* cws, const In* xs,
                                                               int n) {
  Prg::InitWorkgroup();
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  Prg prg(kWideKeys);
  S sch{prg};
  ys[tid] = sch.Eval(party, seeds[tid], cws + size_t(tid) * S::kSlots, xs[tid]);
}
template <typename S, typename Prg, typename In>
__global__ __launch_bounds__(1024) void SoftWideGenKeysKernel(typename S::Cw* cws, const int4* seeds, const In* alphas, const int4* betas, int n) {
  Prg::InitWorkgroup();
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  Prg prg(kWideKeys);
  S sch{prg};
  int4 s[2] = {seeds[tid * 2], seeds[tid * 2 + 1]};
  sch.Gen(cws + size_t(tid) * S::kSlots, s, alphas[tid], betas[tid]);
}

template <int kLog>
static void Wide(bool timing) {
  const int before = g_fail;
  constexpr int kN = 1 << kLog;
  using G = fss::group::Uint<uint64_t>;
  const unsigned char* keys1[1] = {k0};
  const unsigned char* keys2[2] = {k0, k1};
  const unsigned char* keys4[4] = {k0, k1, k2, k3};
  std::vector<int4> seeds(2 * kN), seeds0(kN), betas(kN);
  std::vector<uint32_t> alphas(kN), xs(kN);
  for (int i = 0; i < kN; ++i) {
    seeds0[i] = seeds[2 * i] = RandSeed();
    seeds[2 * i + 1] = RandSeed();
    betas[i] = RandSeed();
    alphas[i] = static_cast<uint32_t>(Next());
    xs[i] = i % 2 ? alphas[i] : static_cast<uint32_t>(Next());
  }
  int4* d_seeds = ToDevice(seeds.data(), 2 * kN);
  int4* d_seeds0 = ToDevice(seeds0.data(), kN);
  int4* d_betas = ToDevice(betas.data(), kN);
  uint32_t* d_alphas = ToDevice(alphas.data(), kN);
  uint32_t* d_xs = ToDevice(xs.data(), kN);
  int4* d_yt = DeviceBuf<int4>(kN);
  int4* d_ye = DeviceBuf<int4>(kN);
  int4* cw_s = DeviceBuf<int4>(size_t(32) * kN);
  int4* cw_v = DeviceBuf<int4>(size_t(32) * kN);
  uint32_t* extra = DeviceBuf<uint32_t>(kN);
  int4* out_cw = DeviceBuf<int4>(kN);
  auto timed = [&](auto&& launch) { return SteadyMs(launch, !timing); };
  int num_cu = 256;
  {
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    num_cu = prop.multiProcessorCount;
  }
  {  // the PRG itself: every output of Gen for one, two and four keys against the engine's fss_amd_prg_gen
    int4* d_out = DeviceBuf<int4>(size_t(4) * 4096);
    auto check = [&](auto prg, auto engine_prg, int mul) {
      WidePrgKernel<<<4, 1024>>>(prg, d_seeds0, d_out, 4096);
      HIP_OK(hipDeviceSynchronize());
      auto got = ToHost(d_out, size_t(mul) * 4096);
      int bad = 0;
      for (int i = 0; i < 64; ++i) {
        const auto want = engine_prg.Gen(seeds0[i * 61 % 4096]);  // host Gen: a batch of one on the engine
        for (int j = 0; j < mul; ++j) bad += !Same(got[size_t(i * 61 % 4096) * mul + j], want[j]);
      }
      EXPECT(bad == 0);
    };
    check(fss::prg::Aes128MmoWide<1>(keys1), fss::prg::Aes128Mmo<1>(keys1), 1);
    check(fss::prg::Aes128MmoWide<2>(keys2), fss::prg::Aes128Mmo<2>(keys2), 2);
    check(fss::prg::Aes128MmoWide<4>(keys4), fss::prg::Aes128Mmo<4>(keys4), 4);
    HIP_OK(hipFree(d_out));
  }
  {  // DPF n = 32: Gen per thread == the engine's Gen; Eval per thread == DpfEvalPointGpu
    using Prg = fss::prg::Aes128MmoWide<2>;
    using Dpf = fss::Dpf<32, G, Prg, uint32_t>;
    Dpf dpf{Prg(keys2)};
    Dpf::Cw* d_cws = DeviceBuf<Dpf::Cw>(size_t(33) * kN);
    Dpf::Cw* d_cws_e = DeviceBuf<Dpf::Cw>(size_t(33) * kN);
    WideGenKeysKernel<Dpf, uint32_t><<<kN / 1024, 1024>>>(dpf, d_cws, d_seeds, d_alphas, d_betas, kN);
    dpf.GenBatch(d_cws_e, d_seeds, d_alphas, d_betas, kN);
    HIP_OK(hipDeviceSynchronize());
    {
      auto a = ToHost(d_cws, size_t(33) * kN), e = ToHost(d_cws_e, size_t(33) * kN);
      int bad = 0;
      for (size_t i = 0; i < a.size(); ++i) bad += !Same(a[i].s, e[i].s) || (i % 33 != 32 && a[i].tr != e[i].tr);
      EXPECT(bad == 0);
    }
    fss::gpu::DpfRelayoutGpu<32, G, Prg, uint32_t>(d_cws, kN, cw_s, extra, out_cw);
    const float ms_t = timed([&] { WideEvalKeysKernel<Dpf, uint32_t><<<kN / 1024, 1024>>>(dpf, d_yt, false, d_seeds0, d_cws, d_xs, kN); });
    const float ms_e = timed([&] { fss::gpu::DpfEvalPointGpu<4>(false, d_seeds0, cw_s, extra, out_cw, d_xs, d_ye, kN, dpf); });
    auto yt = ToHost(d_yt, kN), ye = ToHost(d_ye, kN);
    int bad = 0;
    for (int i = 0; i < kN; ++i) bad += !Same(yt[i], ye[i]);
    EXPECT(bad == 0);
    HIP_OK(hipMemset(d_yt, 0xff, sizeof(int4) * kN));
    const float ms_s = timed([&] { WideEvalKeysStrideKernel<Dpf, uint32_t><<<std::min(kN / 1024, num_cu), 1024>>>(dpf, d_yt, false, d_seeds0, d_cws, d_xs, kN); });
    yt = ToHost(d_yt, kN);
    for (int i = 0; i < kN; ++i) bad += !Same(yt[i], ye[i]);
    EXPECT(bad == 0);
    std::printf("     (WIDE DPF n=32, 2^%d keys, grid-stride loop, %d workgroups of 1024: %.3f ms = %.1f M evals/s)\n", kLog, std::min(kN / 1024, num_cu), ms_s,
                kN / ms_s / 1e3);
    {  // Aes128SoftWide<2>: per-thread construction; Gen and Eval against the same engine results (same keys k0, k1)
      using PrgS = fss::prg::Aes128SoftWide<2>;
      using DpfS = fss::Dpf<32, G, PrgS, uint32_t>;
      DpfS::Cw* d_cws_s = DeviceBuf<DpfS::Cw>(size_t(33) * kN);
      SoftWideGenKeysKernel<DpfS, PrgS, uint32_t><<<kN / 1024, 1024>>>(d_cws_s, d_seeds, d_alphas, d_betas, kN);
      HIP_OK(hipDeviceSynchronize());
      {
        auto a = ToHost(d_cws_s, size_t(33) * kN);
        auto e = ToHost(reinterpret_cast<const DpfS::Cw*>(d_cws), size_t(33) * kN);
        int badg = 0;
        for (size_t i = 0; i < a.size(); ++i) badg += !Same(a[i].s, e[i].s) || (i % 33 != 32 && a[i].tr != e[i].tr);
        EXPECT(badg == 0);
      }
      HIP_OK(hipMemset(d_yt, 0xff, sizeof(int4) * kN));
      const float ms_w = timed([&] { SoftWideEvalKeysKernel<DpfS, PrgS, uint32_t><<<kN / 1024, 1024>>>(d_yt, false, d_seeds0, d_cws_s, d_xs, kN); });
      yt = ToHost(d_yt, kN);
      int bads = 0;
      for (int i = 0; i < kN; ++i) bads += !Same(yt[i], ye[i]);
      EXPECT(bads == 0);
      std::printf("     (WIDE DPF n=32, 2^%d keys, Aes128SoftWide<2> constructed per thread: %.3f ms = %.1f M evals/s; same shares: %s)\n", kLog, ms_w,
                  kN / ms_w / 1e3, bads ? "NO" : "yes");
      HIP_OK(hipFree(d_cws_s));
    }
    std::printf("     (WIDE DPF n=32, 2^%d keys, Aes128MmoWide<2>, 1024 threads per block: per-thread Eval in a caller kernel %.3f ms = %.1f M evals/s; "
                "DpfEvalPointGpu %.3f ms = %.1f M evals/s; same shares: %s)\n", kLog, ms_t, kN / ms_t / 1e3, ms_e, kN / ms_e / 1e3, bad ? "NO" : "yes");
    HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_cws_e));
  }
  {  // DCF n = 32
    using Prg = fss::prg::Aes128MmoWide<4>;
    using Dcf = fss::Dcf<32, G, Prg, uint32_t>;
    Dcf dcf{Prg(keys4)};
    Dcf::Cw* d_dc = DeviceBuf<Dcf::Cw>(size_t(33) * kN);
    Dcf::Cw* d_dc_e = DeviceBuf<Dcf::Cw>(size_t(33) * kN);
    WideGenKeysKernel<Dcf, uint32_t><<<kN / 1024, 1024>>>(dcf, d_dc, d_seeds, d_alphas, d_betas, kN);
    dcf.GenBatch(d_dc_e, d_seeds, d_alphas, d_betas, kN);
    HIP_OK(hipDeviceSynchronize());
    {
      auto a = ToHost(d_dc, size_t(33) * kN), e = ToHost(d_dc_e, size_t(33) * kN);
      EXPECT(std::memcmp(a.data(), e.data(), a.size() * sizeof(a[0])) == 0);
    }
    fss::gpu::DcfRelayoutGpu<32, G, Prg, uint32_t>(d_dc, kN, cw_s, cw_v, out_cw);
    const float ms_t = timed([&] { WideEvalKeysKernel<Dcf, uint32_t><<<kN / 1024, 1024>>>(dcf, d_yt, false, d_seeds0, d_dc, d_xs, kN); });
    const float ms_e = timed([&] { fss::gpu::DcfEvalPointGpu<4>(false, d_seeds0, cw_s, cw_v, out_cw, d_xs, d_ye, kN, dcf); });
    auto yt = ToHost(d_yt, kN), ye = ToHost(d_ye, kN);
    int bad = 0;
    for (int i = 0; i < kN; ++i) bad += !Same(yt[i], ye[i]);
    EXPECT(bad == 0);
    HIP_OK(hipMemset(d_yt, 0xff, sizeof(int4) * kN));
    const float ms_s = timed([&] { WideEvalKeysStrideKernel<Dcf, uint32_t><<<std::min(kN / 1024, num_cu), 1024>>>(dcf, d_yt, false, d_seeds0, d_dc, d_xs, kN); });
    yt = ToHost(d_yt, kN);
    for (int i = 0; i < kN; ++i) bad += !Same(yt[i], ye[i]);
    EXPECT(bad == 0);
    std::printf("     (WIDE DCF n=32, 2^%d keys, grid-stride loop, %d workgroups of 1024: %.3f ms = %.1f M evals/s)\n", kLog, std::min(kN / 1024, num_cu), ms_s,
                kN / ms_s / 1e3);
    {  // Aes128SoftWide<4>
      using PrgS = fss::prg::Aes128SoftWide<4>;
      using DcfS = fss::Dcf<32, G, PrgS, uint32_t>;
      HIP_OK(hipMemset(d_yt, 0xff, sizeof(int4) * kN));
      const float ms_w = timed([&] {
        SoftWideEvalKeysKernel<DcfS, PrgS, uint32_t><<<kN / 1024, 1024>>>(d_yt, false, d_seeds0, reinterpret_cast<const DcfS::Cw*>(d_dc), d_xs, kN);
      });
      yt = ToHost(d_yt, kN);
      int bads = 0;
      for (int i = 0; i < kN; ++i) bads += !Same(yt[i], ye[i]);
      EXPECT(bads == 0);
      std::printf("     (WIDE DCF n=32, 2^%d keys, Aes128SoftWide<4> constructed per thread: %.3f ms = %.1f M evals/s; same shares: %s)\n", kLog, ms_w,
                  kN / ms_w / 1e3, bads ? "NO" : "yes");
    }
    std::printf("     (WIDE DCF n=32, 2^%d keys, Aes128MmoWide<4>, 1024 threads per block: per-thread Eval in a caller kernel %.3f ms = %.1f M evals/s; "
                "DcfEvalPointGpu %.3f ms = %.1f M evals/s; same shares: %s)\n", kLog, ms_t, kN / ms_t / 1e3, ms_e, kN / ms_e / 1e3, bad ? "NO" : "yes");
    HIP_OK(hipFree(d_dc)); HIP_OK(hipFree(d_dc_e));
  }
  if constexpr (kLog <= 18) {  // Half-Tree n = 64 (one key: a plain schedule)
    using Prg = fss::prg::Aes128MmoWide<1>;
    using Ht = fss::HalfTreeDpf<64, G, Prg, uint64_t>;
    Ht ht{Prg(keys1), int4{0x12345678, static_cast<int>(0x9abcdef0u), 0x13572468, 0x2468ace0}};
    std::vector<uint64_t> ha(kN), hx(kN);
    for (int i = 0; i < kN; ++i) {
      ha[i] = Next();
      hx[i] = i % 2 ? ha[i] : Next();
    }
    uint64_t* d_ha = ToDevice(ha.data(), kN);
    uint64_t* d_hx = ToDevice(hx.data(), kN);
    typename Ht::Cw* d_hc = DeviceBuf<typename Ht::Cw>(size_t(64) * kN);
    int4* d_ho = DeviceBuf<int4>(kN);
    int4* h_s = DeviceBuf<int4>(size_t(64) * kN);
    ht.GenBatch(d_hc, d_ho, d_seeds, d_ha, d_betas, kN);
    fss::gpu::HalfTreeDpfRelayoutGpu<64, G, Prg, uint64_t>(d_hc, kN, h_s, extra);
    const float ms_t = timed([&] { WideHtEvalKeysKernel<Ht, uint64_t><<<kN / 1024, 1024>>>(ht, d_yt, false, d_seeds0, d_hc, d_ho, d_hx, kN); });
    const float ms_e = timed([&] { fss::gpu::HalfTreeDpfEvalPointGpu<4>(false, d_seeds0, h_s, extra, d_ho, d_hx, d_ye, kN, ht); });
    auto yt = ToHost(d_yt, kN), ye = ToHost(d_ye, kN);
    int bad = 0;
    for (int i = 0; i < kN; ++i) bad += !Same(yt[i], ye[i]);
    EXPECT(bad == 0);
    std::printf("     (WIDE Half-Tree n=64, 2^%d keys, Aes128MmoWide<1>: per-thread Eval %.3f ms = %.1f M evals/s; HalfTreeDpfEvalPointGpu %.3f ms = "
                "%.1f M evals/s; same shares: %s)\n", kLog, ms_t, kN / ms_t / 1e3, ms_e, kN / ms_e / 1e3, bad ? "NO" : "yes");
    for (void* q : {(void*)d_ha, (void*)d_hx, (void*)d_hc, (void*)d_ho, (void*)h_s}) HIP_OK(hipFree(q));
  }
  for (void* q : {(void*)d_seeds, (void*)d_seeds0, (void*)d_betas, (void*)d_alphas, (void*)d_xs, (void*)d_yt, (void*)d_ye, (void*)cw_s, (void*)cw_v,
                  (void*)extra, (void*)out_cw})
    HIP_OK(hipFree(q));
  std::printf("AES-128-MMO  wide per-thread PRG (Aes128MmoWide<1|2|4>), 2^%d keys: Gen / Eval per thread == engine: %s\n", kLog,
              g_fail == before ? "ok" : "FAILED");
}
}  // namespace p2

// ---------------------------------------------------------------------------------------------- part 3: caller's own types
namespace p3 {
// a toy ARX generator -- NOT a cryptographic PRG; it only has to be deterministic and well mixed for the test
template <int mul>
struct MixPrg {
  uint32_t tweak;
  __host__ __device__ static uint32_t Mix(uint32_t v) {
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
  }
  __host__ __device__ std::array<int4, mul> Gen(int4 seed) const {
    std::array<int4, mul> out;
    for (int i = 0; i < mul; ++i) {
      uint32_t a = Mix(static_cast<uint32_t>(seed.x) + tweak + 0x9e3779b9u * (i + 1));
      uint32_t b = Mix(static_cast<uint32_t>(seed.y) ^ a);
      uint32_t c = Mix(static_cast<uint32_t>(seed.z) + b);
      uint32_t d = Mix(static_cast<uint32_t>(seed.w) ^ c);
      a = Mix(a + d);
      out[i] = int4{static_cast<int>(a), static_cast<int>(b ^ a), static_cast<int>(c + b), static_cast<int>(d ^ c)};
    }
    return out;
  }
};
// (Z_2^32)^3 on x, y, z; w (where the control bit lives) is not part of the group
struct Vec3 {
  uint32_t v[3];
  __host__ __device__ Vec3() : v{0, 0, 0} {}
  __host__ __device__ Vec3 operator+(Vec3 o) const {
    Vec3 r;
    for (int i = 0; i < 3; ++i) r.v[i] = v[i] + o.v[i];
    return r;
  }
  __host__ __device__ Vec3 operator-() const {
    Vec3 r;
    for (int i = 0; i < 3; ++i) r.v[i] = 0u - v[i];
    return r;
  }
  __host__ __device__ static Vec3 From(int4 b) {
    Vec3 r;
    r.v[0] = static_cast<uint32_t>(b.x); r.v[1] = static_cast<uint32_t>(b.y); r.v[2] = static_cast<uint32_t>(b.z);
    return r;
  }
  __host__ __device__ int4 Into() const { return int4{static_cast<int>(v[0]), static_cast<int>(v[1]), static_cast<int>(v[2]), 0}; }
};
static_assert(Prgable<MixPrg<2>, 2> && Prgable<MixPrg<4>, 4> && Groupable<Vec3>);

constexpr int kInBits = 10;
using In = uint16_t;
using Dpf = fss::Dpf<kInBits, Vec3, MixPrg<2>, In>;
using Dcf = fss::Dcf<kInBits, Vec3, MixPrg<4>, In>;
using Ht = fss::HalfTreeDpf<kInBits, Vec3, MixPrg<1>, In>;
static_assert(!Dpf::kEngine && !Dcf::kEngine && !Ht::kEngine);

__global__ void UserDpfKernel(Dpf dpf, Dpf::Cw* cws, const int4* s0s, In alpha, int4 beta, int4* ys /* [2][1 << n] */) {
  // thread 0 of block 0 deals the key, then every thread evaluates its own points of the whole domain
  __shared__ int ready;
  if (threadIdx.x == 0) {
    dpf.Gen(cws, s0s, alpha, beta);
    ready = 1;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < (1 << kInBits); x += blockDim.x)
    for (int b = 0; b < 2; ++b) ys[(b << kInBits) + x] = dpf.Eval(b, s0s[b], cws, static_cast<In>(x));
  (void)ready;
}

static void Run() {
  const int before = g_fail;
  const int4 seeds[2] = {RandSeed(), RandSeed()};
  const In alpha = 777;
  const int4 beta = {11, 22, 33, 0};
  const int n = 1 << kInBits;
  {  // inside the caller's kernel
    Dpf dpf{MixPrg<2>{0xabcdef01u}};
    int4* d_s0s = ToDevice(seeds, 2);
    Dpf::Cw* d_cws = DeviceBuf<Dpf::Cw>(kInBits + 1);
    int4* d_ys = DeviceBuf<int4>(2 * n);
    UserDpfKernel<<<1, 256>>>(dpf, d_cws, d_s0s, alpha, beta, d_ys);
    HIP_OK(hipDeviceSynchronize());
    auto ys = ToHost(d_ys, 2 * n);
    int bad = 0;
    for (int x = 0; x < n; ++x) bad += !Same(Recon<Vec3>(ys[x], ys[n + x]), x == alpha ? beta : kZero);
    EXPECT(bad == 0);
    // the host signatures on the same key: generic kernels, same shares
    Dpf::Cw cws[kInBits + 1];
    dpf.Gen(cws, seeds, alpha, beta);
    auto dc = ToHost(d_cws, kInBits + 1);
    bad = 0;
    for (int i = 0; i <= kInBits; ++i) bad += !Same(cws[i].s, dc[i].s) || cws[i].tr != dc[i].tr;
    EXPECT(bad == 0);
    EXPECT(Same(dpf.Eval(false, seeds[0], cws, alpha), ys[alpha]));
    EXPECT(Same(dpf.Eval(true, seeds[1], cws, In(5)), ys[n + 5]));
    std::vector<int4> all(n);
    dpf.EvalAll(true, seeds[1], cws, all.data());
    bad = 0;
    for (int x = 0; x < n; ++x) bad += !Same(all[x], ys[n + x]);
    EXPECT(bad == 0);
  }
  {  // DCF through the host signatures only
    Dcf dcf{MixPrg<4>{0x1234u}};
    Dcf::Cw cws[kInBits + 1];
    dcf.Gen(cws, seeds, alpha, beta);
    std::vector<int4> y0(n), y1(n);
    dcf.EvalAll(false, seeds[0], cws, y0.data());
    dcf.EvalAll(true, seeds[1], cws, y1.data());
    int bad = 0;
    for (int x = 0; x < n; ++x) bad += !Same(Recon<Vec3>(y0[x], y1[x]), x < alpha ? beta : kZero);
    EXPECT(bad == 0);
    EXPECT(Same(dcf.Eval(false, seeds[0], cws, In(3)), y0[3]));
  }
  {  // Half-Tree: batched device forms with the caller's types
    Ht ht{MixPrg<1>{0x55u}, int4{1, 2, 3, 4}};
    constexpr int kKeys = 300;
    std::vector<int4> s(2 * kKeys), s0(kKeys), s1(kKeys), betas(kKeys);
    std::vector<In> alphas(kKeys);
    for (int i = 0; i < kKeys; ++i) {
      s0[i] = s[2 * i] = RandSeed();
      s1[i] = s[2 * i + 1] = RandSeed();
      betas[i] = int4{(int)Next(), (int)Next(), (int)Next(), 0};
      alphas[i] = static_cast<In>(Next() % n);
    }
    int4* d_s = ToDevice(s.data(), 2 * kKeys);
    int4* d_betas = ToDevice(betas.data(), kKeys);
    In* d_alphas = ToDevice(alphas.data(), kKeys);
    Ht::Cw* d_cws = DeviceBuf<Ht::Cw>(size_t(kInBits) * kKeys);
    int4* d_ocws = DeviceBuf<int4>(kKeys);
    ht.GenBatch(d_cws, d_ocws, d_s, d_alphas, d_betas, kKeys);
    int4* d_y0 = DeviceBuf<int4>(n);
    int4* d_y1 = DeviceBuf<int4>(n);
    std::vector<In> xs(n);
    for (int x = 0; x < n; ++x) xs[x] = static_cast<In>(x);
    In* d_xs = ToDevice(xs.data(), n);
    int4* d_s0 = ToDevice(s0.data(), kKeys);
    int4* d_s1 = ToDevice(s1.data(), kKeys);
    int bad = 0;
    for (int k : {0, 1, 150, kKeys - 1}) {
      ht.EvalPoints(false, d_s0 + k, d_cws + size_t(k) * kInBits, d_ocws + k, d_xs, d_y0, n);
      ht.EvalPoints(true, d_s1 + k, d_cws + size_t(k) * kInBits, d_ocws + k, d_xs, d_y1, n);
      HIP_OK(hipDeviceSynchronize());
      auto y0 = ToHost(d_y0, n), y1 = ToHost(d_y1, n);
      for (int x = 0; x < n; ++x) bad += !Same(Recon<Vec3>(y0[x], y1[x]), x == alphas[k] ? betas[k] : kZero);
    }
    EXPECT(bad == 0);
  }
  {  // the fss::gpu launchers with the caller's types (the reference's are templates over any PRG passed by value,
     // point_eval_gpu.cuh:448-460, eval_all_gpu.cuh:478-535): level-major arrays from *RelayoutGpu, one key per thread
    constexpr int kKeys = 777;
    std::vector<int4> sd(2 * kKeys), s0(kKeys), s1(kKeys), betas(kKeys);
    std::vector<In> alphas(kKeys);
    std::vector<uint32_t> xs(kKeys);
    for (int i = 0; i < kKeys; ++i) {
      s0[i] = sd[2 * i] = RandSeed();
      s1[i] = sd[2 * i + 1] = RandSeed();
      betas[i] = int4{(int)Next(), (int)Next(), (int)Next(), 0};
      alphas[i] = static_cast<In>(Next() % n);
      xs[i] = i % 2 ? alphas[i] : static_cast<uint32_t>(Next() % n);
    }
    int4* d_sd = ToDevice(sd.data(), 2 * kKeys);
    int4* d_s0 = ToDevice(s0.data(), kKeys);
    int4* d_s1 = ToDevice(s1.data(), kKeys);
    int4* d_betas = ToDevice(betas.data(), kKeys);
    In* d_alphas = ToDevice(alphas.data(), kKeys);
    uint32_t* d_xs = ToDevice(xs.data(), kKeys);
    int4* d_y0 = DeviceBuf<int4>(kKeys);
    int4* d_y1 = DeviceBuf<int4>(kKeys);
    int4* cw_s = DeviceBuf<int4>(size_t(kInBits) * kKeys);
    int4* cw_v = DeviceBuf<int4>(size_t(kInBits) * kKeys);
    uint32_t* extra = DeviceBuf<uint32_t>(kKeys);
    int4* out_cw = DeviceBuf<int4>(kKeys);
    int bad = 0;
    {  // DPF: x == alpha on odd keys
      Dpf dpf{MixPrg<2>{0xabcdef01u}};
      Dpf::Cw* d_cws = DeviceBuf<Dpf::Cw>(size_t(kInBits + 1) * kKeys);
      dpf.GenBatch(d_cws, d_sd, d_alphas, d_betas, kKeys);
      fss::gpu::DpfRelayoutGpu<kInBits, Vec3, MixPrg<2>, In>(d_cws, kKeys, cw_s, extra, out_cw);
      fss::gpu::DpfEvalPointGpu<4>(false, d_s0, cw_s, extra, out_cw, d_xs, d_y0, kKeys, dpf);
      fss::gpu::DpfEvalPointGpu<4>(true, d_s1, cw_s, extra, out_cw, d_xs, d_y1, kKeys, dpf);
      HIP_OK(hipDeviceSynchronize());
      auto y0 = ToHost(d_y0, kKeys), y1 = ToHost(d_y1, kKeys);
      for (int i = 0; i < kKeys; ++i) bad += !Same(Recon<Vec3>(y0[i], y1[i]), xs[i] == alphas[i] ? betas[i] : kZero);
      // the whole domain of a few keys, singly and as a batch, against the key-major per-point form
      int4* d_all = DeviceBuf<int4>(size_t(3) * n);
      auto hc = ToHost(d_cws, size_t(kInBits + 1) * kKeys);
      fss::gpu::DpfEvalAllGpu<20, 12, 256>(true, s1[5], d_cws + size_t(5) * (kInBits + 1), d_all, dpf);
      HIP_OK(hipDeviceSynchronize());
      auto one = ToHost(d_all, n);
      for (int x = 0; x < n; x += 37) bad += !Same(one[x], dpf.Eval(true, s1[5], &hc[size_t(5) * (kInBits + 1)], In(x)));
      fss::gpu::DpfEvalAllGpuBatch<17, 9, 256>(false, d_s0 + 4, d_cws + size_t(4) * (kInBits + 1), 3, d_all, dpf);
      HIP_OK(hipDeviceSynchronize());
      auto three = ToHost(d_all, size_t(3) * n);
      for (int k = 0; k < 3; ++k)
        for (int x = 0; x < n; x += 101)
          bad += !Same(three[size_t(k) * n + x], dpf.Eval(false, s0[4 + k], &hc[size_t(4 + k) * (kInBits + 1)], In(x)));
    }
    {  // DCF (x < alpha releases beta)
      Dcf dcf{MixPrg<4>{0x1234u}};
      Dcf::Cw* d_cws = DeviceBuf<Dcf::Cw>(size_t(kInBits + 1) * kKeys);
      dcf.GenBatch(d_cws, d_sd, d_alphas, d_betas, kKeys);
      fss::gpu::DcfRelayoutGpu<kInBits, Vec3, MixPrg<4>, In>(d_cws, kKeys, cw_s, cw_v, out_cw);
      fss::gpu::DcfEvalPointGpu<4>(false, d_s0, cw_s, cw_v, out_cw, d_xs, d_y0, kKeys, dcf);
      fss::gpu::DcfEvalPointGpu<4>(true, d_s1, cw_s, cw_v, out_cw, d_xs, d_y1, kKeys, dcf);
      HIP_OK(hipDeviceSynchronize());
      auto y0 = ToHost(d_y0, kKeys), y1 = ToHost(d_y1, kKeys);
      for (int i = 0; i < kKeys; ++i) bad += !Same(Recon<Vec3>(y0[i], y1[i]), xs[i] < alphas[i] ? betas[i] : kZero);
    }
    {  // Half-Tree
      Ht ht{MixPrg<1>{0x55u}, int4{1, 2, 3, 4}};
      Ht::Cw* d_cws = DeviceBuf<Ht::Cw>(size_t(kInBits) * kKeys);
      int4* d_ocws = DeviceBuf<int4>(kKeys);
      ht.GenBatch(d_cws, d_ocws, d_sd, d_alphas, d_betas, kKeys);
      fss::gpu::HalfTreeDpfRelayoutGpu<kInBits, Vec3, MixPrg<1>, In>(d_cws, kKeys, cw_s, extra);
      fss::gpu::HalfTreeDpfEvalPointGpu<4>(false, d_s0, cw_s, extra, d_ocws, d_xs, d_y0, kKeys, ht);
      fss::gpu::HalfTreeDpfEvalPointGpu<4>(true, d_s1, cw_s, extra, d_ocws, d_xs, d_y1, kKeys, ht);
      HIP_OK(hipDeviceSynchronize());
      auto y0 = ToHost(d_y0, kKeys), y1 = ToHost(d_y1, kKeys);
      for (int i = 0; i < kKeys; ++i) bad += !Same(Recon<Vec3>(y0[i], y1[i]), xs[i] == alphas[i] ? betas[i] : kZero);
      int4* d_all = DeviceBuf<int4>(size_t(2) * n);
      auto ho = ToHost(d_ocws, kKeys);
      fss::gpu::HalfTreeDpfEvalAllGpu<20, 12, 256>(false, s0[9], d_cws + size_t(9) * kInBits, ho[9], d_all, ht);
      fss::gpu::HalfTreeDpfEvalAllGpu<20, 12, 256>(true, s1[9], d_cws + size_t(9) * kInBits, ho[9], d_all + n, ht);
      HIP_OK(hipDeviceSynchronize());
      auto two = ToHost(d_all, size_t(2) * n);
      for (int x = 0; x < n; ++x) bad += !Same(Recon<Vec3>(two[x], two[n + x]), x == alphas[9] ? betas[9] : kZero);
      fss::gpu::HalfTreeDpfEvalAllGpuBatch<17, 9, 256>(false, d_s0 + 9, d_cws + size_t(9) * kInBits, d_ocws + 9, 2, d_all, ht);
      HIP_OK(hipDeviceSynchronize());
      auto batch = ToHost(d_all, size_t(2) * n);
      for (int x = 0; x < n; ++x) bad += !Same(batch[x], two[x]);
    }
    EXPECT(bad == 0);
  }
  std::printf("caller-defined Prgable + Groupable  in a kernel, through the host signatures and the fss::gpu launchers, reconstructs: %s\n",
              g_fail == before ? "ok" : "FAILED");
}
}  // namespace p3

// ---------------------------------------------------------------------------------------------- part 4: Aes128Soft
// fss::prg::Aes128Soft / fss::prg::aes_detail in the shape the reference's GPU benchmark uses them
// (src/bench_gpu.cu:100-138: tables in shared memory filled from ComputeTe0 / Sbox, the PRG constructed per thread from
// keys in constant memory, one key per thread, key-major correction words, 256 threads per block).
namespace p4 {
namespace ad = fss::prg::aes_detail;
constexpr int kBits = 20;
constexpr int kThreads = 256;
using G = fss::group::Uint<uint64_t>;
using Soft2 = fss::prg::Aes128Soft<2>;
using Dpf = fss::Dpf<kBits, G, Soft2, uint32_t>;
using Mmo2 = fss::prg::Aes128Mmo<2>;
using DpfE = fss::Dpf<kBits, G, Mmo2, uint32_t>;  // the engine's view of the same function

__constant__ uint8_t kSoftKeys[4][16] = {{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16},
                                         {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1},
                                         {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8},
                                         {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1}};
// a second key pair, for threads that do NOT share their keys with the rest of their wave
__constant__ uint8_t kOtherKeys[2][16] = {{9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6},
                                          {0xa0, 0xa1, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xab, 0xac, 0xad, 0xae, 0xaf}};

struct SharedTables {
  uint32_t te0[256];
  uint8_t sbox[256];
};
__device__ void FillTables(SharedTables& t) {  // the whole workgroup, then a barrier
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    t.te0[i] = ad::ComputeTe0(static_cast<uint8_t>(i));
    t.sbox[i] = ad::Sbox(static_cast<uint8_t>(i));
  }
  __syncthreads();
}

__global__ void GenKernel(Dpf::Cw* cws, const int4* seeds, const uint32_t* alphas, const int4* betas, int n) {
  __shared__ SharedTables tabs;
  FillTables(tabs);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  Soft2 prg(kSoftKeys, tabs.te0, tabs.sbox);
  Dpf dpf{prg};
  int4 s[2] = {seeds[2 * tid], seeds[2 * tid + 1]};
  dpf.Gen(cws + size_t(tid) * (kBits + 1), s, alphas[tid], betas[tid]);
}
__global__ void EvalKernel(int4* ys, bool party, const int4* seeds, const Dpf::Cw* cws, const uint32_t* xs, int n) {
  __shared__ SharedTables tabs;
  FillTables(tabs);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  Soft2 prg(kSoftKeys, tabs.te0, tabs.sbox);
  Dpf dpf{prg};
  ys[tid] = dpf.Eval(party, seeds[tid], cws + size_t(tid) * (kBits + 1), xs[tid]);
}
// A scheme built on the HOST (its PRG's table pointers are host arrays) and passed to the kernel by value, no tables of the
// kernel's own: the device side of Aes128Soft reads the workgroup's bank-replicated copy, never the caller's pointers.
__global__ void HostBuiltEvalKernel(Dpf dpf, int4* ys, bool party, const int4* seeds, const Dpf::Cw* cws, const uint32_t* xs, int n) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  ys[tid] = dpf.Eval(party, seeds[tid], cws + size_t(tid) * (kBits + 1), xs[tid]);
}
// the same kernel on the bank-replicated table (fss::prg::Aes128SoftBanked): 32 KiB + 256 B of LDS per workgroup
using SoftB2 = fss::prg::Aes128SoftBanked<2>;
using DpfB = fss::Dpf<kBits, G, SoftB2, uint32_t>;
__global__ void BankedEvalKernel(int4* ys, bool party, const int4* seeds, const DpfB::Cw* cws, const uint32_t* xs, int n) {
  __shared__ uint32_t s_te0[ad::kBankedTe0Words];
  __shared__ uint8_t s_sbox[256];
  ad::InitTe0Banked(s_te0, threadIdx.x, blockDim.x);
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s_sbox[i] = ad::Sbox(static_cast<uint8_t>(i));
  __syncthreads();
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  SoftB2 prg(kSoftKeys, s_te0, s_sbox);
  DpfB dpf{prg};
  ys[tid] = dpf.Eval(party, seeds[tid], cws + size_t(tid) * (kBits + 1), xs[tid]);
}
// odd threads use another key pair: the lanes of a wave do not agree on the round keys
__global__ void MixedKeysEvalKernel(int4* ys, const int4* seeds, const Dpf::Cw* cws, const uint32_t* xs, int n) {
  __shared__ SharedTables tabs;
  FillTables(tabs);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  Soft2 prg(tid & 1 ? kOtherKeys : kSoftKeys, tabs.te0, tabs.sbox);
  Dpf dpf{prg};
  ys[tid] = dpf.Eval(false, seeds[tid], cws + size_t(tid) * (kBits + 1), xs[tid]);
}
// raw PRG and the byte-level helpers on the device, tables in LDS and in global memory
__global__ void PrgKernel(int4* out /* [4 + 4 + 1] */, const uint32_t* g_te0, const uint8_t* g_sbox, int4 seed) {
  __shared__ SharedTables tabs;
  FillTables(tabs);
  if (threadIdx.x != 0) return;
  fss::prg::Aes128Soft<4> lds_prg(kSoftKeys, tabs.te0, tabs.sbox);
  auto a = lds_prg.Gen(seed);
  fss::prg::Aes128Soft<4> glob_prg(kSoftKeys, g_te0, g_sbox);  // not LDS: the generic-pointer path
  auto b = glob_prg.Gen(seed);
  for (int i = 0; i < 4; ++i) { out[i] = a[i]; out[4 + i] = b[i]; }
  uint8_t rk[ad::kRoundKeySize], buf[16];
  ad::KeyExpansion(rk, kSoftKeys[0], tabs.sbox);
  __builtin_memcpy(buf, &seed, 16);
  ad::Encrypt(buf, rk, tabs.te0, tabs.sbox);
  int4 c;
  __builtin_memcpy(&c, buf, 16);
  out[8] = int4{c.x ^ seed.x, c.y ^ seed.y, c.z ^ seed.z, c.w ^ seed.w};
}
// DCF over Aes128Soft<4>: a level takes {s, v} of one side (GenSide: two blocks instead of four)
using DcfS = fss::Dcf<16, G, fss::prg::Aes128Soft<4>, uint32_t>;
using DcfE = fss::Dcf<16, G, fss::prg::Aes128Mmo<4>, uint32_t>;
__global__ void DcfEvalKernel(int4* ys, bool party, const int4* seeds, const DcfS::Cw* cws, const uint32_t* xs, int n) {
  __shared__ SharedTables tabs;
  FillTables(tabs);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n) return;
  fss::prg::Aes128Soft<4> prg(kSoftKeys, tabs.te0, tabs.sbox);
  DcfS dcf{prg};
  ys[tid] = dcf.Eval(party, seeds[tid], cws + size_t(tid) * DcfS::kSlots, xs[tid]);
}

static void Run() {
  const int before = g_fail;
  // ---- the helpers' values: the reference's table constants and FIPS-197
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
  {  // FIPS-197 appendix C.1 through KeyExpansion + Encrypt on the host
    uint8_t key[16], pt[16], rk[ad::kRoundKeySize];
    for (int i = 0; i < 16; ++i) { key[i] = static_cast<uint8_t>(i); pt[i] = static_cast<uint8_t>(i * 0x11); }
    ad::KeyExpansion(rk, key, sbox);
    ad::Encrypt(pt, rk, te0, sbox);
    const uint8_t want[16] = {0x69, 0xc4, 0xe0, 0xd8, 0x6a, 0x7b, 0x04, 0x30, 0xd8, 0xcd, 0xb7, 0x80, 0x70, 0xb4, 0xc5, 0x5a};
    EXPECT(std::memcmp(pt, want, 16) == 0);
    EXPECT(rk[160] == 0x13 && rk[161] == 0x11 && rk[162] == 0x1d && rk[163] == 0x7f && rk[174] == 0x30 && rk[175] == 0xc5);  // FIPS-197 C.1: w40 = 13111d7f, w43 = 4d2b30c5
  }
  const uint8_t hkeys[4][16] = {{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16},
                                {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1},
                                {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8},
                                {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1}};
  const int4 kat_seed = I4(0x11111111, 0x22222222, 0x33333333, 0x44444440);
  const int4 kat[4] = {I4(0x1423e6d2, 0x60533602, 0x813b3fbc, 0x412b31dc), I4(0x2d18cbe7, 0xa5eddcc9, 0xc691e4f6, 0x97831704),
                       I4(0x2c046c24, 0x8a033811, 0xcb0335b6, 0x3d799ddb), I4(0x2413b69f, 0xffed7ac4, 0x314804c9, 0x595d2580)};
  {  // host Gen / GenSide: the reference's PRG known answers
    fss::prg::Aes128Soft<4> prg(hkeys, te0, sbox);
    auto o = prg.Gen(kat_seed);
    for (int i = 0; i < 4; ++i) EXPECT(Same(o[i], kat[i]));
    auto l = prg.GenSide(kat_seed, false), r = prg.GenSide(kat_seed, true);
    EXPECT(Same(l[0], kat[0]) && Same(l[1], kat[1]) && Same(r[0], kat[2]) && Same(r[1], kat[3]));
    fss::prg::Aes128Soft<2> prg2(hkeys, te0, sbox);
    auto o2 = prg2.Gen(kat_seed);
    EXPECT(Same(o2[0], kat[0]) && Same(o2[1], kat[1]));
    EXPECT(Same(prg2.GenSide(kat_seed, true)[0], kat[1]) && Same(prg2.GenSide(kat_seed, false)[0], kat[0]));
    fss::prg::Aes128Soft<1> prg1(hkeys, te0, sbox);
    EXPECT(Same(prg1.Gen(kat_seed)[0], kat[0]));
  }
  {  // the same on the device: LDS tables, tables in global memory, and the byte-level Encrypt
    uint32_t* d_te0 = ToDevice(te0, 256);
    uint8_t* d_sbox = ToDevice(sbox, 256);
    int4* d_out = DeviceBuf<int4>(9);
    PrgKernel<<<1, 64>>>(d_out, d_te0, d_sbox, kat_seed);
    HIP_OK(hipDeviceSynchronize());
    auto o = ToHost(d_out, 9);
    for (int i = 0; i < 4; ++i) EXPECT(Same(o[i], kat[i]) && Same(o[4 + i], kat[i]));
    EXPECT(Same(o[8], kat[0]));
  }
  // ---- 2^20 keys, one per thread: generated and evaluated in the reference's kernel shape, against the engine
  constexpr int kN = 1 << 20;
  std::vector<int4> seeds(2 * size_t(kN)), seeds0(kN), seeds1(kN), betas(kN);
  std::vector<uint32_t> alphas(kN), xs(kN);
  for (int i = 0; i < kN; ++i) {
    seeds0[i] = seeds[2 * size_t(i)] = RandSeed();
    seeds1[i] = seeds[2 * size_t(i) + 1] = RandSeed();
    betas[i] = RandSeed();
    alphas[i] = static_cast<uint32_t>(Next()) & ((1u << kBits) - 1);
    xs[i] = i % 2 ? alphas[i] : static_cast<uint32_t>(Next()) & ((1u << kBits) - 1);
  }
  int4* d_seeds = ToDevice(seeds.data(), seeds.size());
  int4* d_seeds0 = ToDevice(seeds0.data(), kN);
  int4* d_seeds1 = ToDevice(seeds1.data(), kN);
  int4* d_betas = ToDevice(betas.data(), kN);
  uint32_t* d_alphas = ToDevice(alphas.data(), kN);
  uint32_t* d_xs = ToDevice(xs.data(), kN);
  Dpf::Cw* d_cws = DeviceBuf<Dpf::Cw>(size_t(kBits + 1) * kN);
  int4* d_y0 = DeviceBuf<int4>(kN);
  int4* d_y1 = DeviceBuf<int4>(kN);
  GenKernel<<<kN / kThreads, kThreads>>>(d_cws, d_seeds, d_alphas, d_betas, kN);
  EvalKernel<<<kN / kThreads, kThreads>>>(d_y0, false, d_seeds0, d_cws, d_xs, kN);
  EvalKernel<<<kN / kThreads, kThreads>>>(d_y1, true, d_seeds1, d_cws, d_xs, kN);
  HIP_OK(hipDeviceSynchronize());
  auto y0 = ToHost(d_y0, kN), y1 = ToHost(d_y1, kN);
  int bad = 0;
  for (int i = 0; i < kN; ++i) bad += !Same(Recon<G>(y0[i], y1[i]), xs[i] == alphas[i] ? G::From(betas[i]).Into() : kZero);
  EXPECT(bad == 0);
  // the engine on the same inputs: keys (every byte of every slot) and both parties' shares
  const unsigned char* ekeys[2] = {p2::k0, p2::k1};
  Mmo2 eprg(ekeys);
  DpfE edpf{eprg};
  DpfE::Cw* d_cws_e = DeviceBuf<DpfE::Cw>(size_t(kBits + 1) * kN);
  edpf.GenBatch(d_cws_e, d_seeds, d_alphas, d_betas, kN);
  HIP_OK(hipDeviceSynchronize());
  {
    auto a = ToHost(reinterpret_cast<const int4*>(d_cws), size_t(2) * (kBits + 1) * kN);
    auto e = ToHost(reinterpret_cast<const int4*>(d_cws_e), size_t(2) * (kBits + 1) * kN);
    EXPECT(std::memcmp(a.data(), e.data(), a.size() * sizeof(int4)) == 0);
  }
  int4* cw_s = DeviceBuf<int4>(size_t(kBits) * kN);
  uint32_t* extra = DeviceBuf<uint32_t>(kN);
  int4* out_cw = DeviceBuf<int4>(kN);
  int4* d_ye = DeviceBuf<int4>(kN);
  fss::gpu::DpfRelayoutGpu<kBits, G, Mmo2, uint32_t>(d_cws_e, kN, cw_s, extra, out_cw);
  for (int b = 0; b < 2; ++b) {
    fss::gpu::DpfEvalPointGpu<4>(b != 0, b ? d_seeds1 : d_seeds0, cw_s, extra, out_cw, d_xs, d_ye, kN, edpf);
    HIP_OK(hipDeviceSynchronize());
    auto ye = ToHost(d_ye, kN);
    EXPECT(std::memcmp(ye.data(), (b ? y1 : y0).data(), sizeof(int4) * kN) == 0);
  }
  // a scheme over Aes128Soft is an engine scheme too: the launchers take it as they take Aes128Mmo
  {
    Soft2 hprg(hkeys, te0, sbox);
    Dpf hdpf{hprg};
    static_assert(Dpf::kEngine);
    fss::gpu::DpfEvalPointGpu<4>(false, d_seeds0, cw_s, extra, out_cw, d_xs, d_ye, kN, hdpf);
    HIP_OK(hipDeviceSynchronize());
    auto ye = ToHost(d_ye, kN);
    EXPECT(std::memcmp(ye.data(), y0.data(), sizeof(int4) * kN) == 0);
    // ... and the same host-built object inside a kernel, per thread (party 1)
    HostBuiltEvalKernel<<<kN / kThreads, kThreads>>>(hdpf, d_ye, true, d_seeds1, d_cws, d_xs, kN);
    HIP_OK(hipDeviceSynchronize());
    ye = ToHost(d_ye, kN);
    EXPECT(std::memcmp(ye.data(), y1.data(), sizeof(int4) * kN) == 0);
  }
  // ---- DCF over Aes128Soft<4> (GenSide takes {s, v} of one side): 4096 keys of n = 16 against the engine
  {
    constexpr int kM = 1 << 12;
    const unsigned char* k4[4] = {p2::k0, p2::k1, p2::k2, p2::k3};
    fss::prg::Aes128Mmo<4> eprg4(k4);
    DcfE edcf{eprg4};
    std::vector<uint32_t> a16(kM), x16(kM);
    for (int i = 0; i < kM; ++i) {
      a16[i] = alphas[i] & 0xffffu;
      x16[i] = i % 3 == 0 ? a16[i] : static_cast<uint32_t>(Next()) & 0xffffu;
    }
    uint32_t* d_a16 = ToDevice(a16.data(), kM);
    uint32_t* d_x16 = ToDevice(x16.data(), kM);
    DcfE::Cw* d_c = DeviceBuf<DcfE::Cw>(size_t(17) * kM);
    edcf.GenBatch(d_c, d_seeds, d_a16, d_betas, kM);
    int4* d_s = DeviceBuf<int4>(size_t(16) * kM);
    int4* d_v = DeviceBuf<int4>(size_t(16) * kM);
    int4* d_o = DeviceBuf<int4>(kM);
    int4* d_yt = DeviceBuf<int4>(kM);
    int4* d_yw = DeviceBuf<int4>(kM);
    fss::gpu::DcfRelayoutGpu<16, G, fss::prg::Aes128Mmo<4>, uint32_t>(d_c, kM, d_s, d_v, d_o);
    for (int b = 0; b < 2; ++b) {
      DcfEvalKernel<<<kM / kThreads, kThreads>>>(d_yt, b != 0, b ? d_seeds1 : d_seeds0, reinterpret_cast<const DcfS::Cw*>(d_c), d_x16, kM);
      fss::gpu::DcfEvalPointGpu<4>(b != 0, b ? d_seeds1 : d_seeds0, d_s, d_v, d_o, d_x16, d_yw, kM, edcf);
      HIP_OK(hipDeviceSynchronize());
      auto yt = ToHost(d_yt, kM), yw = ToHost(d_yw, kM);
      EXPECT(std::memcmp(yt.data(), yw.data(), sizeof(int4) * kM) == 0);
    }
  }
  // ---- lanes with keys of their own (the per-lane fallback): odd threads on the other key pair
  {
    constexpr int kM = 1 << 12;
    const unsigned char o0[16] = {9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6};
    unsigned char o1[16];
    for (int i = 0; i < 16; ++i) o1[i] = static_cast<unsigned char>(0xa0 + i);
    const unsigned char* okeys[2] = {o0, o1};
    Mmo2 oprg(okeys);
    DpfE odpf{oprg};
    // keys of the odd threads are regenerated under the other key pair by the engine
    DpfE::Cw* d_cws_o = DeviceBuf<DpfE::Cw>(size_t(kBits + 1) * kM);
    odpf.GenBatch(d_cws_o, d_seeds, d_alphas, d_betas, kM);
    HIP_OK(hipDeviceSynchronize());
    auto ca = ToHost(d_cws_e, size_t(kBits + 1) * kM);
    auto co = ToHost(d_cws_o, size_t(kBits + 1) * kM);
    for (int i = 1; i < kM; i += 2)
      std::memcpy(&ca[size_t(i) * (kBits + 1)], &co[size_t(i) * (kBits + 1)], sizeof(DpfE::Cw) * (kBits + 1));
    Dpf::Cw* d_mix = reinterpret_cast<Dpf::Cw*>(ToDevice(ca.data(), ca.size()));
    int4* d_ym = DeviceBuf<int4>(kM);
    MixedKeysEvalKernel<<<kM / kThreads, kThreads>>>(d_ym, d_seeds0, d_mix, d_xs, kM);
    HIP_OK(hipDeviceSynchronize());
    auto ym = ToHost(d_ym, kM);
    // expected: even keys from the run above, odd keys from the engine under the other pair
    int4* c_s = DeviceBuf<int4>(size_t(kBits) * kM);
    uint32_t* c_x = DeviceBuf<uint32_t>(kM);
    int4* c_o = DeviceBuf<int4>(kM);
    int4* d_yo = DeviceBuf<int4>(kM);
    fss::gpu::DpfRelayoutGpu<kBits, G, Mmo2, uint32_t>(d_cws_o, kM, c_s, c_x, c_o);
    fss::gpu::DpfEvalPointGpu<4>(false, d_seeds0, c_s, c_x, c_o, d_xs, d_yo, kM, odpf);
    HIP_OK(hipDeviceSynchronize());
    auto yo = ToHost(d_yo, kM);
    bad = 0;
    for (int i = 0; i < kM; ++i) bad += !Same(ym[i], i & 1 ? yo[i] : y0[i]);
    EXPECT(bad == 0);
  }
  // ---- rate of the reference-shaped Eval kernel: n = 20, 2^20 keys, one per thread (the reference publishes
  // 339.6 M evals/s for this kernel on an RTX PRO 5000, README.md:377)
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 8; ++rep) {
    float ms = 0;
    HIP_OK(hipEventRecord(e0));
    for (int k = 0; k < 4; ++k) EvalKernel<<<kN / kThreads, kThreads>>>(d_y0, false, d_seeds0, d_cws, d_xs, kN);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (rep >= 2 && ms / 4 < best) best = ms / 4;
  }
  const double rate = kN / (best * 1e-3);
  EXPECT(rate >= 340e6);
  // the bank-replicated opt-in: same shares, and its rate
  float best_b = 1e30f;
  {
    std::vector<uint32_t> hb(ad::kBankedTe0Words);
    ad::InitTe0Banked(hb.data(), 0, 1);
    EXPECT(hb[0] == 0xc66363a5u && hb[31] == 0xc66363a5u && hb[32] == 0xf87c7c84u);
    SoftB2 hprg(hkeys, hb.data(), sbox);  // host Gen on the banked table
    EXPECT(Same(hprg.Gen(kat_seed)[1], kat[1]) && Same(hprg.GenSide(kat_seed, false)[0], kat[0]));
    int4* d_yb = DeviceBuf<int4>(kN);
    for (int b = 0; b < 2; ++b) {
      BankedEvalKernel<<<kN / kThreads, kThreads>>>(d_yb, b != 0, b ? d_seeds1 : d_seeds0, reinterpret_cast<const DpfB::Cw*>(d_cws), d_xs, kN);
      HIP_OK(hipDeviceSynchronize());
      auto yb = ToHost(d_yb, kN);
      EXPECT(std::memcmp(yb.data(), (b ? y1 : y0).data(), sizeof(int4) * kN) == 0);
    }
    for (int rep = 0; rep < 8; ++rep) {
      float ms = 0;
      HIP_OK(hipEventRecord(e0));
      for (int k = 0; k < 4; ++k)
        BankedEvalKernel<<<kN / kThreads, kThreads>>>(d_yb, false, d_seeds0, reinterpret_cast<const DpfB::Cw*>(d_cws), d_xs, kN);
      HIP_OK(hipEventRecord(e1));
      HIP_OK(hipEventSynchronize(e1));
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2 && ms / 4 < best_b) best_b = ms / 4;
    }
  }
  std::printf("Aes128Soft  DPF n=20, 2^20 keys, the reference's per-thread kernel shape: == engine (keys, shares), reconstructs: %s\n",
              g_fail == before ? "ok" : "FAILED");
  std::printf("     (per-thread Eval kernel, 256 threads per block: %.3f ms = %.1f M evals/s = %.1f G AES blocks/s computed; "
              "reference: 339.6 M evals/s)\n", best, rate / 1e6, rate * kBits / 1e9);
  std::printf("     (the same kernel with fss::prg::Aes128SoftBanked, 32 KiB bank-replicated table: %.3f ms = %.1f M evals/s)\n",
              best_b, kN / (best_b * 1e-3) / 1e6);
}
}  // namespace p4

// usage: dropin_device_test [--seed N] [--no-timing]   (another stream of seeds, alphas and points for a soak run; the
// known-answer parts do not depend on it)
int main(int argc, char** argv) {
  bool timing = true;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--seed") && i + 1 < argc) g_rng = 0x5EED0000u + std::strtoull(argv[++i], nullptr, 0);
    else if (!std::strcmp(argv[i], "--no-timing")) timing = false;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "no GPU\n");
    return 3;
  }
  try {
    p1::Run();
    p2::Kats();
    p2::Batches();
    p2::Groups();
    p2::KeySets();
    p2::Timing<18>();
    if (timing) p2::Timing<21>();
    p2::Wide<18>(false);
    if (timing) p2::Wide<21>(true);
    p3::Run();
    p4::Run();
    fss::ReleaseEngines();
  } catch (const fss::Error& e) {
    std::fprintf(stderr, "fss::Error: %s\n", e.what());
    return 2;
  }
  if (g_fail) {
    std::fprintf(stderr, "%d check(s) FAILED\n", g_fail);
    return 1;
  }
  std::printf("OK\n");
  return 0;
}
