// fss::gpu::DeviceSet (include/fss_amd_multi.hpp): a job cut across the members of a device set gives, block for block,
// the bytes of one launch on one device.  On a one-GPU box the set is EIGHT members that share device 0 (one context and
// one stream each, one host thread each): what an 8-GPU node runs, minus the parallelism.
//   headline    DPF kInBits=32, one key, 2^24 points (BASELINE metric)           -> EvalPointsSharded
//   configs[3]  DPF EvalAll kInBits=28, the domain in 8 blocks of 2^25 leaves     -> EvalAllSharded
//   configs[4]  Half-Tree DPF and Grotto DCF kInBits=64, 2^24 independent keys x 1 pt -> GenKeysSharded (the keys generated
//               on the members, level-major) + EvalResidentKeysSharded; and EvalKeysSharded on 2^18 keys that come from host
//               memory, key-major as Gen writes them
//   + DCF points, DPF / DCF keys, Grotto EvalAll (the one result with a carry across blocks), uneven slices (7 members).
// Every device on the box is used as well when there is more than one.  Exit code 0 and a final "OK" line on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
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

static unsigned char k0[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static unsigned char k1[16] = {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1};
static unsigned char k2[16] = {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8};
static unsigned char k3[16] = {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1};
static const int4 kSeeds[2] = {{0x11111111, 0x22222222, 0x33333333, 0x44444440},
                               {0x55555555, 0x66666666, 0x77777777, static_cast<int>(0x88888880u)}};
using U64 = fss::group::Uint<uint64_t>;
using fss::gpu::DeviceSet;

template <typename T>
static T* ToDevice(const T* host, size_t n) {
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, n * sizeof(T)));
  HIP_OK(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
  return d;
}
// the shards of `res`, concatenated on the host
template <typename T>
static std::vector<T> Gather(const DeviceSet& set, const DeviceSet::Result& res, size_t total) {
  std::vector<T> out(total);
  size_t at = 0;
  for (const auto& s : res.shards) {
    EXPECT(static_cast<size_t>(s.first) == at);
    set.ToHost(s, out.data() + at);
    at += static_cast<size_t>(s.count);
  }
  EXPECT(at == total);
  return out;
}
template <typename T>
static std::vector<T> FromDevice(const T* d, size_t n) {
  std::vector<T> h(n);
  HIP_OK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost));
  return h;
}
template <typename T>
static bool SameBytes(const std::vector<T>& a, const std::vector<T>& b) {
  return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0;
}

// rows [0, rows) x columns [lo, lo + cnt) of a host [rows][n] array, as a [rows][cnt] array in the memory of device `dev`
// (what a caller's own kernels on that GPU would have produced there); the current device is restored
template <typename T>
static T* SliceOn(int dev, const std::vector<T>& full, size_t rows, size_t n, size_t lo, size_t cnt) {
  std::vector<T> part(rows * cnt);
  for (size_t r = 0; r < rows; ++r) std::memcpy(part.data() + r * cnt, full.data() + r * n + lo, cnt * sizeof(T));
  int prev = 0;
  HIP_OK(hipGetDevice(&prev));
  HIP_OK(hipSetDevice(dev));
  T* d = nullptr;
  HIP_OK(hipMalloc(&d, (part.size() ? part.size() : 1) * sizeof(T)));
  if (!part.empty()) HIP_OK(hipMemcpy(d, part.data(), part.size() * sizeof(T), hipMemcpyHostToDevice));
  HIP_OK(hipSetDevice(prev));
  return d;
}
struct OnDeviceArrays {  // frees what SliceOn handed out
  std::vector<std::pair<int, void*>> v;
  template <typename T>
  T* Keep(int dev, T* p) { v.emplace_back(dev, (void*)p); return p; }
  ~OnDeviceArrays() {
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (auto& e : v) { (void)hipSetDevice(e.first); (void)hipFree(e.second); }
    (void)hipSetDevice(prev);
  }
};

static void Headline(const std::vector<int>& devs, int64_t n) {
  const int before = g_fail;
  using Prg = fss::prg::Aes128Mmo<2>;
  using DpfT = fss::Dpf<32, U64, Prg, uint32_t>;
  const unsigned char* keys[2] = {k0, k1};
  DpfT dpf{Prg(keys)};
  int4 seeds[2] = {kSeeds[0], kSeeds[1]};
  std::vector<DpfT::Cw> cws(33);
  dpf.Gen(cws.data(), seeds, 0x12345678u, int4{7, 0, 0, 0});
  std::vector<uint32_t> xs(static_cast<size_t>(n));
  std::mt19937 rng(5);
  for (auto& x : xs) x = rng();
  xs[3] = 0x12345678u;
  // one launch on device 0 (the plain walk from the root: the per-point traversal the metric names)
  int4* d_s0 = ToDevice(&seeds[0], 1);
  DpfT::Cw* d_cws = ToDevice(cws.data(), 33);
  uint32_t* d_xs = ToDevice(xs.data(), xs.size());
  int4* d_ys = nullptr;
  HIP_OK(hipMalloc(&d_ys, xs.size() * sizeof(int4)));
  fss_amd_ctx_set_points_plan(dpf.handle(), FSS_AMD_POINTS_PLAIN);
  dpf.EvalPoints(false, d_s0, d_cws, d_xs, d_ys, n);
  HIP_OK(hipDeviceSynchronize());
  fss_amd_ctx_set_points_plan(dpf.handle(), FSS_AMD_POINTS_AUTO);
  const std::vector<int4> one = FromDevice(d_ys, xs.size());
  HIP_OK(hipFree(d_xs)); HIP_OK(hipFree(d_ys));
  DeviceSet set(dpf, devs);
  for (int plan : {FSS_AMD_POINTS_PLAIN, FSS_AMD_POINTS_AUTO}) {  // each member's slice from the root / by the shared-prefix plan
    set.SetPointsPlan(plan);
    const DeviceSet::Result res = set.EvalPointsSharded(dpf, false, seeds[0], cws.data(), xs.data(), n);
    EXPECT(res.shards.size() == devs.size() && res.max_ms > 0.f);
    EXPECT(SameBytes(Gather<int4>(set, res, xs.size()), one));
    std::printf("  DPF n=32, 2^%d points over %zu members (%s): %.3f ms max over members (a first call; members that share a GPU overlap)\n",
                (int)std::log2((double)n), devs.size(), plan == FSS_AMD_POINTS_PLAIN ? "plain" : "auto", res.max_ms);
  }
  {  // the points already ON the members' devices (EvalPointsShardedOn: device pointers in, as the reference's launchers
     // take them): uneven parts, an empty one, the last member writing into the caller's own buffer
    OnDeviceArrays keep;
    const int world = static_cast<int>(devs.size());
    std::vector<DeviceSet::PointsOn<uint32_t>> per(devs.size());
    std::vector<int64_t> cut(devs.size() + 1, 0);
    for (int r = 0; r < world; ++r) cut[r + 1] = r + 1 == world ? n : std::min<int64_t>(n, cut[r] + (r == 1 ? 0 : n / world + 17 * r));
    for (int r = 0; r < world; ++r) {
      const size_t lo = static_cast<size_t>(cut[r]), cnt = static_cast<size_t>(cut[r + 1] - cut[r]);
      per[r].xs = keep.Keep(devs[r], SliceOn(devs[r], xs, 1, xs.size(), lo, cnt));
      per[r].n = static_cast<int64_t>(cnt);
    }
    int prev = 0;
    HIP_OK(hipGetDevice(&prev));
    HIP_OK(hipSetDevice(devs.back()));
    int4* own = nullptr;
    HIP_OK(hipMalloc(&own, (per.back().n ? per.back().n : 1) * sizeof(int4)));
    HIP_OK(hipSetDevice(prev));
    per.back().ys = keep.Keep(devs.back(), own);
    for (int plan : {FSS_AMD_POINTS_PLAIN, FSS_AMD_POINTS_AUTO}) {
      set.SetPointsPlan(plan);
      const DeviceSet::Result res = set.EvalPointsShardedOn(dpf, false, seeds[0], cws.data(), std::span<const DeviceSet::PointsOn<uint32_t>>(per));
      EXPECT(res.shards.size() == devs.size() && res.shards.back().ys == own);
      EXPECT(SameBytes(Gather<int4>(set, res, xs.size()), one));
    }
    if (world > 1) {
      EXPECT(per[1].n == 0);
      bool refused = false;  // one entry per member
      try { (void)set.EvalPointsShardedOn(dpf, false, seeds[0], cws.data(), std::span<const DeviceSet::PointsOn<uint32_t>>(per.data(), 1)); }
      catch (const fss::Error& e) { refused = e.code == FSS_AMD_EINVAL; }
      EXPECT(refused);
    }
    {  // host memory where device memory is expected: refused with the member's own code (EINVAL, not a generic HIP error)
      auto bad = per;
      bad[0].xs = xs.data();
      bool refused = false;
      try { (void)set.EvalPointsShardedOn(dpf, false, seeds[0], cws.data(), std::span<const DeviceSet::PointsOn<uint32_t>>(bad)); }
      catch (const fss::Error& e) { refused = e.code == FSS_AMD_EINVAL; }
      EXPECT(refused);
    }
    std::printf("  DPF n=32, 2^%d points already on the members' devices (uneven parts, caller-owned output): same bytes\n", (int)std::log2((double)n));
  }
  {  // the points resident on the members: two keys over one upload, and the refusals
    set.SetPointsPlan(FSS_AMD_POINTS_AUTO);
    bool refused = false;
    try { (void)set.EvalResidentPointsSharded(dpf, false, seeds[0], cws.data()); } catch (const fss::Error&) { refused = true; }
    EXPECT(refused);  // nothing uploaded yet
    set.UploadPointsSharded(xs.data(), n);
    const DeviceSet::Result r0 = set.EvalResidentPointsSharded(dpf, false, seeds[0], cws.data());
    EXPECT(SameBytes(Gather<int4>(set, r0, xs.size()), one));
    std::vector<DpfT::Cw> cws2(33);
    int4 seeds2[2] = {kSeeds[1], kSeeds[0]};
    dpf.Gen(cws2.data(), seeds2, 0x0badcafeu, int4{9, 0, 0, 0});
    const DeviceSet::Result want2 = set.EvalPointsSharded(dpf, true, seeds2[1], cws2.data(), xs.data(), n);
    const std::vector<int4> w2 = Gather<int4>(set, want2, xs.size());
    const DeviceSet::Result r2 = set.EvalResidentPointsSharded(dpf, true, seeds2[1], cws2.data());
    EXPECT(SameBytes(Gather<int4>(set, r2, xs.size()), w2));
    EXPECT(!SameBytes(w2, one));
    refused = false;
    try {  // a resident batch of 4-byte points under a scheme that takes 8-byte ones
      using Dpf64 = fss::Dpf<64, U64, Prg, uint64_t>;
      Dpf64 d64{Prg(keys)};
      std::vector<Dpf64::Cw> c64(65);
      (void)set.EvalResidentPointsSharded(d64, false, seeds[0], c64.data());
    } catch (const fss::Error&) { refused = true; }
    EXPECT(refused);
    std::printf("  DPF n=32, 2^%d resident points over %zu members: %.3f ms max over members; a second key on the same upload: %.3f\n",
                (int)std::log2((double)n), devs.size(), r0.max_ms, r2.max_ms);
  }
  {  // DCF on the same points, uneven slices: seven members
    using Prg4 = fss::prg::Aes128Mmo<4>;
    using DcfT = fss::Dcf<32, U64, Prg4, uint32_t>;
    const unsigned char* keys4[4] = {k0, k1, k2, k3};
    DcfT dcf{Prg4(keys4)};
    std::vector<DcfT::Cw> dcws(33);
    dcf.Gen(dcws.data(), seeds, 0x12345678u, int4{7, 0, 0, 0});
    const int64_t m = (n >> 3) + 13;
    DcfT::Cw* dd = ToDevice(dcws.data(), 33);
    uint32_t* dx = ToDevice(xs.data(), static_cast<size_t>(m));
    int4* dy = nullptr;
    HIP_OK(hipMalloc(&dy, static_cast<size_t>(m) * sizeof(int4)));
    dcf.EvalPoints(true, d_s0, dd, dx, dy, m);  // (party 1's outputs of party 0's seed: any seed will do for a byte comparison)
    HIP_OK(hipDeviceSynchronize());
    const std::vector<int4> done = FromDevice(dy, static_cast<size_t>(m));
    std::vector<int> seven(7, devs[0]);
    DeviceSet set7(dcf, seven);
    const DeviceSet::Result res = set7.EvalPointsSharded(dcf, true, seeds[0], dcws.data(), xs.data(), m);
    EXPECT(SameBytes(Gather<int4>(set7, res, static_cast<size_t>(m)), done));
    HIP_OK(hipFree(dd)); HIP_OK(hipFree(dx)); HIP_OK(hipFree(dy));
  }
  HIP_OK(hipFree(d_s0)); HIP_OK(hipFree(d_cws));
  std::printf("headline batch + DCF points across the set: %s\n", g_fail == before ? "ok" : "FAILED");
}

template <int in_bits>
static void EvalAllBlocks(const std::vector<int>& devs) {
  const int before = g_fail;
  using Prg = fss::prg::Aes128Mmo<2>;
  using DpfT = fss::Dpf<in_bits, U64, Prg, uint32_t>;
  const unsigned char* keys[2] = {k0, k1};
  DpfT dpf{Prg(keys)};
  int4 seeds[2] = {kSeeds[0], kSeeds[1]};
  std::vector<typename DpfT::Cw> cws(in_bits + 1);
  dpf.Gen(cws.data(), seeds, 0x1234567u & ((1u << in_bits) - 1), int4{7, 1, 0, 0});
  const size_t N = size_t(1) << in_bits;
  int4* d_s0 = ToDevice(&seeds[1], 1);
  typename DpfT::Cw* d_cws = ToDevice(cws.data(), cws.size());
  int4* d_ys = nullptr;
  void* d_ws = nullptr;
  const size_t ws = DpfT::EvalAllWorkspace();
  HIP_OK(hipMalloc(&d_ys, N * sizeof(int4)));
  HIP_OK(hipMalloc(&d_ws, ws));
  dpf.EvalAllBlock(true, d_s0, d_cws, d_ys, d_ws, ws);
  HIP_OK(hipDeviceSynchronize());
  DeviceSet set(dpf, devs);
  const DeviceSet::Result res = set.EvalAllSharded(dpf, true, seeds[1], cws.data());
  // block by block against the whole domain's bytes, without a second 4 GiB on the host
  std::vector<int4> a, b;
  for (const auto& s : res.shards) {
    a.resize(static_cast<size_t>(s.count));
    set.ToHost(s, a.data());
    b = FromDevice(d_ys + s.first, static_cast<size_t>(s.count));
    EXPECT(SameBytes(a, b));
    EXPECT(s.first % s.count == 0 && s.count == static_cast<int64_t>(N / devs.size()));
  }
  std::printf("  DPF EvalAll n=%d in %zu blocks: %.3f ms max over members\n", in_bits, devs.size(), res.max_ms);
  HIP_OK(hipFree(d_s0)); HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_ys)); HIP_OK(hipFree(d_ws));
  {  // Grotto EvalAll n = 20: the running XOR crosses the blocks, the set applies the carries
    using Gr = fss::GrottoDcf<20, Prg, uint32_t>;
    Gr gd{Prg(keys)};
    std::vector<typename Gr::Cw> gcws(21);
    gd.Gen(gcws.data(), seeds, 777777u);
    std::vector<uint8_t> whole(size_t(1) << 20);
    gd.EvalAll(false, seeds[0], gcws.data(), reinterpret_cast<bool*>(whole.data()));
    DeviceSet gset(gd, devs);
    const DeviceSet::Result gr = gset.EvalAllSharded(gd, false, seeds[0], gcws.data());
    EXPECT(SameBytes(Gather<uint8_t>(gset, gr, whole.size()), whole));
  }
  std::printf("full domain in blocks (DPF n=%d, Grotto n=20 with carries): %s\n", in_bits, g_fail == before ? "ok" : "FAILED");
}

static void IndependentKeys(const std::vector<int>& devs, int64_t nkeys) {
  const int before = g_fail;
  const unsigned char* keys1[1] = {k0};
  const unsigned char* keys2[2] = {k0, k1};
  const unsigned char* keys4[4] = {k0, k1, k2, k3};
  std::mt19937_64 rng(11);
  const size_t nk = static_cast<size_t>(nkeys);
  std::vector<int4> s0s(2 * nk), betas(nk);
  std::vector<uint64_t> alphas(nk), xs(nk);
  for (size_t k = 0; k < nk; ++k) {
    for (int j = 0; j < 2; ++j) s0s[2 * k + j] = int4{(int)rng(), (int)rng(), (int)rng(), (int)(rng() & ~1ull)};
    betas[k] = int4{(int)rng(), (int)rng(), 0, 0};
    alphas[k] = rng();
    xs[k] = (k & 1) ? rng() : alphas[k];
  }
  std::vector<int4> seeds0(nk);
  for (size_t k = 0; k < nk; ++k) seeds0[k] = s0s[2 * k];
  int4* d_s0s = ToDevice(s0s.data(), s0s.size());
  int4* d_seeds0 = ToDevice(seeds0.data(), nk);
  uint64_t* d_al = ToDevice(alphas.data(), nk);
  uint64_t* d_xs = ToDevice(xs.data(), nk);
  int4* d_be = ToDevice(betas.data(), nk);
  int4* d_ys = nullptr;
  HIP_OK(hipMalloc(&d_ys, nk * sizeof(int4)));
  {  // Half-Tree DPF n = 64 (configs[4])
    using Prg = fss::prg::Aes128Mmo<1>;
    using HT = fss::HalfTreeDpf<64, U64, Prg, uint64_t>;
    HT ht{Prg(keys1), int4{0x12345678, (int)0x9abcdef0u, 0x13572468, 0x2468ace0}};
    typename HT::Cw* d_cws = nullptr;
    int4 *d_ocws = nullptr, *d_cw_s = nullptr;
    uint32_t* d_extra = nullptr;
    HIP_OK(hipMalloc(&d_cws, nk * 64 * sizeof(typename HT::Cw)));
    HIP_OK(hipMalloc(&d_ocws, nk * sizeof(int4)));
    HIP_OK(hipMalloc(&d_cw_s, nk * 64 * sizeof(int4)));
    HIP_OK(hipMalloc(&d_extra, nk * sizeof(uint32_t)));
    ht.GenBatch(d_cws, d_ocws, d_s0s, d_al, d_be, nkeys);
    fss::gpu::HalfTreeDpfRelayoutGpu<64, U64, Prg, uint64_t>(d_cws, (int)nkeys, d_cw_s, d_extra);
    fss::gpu::HalfTreeDpfEvalPointGpu(false, d_seeds0, d_cw_s, d_extra, d_ocws, d_xs, d_ys, (int)nkeys, ht);
    HIP_OK(hipDeviceSynchronize());
    const std::vector<int4> one = FromDevice(d_ys, nk);
    const auto h_cws = FromDevice(d_cws, nk * 64);
    const auto h_ocws = FromDevice(d_ocws, nk);
    DeviceSet set(ht, devs);
    const DeviceSet::Result res = set.EvalKeysSharded(ht, false, seeds0.data(), h_cws.data(), h_ocws.data(), xs.data(), nkeys);
    EXPECT(SameBytes(Gather<int4>(set, res, nk), one));
    std::printf("  Half-Tree n=64, %lld keys over %zu members: %.3f ms max\n", (long long)nkeys, devs.size(), res.max_ms);
    {  // the same keys level-major ON the members (EvalKeysShardedOn)
      OnDeviceArrays keep;
      const auto h_cw_s = FromDevice(d_cw_s, nk * 64);
      const auto h_extra = FromDevice(d_extra, nk);
      std::vector<DeviceSet::KeysOn<uint64_t>> per(devs.size());
      for (size_t r = 0; r < devs.size(); ++r) {
        const fss::gpu::Slice sl = fss::gpu::BatchSlice(nkeys, (int)r, (int)devs.size());
        const size_t lo = (size_t)sl.lo, cnt = (size_t)(sl.hi - sl.lo);
        const int d = devs[r];
        per[r].seeds = keep.Keep(d, SliceOn(d, seeds0, 1, nk, lo, cnt));
        per[r].cw_s = keep.Keep(d, SliceOn(d, h_cw_s, 64, nk, lo, cnt));
        per[r].tr_bits = keep.Keep(d, SliceOn(d, h_extra, 1, nk, lo, cnt));
        per[r].out_cw = keep.Keep(d, SliceOn(d, h_ocws, 1, nk, lo, cnt));
        per[r].xs = keep.Keep(d, SliceOn(d, xs, 1, nk, lo, cnt));
        per[r].nkeys = (int64_t)cnt;
      }
      const DeviceSet::Result on = set.EvalKeysShardedOn(ht, false, std::span<const DeviceSet::KeysOn<uint64_t>>(per));
      EXPECT(SameBytes(Gather<int4>(set, on, nk), one));
    }
    HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_ocws)); HIP_OK(hipFree(d_cw_s)); HIP_OK(hipFree(d_extra));
  }
  {  // Grotto DCF n = 64 path walk (configs[4]) and DPF n = 64 on the same DPF-format keys
    using Prg = fss::prg::Aes128Mmo<2>;
    using Gr = fss::GrottoDcf<64, Prg, uint64_t>;
    using DpfT = fss::Dpf<64, U64, Prg, uint64_t>;
    Gr gd{Prg(keys2)};
    DpfT dpf{Prg(keys2)};
    typename DpfT::Cw* d_cws = nullptr;
    int4 *d_cw_s = nullptr, *d_out_cw = nullptr;
    uint64_t* d_tr = nullptr;
    uint8_t* d_yb = nullptr;
    HIP_OK(hipMalloc(&d_cws, nk * 65 * sizeof(typename DpfT::Cw)));
    HIP_OK(hipMalloc(&d_cw_s, nk * 64 * sizeof(int4)));
    HIP_OK(hipMalloc(&d_out_cw, nk * sizeof(int4)));
    HIP_OK(hipMalloc(&d_tr, nk * sizeof(uint64_t)));
    HIP_OK(hipMalloc(&d_yb, nk));
    dpf.GenBatch(d_cws, d_s0s, d_al, d_be, nkeys);
    fss::gpu::DpfRelayoutGpu<64, U64, Prg, uint64_t>(d_cws, nkeys, d_cw_s, d_tr, d_out_cw);
    fss::gpu::DpfEvalPointGpu(false, d_seeds0, d_cw_s, d_tr, d_out_cw, d_xs, d_ys, nkeys, dpf);
    gd.EvalKeysWalk(false, d_seeds0, d_cw_s, d_tr, d_xs, d_yb, nkeys);
    HIP_OK(hipDeviceSynchronize());
    const std::vector<int4> one = FromDevice(d_ys, nk);
    const std::vector<uint8_t> oneb = FromDevice(d_yb, nk);
    const auto h_cws = FromDevice(d_cws, nk * 65);
    DeviceSet set(dpf, devs);
    const DeviceSet::Result res = set.EvalKeysSharded(dpf, false, seeds0.data(), h_cws.data(), xs.data(), nkeys);
    EXPECT(SameBytes(Gather<int4>(set, res, nk), one));
    const DeviceSet::Result gres = set.EvalKeysSharded(gd, false, seeds0.data(), reinterpret_cast<const typename Gr::Cw*>(h_cws.data()),
                                                       xs.data(), nkeys);
    EXPECT(SameBytes(Gather<uint8_t>(set, gres, nk), oneb));
    std::printf("  DPF / Grotto n=64, %lld keys over %zu members: %.3f / %.3f ms max\n", (long long)nkeys, devs.size(), res.max_ms,
                gres.max_ms);
    {  // level-major ON the members: DPF and the Grotto walk on the same arrays
      OnDeviceArrays keep;
      const auto h_cw_s = FromDevice(d_cw_s, nk * 64);
      const auto h_tr = FromDevice(d_tr, nk);
      const auto h_out = FromDevice(d_out_cw, nk);
      std::vector<DeviceSet::KeysOn<uint64_t>> per(devs.size());
      for (size_t r = 0; r < devs.size(); ++r) {
        const fss::gpu::Slice sl = fss::gpu::BatchSlice(nkeys, (int)r, (int)devs.size());
        const size_t lo = (size_t)sl.lo, cnt = (size_t)(sl.hi - sl.lo);
        const int d = devs[r];
        per[r].seeds = keep.Keep(d, SliceOn(d, seeds0, 1, nk, lo, cnt));
        per[r].cw_s = keep.Keep(d, SliceOn(d, h_cw_s, 64, nk, lo, cnt));
        per[r].tr_bits = keep.Keep(d, SliceOn(d, h_tr, 1, nk, lo, cnt));
        per[r].out_cw = keep.Keep(d, SliceOn(d, h_out, 1, nk, lo, cnt));
        per[r].xs = keep.Keep(d, SliceOn(d, xs, 1, nk, lo, cnt));
        per[r].nkeys = (int64_t)cnt;
      }
      const DeviceSet::Result on = set.EvalKeysShardedOn(dpf, false, std::span<const DeviceSet::KeysOn<uint64_t>>(per));
      EXPECT(SameBytes(Gather<int4>(set, on, nk), one));
      const DeviceSet::Result gon = set.EvalKeysShardedOn(gd, false, std::span<const DeviceSet::KeysOn<uint64_t>>(per));
      EXPECT(SameBytes(Gather<uint8_t>(set, gon, nk), oneb));
      auto bad = per;  // a DPF entry without its tr bits
      bad[0].tr_bits = nullptr;
      bool refused = false;
      try { (void)set.EvalKeysShardedOn(dpf, false, std::span<const DeviceSet::KeysOn<uint64_t>>(bad)); }
      catch (const fss::Error& e) { refused = e.code == FSS_AMD_EINVAL; }
      EXPECT(refused);
    }
    HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_cw_s)); HIP_OK(hipFree(d_out_cw)); HIP_OK(hipFree(d_tr)); HIP_OK(hipFree(d_yb));
  }
  {  // DCF n = 64
    using Prg = fss::prg::Aes128Mmo<4>;
    using DcfT = fss::Dcf<64, U64, Prg, uint64_t>;
    DcfT dcf{Prg(keys4)};
    typename DcfT::Cw* d_cws = nullptr;
    int4 *d_cw_s = nullptr, *d_cw_v = nullptr, *d_out_cw = nullptr;
    HIP_OK(hipMalloc(&d_cws, nk * 65 * sizeof(typename DcfT::Cw)));
    HIP_OK(hipMalloc(&d_cw_s, nk * 64 * sizeof(int4)));
    HIP_OK(hipMalloc(&d_cw_v, nk * 64 * sizeof(int4)));
    HIP_OK(hipMalloc(&d_out_cw, nk * sizeof(int4)));
    dcf.GenBatch(d_cws, d_s0s, d_al, d_be, nkeys);
    fss::gpu::DcfRelayoutGpu<64, U64, Prg, uint64_t>(d_cws, (int)nkeys, d_cw_s, d_cw_v, d_out_cw);
    fss::gpu::DcfEvalPointGpu(false, d_seeds0, d_cw_s, d_cw_v, d_out_cw, d_xs, d_ys, (int)nkeys, dcf);
    HIP_OK(hipDeviceSynchronize());
    const std::vector<int4> one = FromDevice(d_ys, nk);
    const auto h_cws = FromDevice(d_cws, nk * 65);
    DeviceSet set(dcf, devs);
    const DeviceSet::Result res = set.EvalKeysSharded(dcf, false, seeds0.data(), h_cws.data(), xs.data(), nkeys);
    EXPECT(SameBytes(Gather<int4>(set, res, nk), one));
    {  // level-major ON the members
      OnDeviceArrays keep;
      const auto h_cw_s = FromDevice(d_cw_s, nk * 64);
      const auto h_cw_v = FromDevice(d_cw_v, nk * 64);
      const auto h_out = FromDevice(d_out_cw, nk);
      std::vector<DeviceSet::KeysOn<uint64_t>> per(devs.size());
      for (size_t r = 0; r < devs.size(); ++r) {
        const fss::gpu::Slice sl = fss::gpu::BatchSlice(nkeys, (int)r, (int)devs.size());
        const size_t lo = (size_t)sl.lo, cnt = (size_t)(sl.hi - sl.lo);
        const int d = devs[r];
        per[r].seeds = keep.Keep(d, SliceOn(d, seeds0, 1, nk, lo, cnt));
        per[r].cw_s = keep.Keep(d, SliceOn(d, h_cw_s, 64, nk, lo, cnt));
        per[r].cw_v = keep.Keep(d, SliceOn(d, h_cw_v, 64, nk, lo, cnt));
        per[r].out_cw = keep.Keep(d, SliceOn(d, h_out, 1, nk, lo, cnt));
        per[r].xs = keep.Keep(d, SliceOn(d, xs, 1, nk, lo, cnt));
        per[r].nkeys = (int64_t)cnt;
      }
      const DeviceSet::Result on = set.EvalKeysShardedOn(dcf, false, std::span<const DeviceSet::KeysOn<uint64_t>>(per));
      EXPECT(SameBytes(Gather<int4>(set, on, nk), one));
    }
    HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_cw_s)); HIP_OK(hipFree(d_cw_v)); HIP_OK(hipFree(d_out_cw));
  }
  HIP_OK(hipFree(d_s0s)); HIP_OK(hipFree(d_seeds0)); HIP_OK(hipFree(d_al)); HIP_OK(hipFree(d_xs)); HIP_OK(hipFree(d_be)); HIP_OK(hipFree(d_ys));
  std::printf("independent keys across the set (Half-Tree, DPF, Grotto, DCF; n=64): %s\n", g_fail == before ? "ok" : "FAILED");
}

// configs[4] at its full size through the set: the keys are GENERATED on the members (GenKeysSharded: each member its
// slice, level-major, from 32 B of seeds + alpha + beta per key) and evaluated where they are (EvalResidentKeysSharded),
// against one Gen + one EvalPoint launch over the whole batch on device 0; both parties, and the FSS property on a sample.
static void ResidentKeys(const std::vector<int>& devs, int64_t nkeys) {
  const int before = g_fail;
  const unsigned char* keys1[1] = {k0};
  const unsigned char* keys2[2] = {k0, k1};
  std::mt19937_64 rng(23);
  const size_t nk = static_cast<size_t>(nkeys);
  std::vector<int4> s0s(2 * nk), betas(nk);
  std::vector<uint64_t> alphas(nk), xs(nk);
  for (size_t k = 0; k < nk; ++k) {
    const uint64_t a = rng(), b2 = rng(), c = rng(), d = rng();
    s0s[2 * k] = int4{(int)a, (int)(a >> 32), (int)b2, (int)((b2 >> 32) & ~1ull)};
    s0s[2 * k + 1] = int4{(int)c, (int)(c >> 32), (int)d, (int)((d >> 32) & ~1ull)};
    const uint64_t e = rng();
    betas[k] = int4{(int)e, (int)(e >> 32), 0, 0};
    alphas[k] = rng();
    xs[k] = (k & 1) ? rng() : alphas[k];
  }
  int4* d_s0s = ToDevice(s0s.data(), s0s.size());
  uint64_t* d_al = ToDevice(alphas.data(), nk);
  uint64_t* d_xs = ToDevice(xs.data(), nk);
  int4* d_be = ToDevice(betas.data(), nk);
  int4 *d_seeds = nullptr, *d_ys = nullptr, *d_cw_s = nullptr, *d_aux = nullptr;
  HIP_OK(hipMalloc(&d_seeds, nk * sizeof(int4)));
  HIP_OK(hipMalloc(&d_ys, nk * sizeof(int4)));
  HIP_OK(hipMalloc(&d_cw_s, nk * 64 * sizeof(int4)));
  HIP_OK(hipMalloc(&d_aux, nk * sizeof(int4)));  // ocws / out_cw
  void* d_bits = nullptr;                        // extra (uint32) / tr_bits (uint64)
  HIP_OK(hipMalloc(&d_bits, nk * sizeof(uint64_t)));
  std::vector<int4> one[2], got[2];
  {  // Half-Tree DPF n = 64
    using Prg = fss::prg::Aes128Mmo<1>;
    using HT = fss::HalfTreeDpf<64, U64, Prg, uint64_t>;
    HT ht{Prg(keys1), int4{0x12345678, (int)0x9abcdef0u, 0x13572468, 0x2468ace0}};
    ht.GenLevelMajor(d_cw_s, static_cast<uint32_t*>(d_bits), d_aux, d_s0s, d_al, d_be, nkeys);
    DeviceSet set(ht, devs);
    set.GenKeysSharded(ht, s0s.data(), alphas.data(), betas.data(), nkeys);
    for (int b = 0; b < 2; ++b) {
      HIP_OK(hipMemcpy2D(d_seeds, 16, reinterpret_cast<const char*>(d_s0s) + 16 * b, 32, 16, nk, hipMemcpyDeviceToDevice));
      fss::gpu::HalfTreeDpfEvalPointGpu(b != 0, d_seeds, d_cw_s, static_cast<const uint32_t*>(d_bits), d_aux, d_xs, d_ys, nkeys, ht);
      HIP_OK(hipDeviceSynchronize());
      one[b] = FromDevice(d_ys, nk);
      const DeviceSet::Result res = set.EvalResidentKeysSharded(ht, b != 0, xs.data());
      got[b] = Gather<int4>(set, res, nk);
      EXPECT(SameBytes(got[b], one[b]));
      if (b == 0) std::printf("  Half-Tree n=64, %lld keys generated and evaluated on %zu members: %.3f ms max\n", (long long)nkeys, devs.size(), res.max_ms);
    }
    int64_t bad = 0;
    for (size_t k = 0; k < nk; k += 997) {
      const uint64_t y0 = (uint32_t)got[0][k].x | (uint64_t)(uint32_t)got[0][k].y << 32, y1 = (uint32_t)got[1][k].x | (uint64_t)(uint32_t)got[1][k].y << 32;
      const uint64_t be = (uint32_t)betas[k].x | (uint64_t)(uint32_t)betas[k].y << 32;
      bad += (y0 + y1) != (xs[k] == alphas[k] ? be : 0);
    }
    EXPECT(bad == 0);
  }
  {  // Grotto DCF n = 64, the path walk: one byte per key
    using Prg = fss::prg::Aes128Mmo<2>;
    using Gr = fss::GrottoDcf<64, Prg, uint64_t>;
    Gr gd{Prg(keys2)};
    uint8_t* d_yb = nullptr;
    HIP_OK(hipMalloc(&d_yb, nk));
    gd.GenLevelMajor(d_cw_s, static_cast<uint64_t*>(d_bits), d_aux, d_s0s, d_al, nkeys);
    DeviceSet set(gd, devs);
    set.GenKeysSharded(gd, s0s.data(), alphas.data(), nkeys);
    std::vector<uint8_t> gb[2];
    for (int b = 0; b < 2; ++b) {
      HIP_OK(hipMemcpy2D(d_seeds, 16, reinterpret_cast<const char*>(d_s0s) + 16 * b, 32, 16, nk, hipMemcpyDeviceToDevice));
      gd.EvalKeysWalk(b != 0, d_seeds, d_cw_s, static_cast<const uint64_t*>(d_bits), d_xs, d_yb, nkeys);
      HIP_OK(hipDeviceSynchronize());
      const std::vector<uint8_t> oneb = FromDevice(d_yb, nk);
      const DeviceSet::Result res = set.EvalResidentKeysSharded(gd, b != 0, xs.data());
      gb[b] = Gather<uint8_t>(set, res, nk);
      EXPECT(SameBytes(gb[b], oneb));
      if (b == 0) std::printf("  Grotto n=64, %lld keys generated and evaluated on %zu members: %.3f ms max\n", (long long)nkeys, devs.size(), res.max_ms);
    }
    int64_t bad = 0;
    for (size_t k = 0; k < nk; k += 997) bad += ((gb[0][k] ^ gb[1][k]) & 1) != (alphas[k] <= xs[k] ? 1 : 0);
    EXPECT(bad == 0);
    HIP_OK(hipFree(d_yb));
  }
  HIP_OK(hipFree(d_s0s)); HIP_OK(hipFree(d_al)); HIP_OK(hipFree(d_xs)); HIP_OK(hipFree(d_be)); HIP_OK(hipFree(d_seeds));
  HIP_OK(hipFree(d_ys)); HIP_OK(hipFree(d_cw_s)); HIP_OK(hipFree(d_aux)); HIP_OK(hipFree(d_bits));
  std::printf("configs[4] with the keys generated on the members (Half-Tree, Grotto; n=64, 2^%d keys): %s\n", (int)std::log2((double)nkeys),
              g_fail == before ? "ok" : "FAILED");
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && std::strcmp(argv[1], "--quick") == 0;  // smaller sizes (development)
  try {
    // the split itself
    int64_t at = 0;
    for (int r = 0; r < 7; ++r) {
      const fss::gpu::Slice s = fss::gpu::BatchSlice(100, r, 7);
      EXPECT(s.lo == at && (s.hi - s.lo == 14 || s.hi - s.lo == 15));
      at = s.hi;
    }
    EXPECT(at == 100);
    EXPECT(fss::gpu::DomainBlock(28, 5, 8).first == 5 * (int64_t(1) << 25) && fss::gpu::DomainBlock(28, 5, 8).nleaves == int64_t(1) << 25);
    EXPECT((fss::gpu::GrottoCarries({1, 0, 1, 1}) == std::vector<uint8_t>{0, 1, 1, 0}));
    int64_t lo, hi;
    EXPECT(fss_amd_batch_slice(10, 3, 3, &lo, &hi) == FSS_AMD_EINVAL && fss_amd_domain_block(28, 0, 3, &lo, &hi) == FSS_AMD_EINVAL);

    {  // refusals: a domain cannot be cut into seven aligned blocks; a device that does not exist
      using Prg = fss::prg::Aes128Mmo<2>;
      using DpfT = fss::Dpf<12, U64, Prg, uint32_t>;
      const unsigned char* keys[2] = {k0, k1};
      DpfT dpf{Prg(keys)};
      int4 seeds[2] = {kSeeds[0], kSeeds[1]};
      std::vector<DpfT::Cw> cws(13);
      dpf.Gen(cws.data(), seeds, 5u, int4{1, 0, 0, 0});
      bool threw = false;
      try {
        DeviceSet seven(dpf, std::vector<int>(7, 0));
        (void)seven.EvalAllSharded(dpf, false, seeds[0], cws.data());
      } catch (const fss::Error&) { threw = true; }
      EXPECT(threw);
      threw = false;
      try { DeviceSet bad(dpf, {4096}); } catch (const fss::Error&) { threw = true; }
      EXPECT(threw);
      DeviceSet one(dpf, {0});  // a set of one is the plain call
      std::vector<uint32_t> xs = {0u, 5u, 4095u};
      const auto r = one.EvalPointsSharded(dpf, false, seeds[0], cws.data(), xs.data(), 3);
      std::vector<int4> y(3);
      one.ToHost(r.shards[0], y.data());
      EXPECT(r.shards.size() == 1 && r.shards[0].count == 3 && std::memcmp(&y[1], &y[1], 16) == 0);
      const int4 want = dpf.Eval(false, seeds[0], cws.data(), 5u);
      EXPECT(std::memcmp(&y[1], &want, sizeof want) == 0);
      const auto none = one.EvalPointsSharded(dpf, false, seeds[0], cws.data(), xs.data(), 0);  // an empty batch
      EXPECT(none.shards[0].count == 0);
    }
    const std::vector<int> eight(8, 0);  // eight members sharing device 0
    Headline(eight, quick ? int64_t(1) << 20 : int64_t(1) << 24);
    if (quick) EvalAllBlocks<22>(eight);
    else EvalAllBlocks<28>(eight);
    IndependentKeys(eight, quick ? int64_t(1) << 14 : int64_t(1) << 18);
    ResidentKeys(eight, quick ? int64_t(1) << 16 : int64_t(1) << 24);
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (ndev > 1) {  // and every GPU of the box, once each (a power of two of them for the domain blocks)
      std::vector<int> all;
      for (int d = 0; d < ndev; ++d) all.push_back(d);
      Headline(all, int64_t(1) << 24);
      while (all.size() & (all.size() - 1)) all.pop_back();
      EvalAllBlocks<28>(all);
      IndependentKeys(all, int64_t(1) << 18);
      ResidentKeys(all, int64_t(1) << 24);
    }
    fss::ReleaseEngines();
  } catch (const fss::Error& e) {
    std::fprintf(stderr, "fss::Error: %s\n", e.what());
    return 3;
  }
  std::printf(g_fail ? "FAILED (%d)\n" : "OK\n", g_fail);
  return g_fail ? 1 : 0;
}
