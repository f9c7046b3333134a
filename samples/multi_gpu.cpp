// One evaluation job across every GPU of the node, from C++ (include/fss_amd_multi.hpp: fss::gpu::DeviceSet).
// What each device runs is the reference's launcher on its share (src/bench_gpu.cu:604-627 -> point_eval_gpu.cuh:448-460,
// eval_all_gpu.cuh:478-488); the split is fss_amd_batch_slice / fss_amd_domain_block, no collective.
//   ./multi_gpu [log2_points=24] [eval_all_in_bits=28]
// Prints per-device kernel times and the job's rate (total units / slowest device); checks the FSS property on a sample.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>

#include "fss_amd.hpp"

int main(int argc, char** argv) {
  const int lg = argc > 1 ? std::atoi(argv[1]) : 24;
  const int64_t n = int64_t(1) << lg;
  static unsigned char k0[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  static unsigned char k1[16] = {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1};
  const unsigned char* keys[2] = {k0, k1};
  using Prg = fss::prg::Aes128Mmo<2>;
  using Group = fss::group::Uint<uint64_t>;
  using DpfT = fss::Dpf<32, Group, Prg, uint32_t>;
  try {
    DpfT dpf{Prg(keys)};
    int4 seeds[2] = {{0x11111111, 0x22222222, 0x33333333, 0x44444440}, {0x55555555, 0x66666666, 0x77777777, (int)0x88888880u}};
    const uint32_t alpha = 0x12345678u;
    const int4 beta{7, 0, 0, 0};
    std::vector<DpfT::Cw> cws(33);
    dpf.Gen(cws.data(), seeds, alpha, beta);  // the dealer (on the GPU, through the host signature)
    std::vector<uint32_t> xs(static_cast<size_t>(n));
    std::mt19937 rng(1);
    for (auto& x : xs) x = rng();
    xs[n / 2] = alpha;

    fss::gpu::DeviceSet set = fss::gpu::DeviceSet::AllDevices(dpf);
    std::printf("%d device(s)\n", set.size());
    fss::gpu::DeviceSet::Result y[2];
    std::vector<int4> host[2];
    (void)set.EvalPointsSharded(dpf, false, seeds[0], cws.data(), xs.data(), n);  // warm-up: buffers sized, clocks up
    for (int b = 0; b < 2; ++b) {
      y[b] = set.EvalPointsSharded(dpf, b != 0, seeds[b], cws.data(), xs.data(), n);
      host[b].resize(xs.size());
      for (const auto& s : y[b].shards) {
        set.ToHost(s, host[b].data() + s.first);
        if (b == 0) std::printf("  device %d: points [%lld, %lld)  %.3f ms\n", s.device, (long long)s.first, (long long)(s.first + s.count), s.ms);
      }
    }
    std::printf("DPF kInBits=32, 1 key, 2^%d points: %.3f ms (slowest device) = %.2f G evals/s\n", lg, y[0].max_ms, n / y[0].max_ms / 1e6);
    {  // the same points left on the devices: only the key crosses PCIe per call
      set.UploadPointsSharded(xs.data(), n);
      const fss::gpu::DeviceSet::Result r = set.EvalResidentPointsSharded(dpf, false, seeds[0], cws.data());
      std::vector<int4> again(xs.size());
      for (const auto& s : r.shards) set.ToHost(s, again.data() + s.first);
      const bool same = std::memcmp(again.data(), host[0].data(), again.size() * sizeof(int4)) == 0;
      std::printf("  ... on resident points: %.3f ms (slowest device), %s\n", r.max_ms, same ? "same shares" : "DIFFERENT SHARES");
      if (!same) return 1;
    }
    int64_t bad = 0;
    for (int64_t i = 0; i < n; i += (i == n / 2 - 1 ? 1 : 97)) {
      const uint64_t a = (uint32_t)host[0][i].x | (uint64_t)(uint32_t)host[0][i].y << 32, c = (uint32_t)host[1][i].x | (uint64_t)(uint32_t)host[1][i].y << 32;
      bad += (a + c) != (xs[i] == alpha ? 7u : 0u);
    }
    if ((set.size() & (set.size() - 1)) == 0) {  // the full domain in aligned blocks needs a power of two of devices
      constexpr int kAll = 28;
      using DpfA = fss::Dpf<kAll, Group, Prg, uint32_t>;
      DpfA dpa{Prg(keys)};
      std::vector<DpfA::Cw> acws(kAll + 1);
      dpa.Gen(acws.data(), seeds, alpha & ((1u << kAll) - 1), beta);
      (void)set.EvalAllSharded(dpa, false, seeds[0], acws.data());  // warm-up
      const auto r = set.EvalAllSharded(dpa, false, seeds[0], acws.data());
      for (const auto& s : r.shards) std::printf("  device %d: leaves [%lld, +%lld)  %.3f ms\n", s.device, (long long)s.first, (long long)s.count, s.ms);
      std::printf("DPF EvalAll kInBits=%d: %.3f ms (slowest device) = %.2f G leaves/s\n", kAll, r.max_ms, (double)(int64_t(1) << kAll) / r.max_ms / 1e6);
    }
    std::printf(bad ? "FAILED: %lld sampled points do not reconstruct\n" : "OK\n", (long long)bad);
    return bad ? 1 : 0;
  } catch (const fss::Error& e) {
    std::fprintf(stderr, "fss::Error: %s\n", e.what());
    return 3;
  }
}
