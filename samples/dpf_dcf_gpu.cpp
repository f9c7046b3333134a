// Batched DPF and DCF on the GPU through the C++ layer (include/fss_amd.hpp).
// Same flow as the reference's samples/dpf_dcf_gpu.cu:84-236 -- one key per lane: Gen on the
// device, evaluate both parties, reconstruct on the host and self-check -- with the AES-128-MMO
// PRG instead of ChaCha, plus the level-major relayout + EvalPointGpu launchers.
//
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude samples/dpf_dcf_gpu.cpp \
//              -Lfss_amd -lfss_amd -Wl,-rpath,$PWD/fss_amd -o samples/dpf_dcf_gpu
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fss_amd.hpp"

#define HIP_OK(e)                                                      \
  do {                                                                 \
    hipError_t r_ = (e);                                               \
    if (r_ != hipSuccess) { std::fprintf(stderr, "HIP error %d at %s:%d\n", (int)r_, __FILE__, __LINE__); return 2; } \
  } while (0)

constexpr int kInBits = 32;
constexpr int kN = 1024;
using In = uint32_t;
using Group = fss::group::Uint<uint64_t>;
using DpfPrg = fss::prg::Aes128Mmo<2>;
using DcfPrg = fss::prg::Aes128Mmo<4>;
using DpfT = fss::Dpf<kInBits, Group, DpfPrg, In>;
using DcfT = fss::Dcf<kInBits, Group, DcfPrg, In, fss::DcfPred::kLt>;

template <typename T>
T* ToDevice(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
static uint64_t Val(const int4& v) { return (uint64_t)(uint32_t)v.x | ((uint64_t)(uint32_t)v.y << 32); }

int main() {
  const unsigned char k0[16] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  const unsigned char k1[16] = {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1};
  const unsigned char k2[16] = {1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8};
  const unsigned char k3[16] = {8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1};
  const unsigned char* keys2[2] = {k0, k1};
  const unsigned char* keys4[4] = {k0, k1, k2, k3};

  std::srand(42);
  std::vector<int4> s0s(2 * kN), betas(kN), seeds0(kN), seeds1(kN);
  std::vector<In> alphas(kN), xs(kN);
  for (int i = 0; i < kN; ++i) {
    s0s[2 * i] = {std::rand(), std::rand(), std::rand(), std::rand() & ~1};
    s0s[2 * i + 1] = {std::rand(), std::rand(), std::rand(), std::rand() & ~1};
    seeds0[i] = s0s[2 * i];
    seeds1[i] = s0s[2 * i + 1];
    alphas[i] = (In)std::rand();
    betas[i] = {std::rand(), std::rand(), 0, 0};
    xs[i] = (i % 2) ? alphas[i] : (In)std::rand();  // half the points hit alpha
  }
  int4 *d_s0s = ToDevice(s0s), *d_betas = ToDevice(betas), *d_seeds0 = ToDevice(seeds0), *d_seeds1 = ToDevice(seeds1);
  In *d_alphas = ToDevice(alphas), *d_xs = ToDevice(xs);
  int4 *d_y0, *d_y1, *d_cw_s, *d_cw_v, *d_out_cw;
  uint64_t* d_tr;
  HIP_OK(hipMalloc(&d_y0, kN * sizeof(int4)));
  HIP_OK(hipMalloc(&d_y1, kN * sizeof(int4)));
  HIP_OK(hipMalloc(&d_cw_s, (size_t)kInBits * kN * sizeof(int4)));
  HIP_OK(hipMalloc(&d_cw_v, (size_t)kInBits * kN * sizeof(int4)));
  HIP_OK(hipMalloc(&d_out_cw, kN * sizeof(int4)));
  HIP_OK(hipMalloc(&d_tr, kN * sizeof(uint64_t)));
  std::vector<int4> y0(kN), y1(kN);
  int errors = 0;

  try {
    {  // ---- DPF: y0 + y1 == beta at alpha, 0 elsewhere
      DpfT dpf{DpfPrg(keys2)};
      DpfT::Cw* d_cws;
      HIP_OK(hipMalloc(&d_cws, (size_t)kN * DpfT::kSlots * sizeof(DpfT::Cw)));
      dpf.Gen(d_cws, d_s0s, d_alphas, d_betas, kN);
      fss::gpu::DpfRelayoutGpu(d_cws, kN, d_cw_s, d_tr, d_out_cw, dpf);
      fss::gpu::DpfEvalPointGpu(false, d_seeds0, d_cw_s, d_tr, d_out_cw, d_xs, d_y0, kN, dpf);
      fss::gpu::DpfEvalPointGpu(true, d_seeds1, d_cw_s, d_tr, d_out_cw, d_xs, d_y1, kN, dpf);
      HIP_OK(hipMemcpy(y0.data(), d_y0, kN * sizeof(int4), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(y1.data(), d_y1, kN * sizeof(int4), hipMemcpyDeviceToHost));
      for (int i = 0; i < kN; ++i) {
        uint64_t want = xs[i] == alphas[i] ? Val(betas[i]) : 0;
        if (Val(y0[i]) + Val(y1[i]) != want) ++errors;
      }
      // one key, all points through the shared-key entry point must agree with the per-key launcher
      dpf.EvalPoints(false, d_seeds0, d_cws, d_xs, d_y1, 1);
      int4 a;
      HIP_OK(hipMemcpy(&a, d_y1, sizeof(int4), hipMemcpyDeviceToHost));
      if (Val(a) != Val(y0[0])) ++errors;
      std::printf("DPF  n=%d: %d keys, %d mismatches\n", kInBits, kN, errors);
      HIP_OK(hipFree(d_cws));
    }
    {  // ---- DCF (x < alpha): y0 + y1 == beta below alpha, 0 otherwise
      int before = errors;
      DcfT dcf{DcfPrg(keys4)};
      DcfT::Cw* d_cws;
      HIP_OK(hipMalloc(&d_cws, (size_t)kN * DcfT::kSlots * sizeof(DcfT::Cw)));
      dcf.Gen(d_cws, d_s0s, d_alphas, d_betas, kN);
      fss::gpu::DcfRelayoutGpu(d_cws, kN, d_cw_s, d_cw_v, d_out_cw, dcf);
      fss::gpu::DcfEvalPointGpu(false, d_seeds0, d_cw_s, d_cw_v, d_out_cw, d_xs, d_y0, kN, dcf);
      fss::gpu::DcfEvalPointGpu(true, d_seeds1, d_cw_s, d_cw_v, d_out_cw, d_xs, d_y1, kN, dcf);
      HIP_OK(hipMemcpy(y0.data(), d_y0, kN * sizeof(int4), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(y1.data(), d_y1, kN * sizeof(int4), hipMemcpyDeviceToHost));
      for (int i = 0; i < kN; ++i) {
        uint64_t want = xs[i] < alphas[i] ? Val(betas[i]) : 0;
        if (Val(y0[i]) + Val(y1[i]) != want) ++errors;
      }
      std::printf("DCF  n=%d: %d keys, %d mismatches\n", kInBits, kN, errors - before);
      HIP_OK(hipFree(d_cws));
    }
  } catch (const fss::Error& e) {
    std::fprintf(stderr, "fss error: %s\n", e.what());
    return 3;
  }
  std::printf(errors ? "FAILED\n" : "OK\n");
  return errors ? 1 : 0;
}
