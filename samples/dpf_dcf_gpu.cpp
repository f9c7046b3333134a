// Batched DPF and DCF on the GPU through the C++ layer (include/fss_amd.hpp).
// Same flow as the reference's samples/dpf_dcf_gpu.cu:84-236 -- one key per lane: Gen on the
// device, evaluate both parties, reconstruct on the host and self-check -- with the AES-128-MMO
// PRG instead of ChaCha, plus the level-major relayout + EvalPointGpu launchers.  Then the flows of
// samples/half_tree_dpf_cpu.cu (Half-Tree, here n = 64 with the ChaCha PRG and Gen straight into the
// level-major layout) and samples/grotto_dcf_cpu.cu (Gen, Preprocess + Eval, EvalAll, reconstruct).
//
// build: hipcc --offload-arch=gfx950 -O2 -std=c++20 -Iinclude samples/dpf_dcf_gpu.cpp \
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
      dpf.GenBatch(d_cws, d_s0s, d_alphas, d_betas, kN);
      fss::gpu::DpfRelayoutGpu<kInBits, Group, DpfPrg, In>(d_cws, kN, d_cw_s, d_tr, d_out_cw);
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
      dcf.GenBatch(d_cws, d_s0s, d_alphas, d_betas, kN);
      fss::gpu::DcfRelayoutGpu<kInBits, Group, DcfPrg, In>(d_cws, kN, d_cw_s, d_cw_v, d_out_cw);
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
    {  // ---- Half-Tree DPF, n = 64, ChaCha PRG, keys generated level-major (no relayout)
      int before = errors;
      using HtPrg = fss::prg::ChaCha<1>;
      using HtT = fss::HalfTreeDpf<64, Group, HtPrg, uint64_t>;
      const int nonce[2] = {0x12345678, (int)0x9abcdef0};
      const int4 hash_key = {0x12345678, (int)0x9abcdef0, 0x13572468, 0x2468ace0};
      HtT ht{HtPrg(nonce), hash_key};
      std::vector<uint64_t> a64(kN), x64(kN);
      for (int i = 0; i < kN; ++i) {
        a64[i] = ((uint64_t)std::rand() << 40) ^ ((uint64_t)std::rand() << 20) ^ (uint64_t)std::rand();
        x64[i] = (i % 2) ? a64[i] : a64[i] ^ (1ull << (i % 64));
      }
      uint64_t *d_a64 = ToDevice(a64), *d_x64 = ToDevice(x64);
      int4 *d_ht_s, *d_ocws;
      uint32_t* d_extra;
      HIP_OK(hipMalloc(&d_ht_s, (size_t)64 * kN * sizeof(int4)));
      HIP_OK(hipMalloc(&d_ocws, kN * sizeof(int4)));
      HIP_OK(hipMalloc(&d_extra, kN * sizeof(uint32_t)));
      ht.GenLevelMajor(d_ht_s, d_extra, d_ocws, d_s0s, d_a64, d_betas, kN);
      fss::gpu::HalfTreeDpfEvalPointGpu(false, d_seeds0, d_ht_s, d_extra, d_ocws, d_x64, d_y0, kN, ht);
      fss::gpu::HalfTreeDpfEvalPointGpu(true, d_seeds1, d_ht_s, d_extra, d_ocws, d_x64, d_y1, kN, ht);
      HIP_OK(hipMemcpy(y0.data(), d_y0, kN * sizeof(int4), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(y1.data(), d_y1, kN * sizeof(int4), hipMemcpyDeviceToHost));
      for (int i = 0; i < kN; ++i) {
        uint64_t want = x64[i] == a64[i] ? Val(betas[i]) : 0;
        if (Val(y0[i]) + Val(y1[i]) != want) ++errors;
      }
      std::printf("HalfTree n=64 (ChaCha): %d keys, %d mismatches\n", kN, errors - before);
      HIP_OK(hipFree(d_ht_s)); HIP_OK(hipFree(d_ocws)); HIP_OK(hipFree(d_extra)); HIP_OK(hipFree(d_a64)); HIP_OK(hipFree(d_x64));
    }
    {  // ---- Grotto DCF, n = 16: share0 ^ share1 == (alpha <= x), by table and by EvalAll
      int before = errors;
      constexpr int kG = 16;
      using GrT = fss::GrottoDcf<kG, DpfPrg, In>;
      GrT gr{DpfPrg(keys2)};
      const In alpha = 12345;
      std::vector<In> ga(1, alpha), gx(kN);
      for (int i = 0; i < kN; ++i) gx[i] = (In)(std::rand() & ((1 << kG) - 1));
      gx[0] = alpha; gx[1] = alpha - 1; gx[2] = 0; gx[3] = (1 << kG) - 1;
      In *d_ga = ToDevice(ga), *d_gx = ToDevice(gx);
      GrT::Cw* d_cws;
      uint8_t *d_pt[2], *d_b[2], *d_all[2];
      void* d_ws;
      const size_t ws_bytes = GrT::PreprocessWorkspace();
      HIP_OK(hipMalloc(&d_cws, GrT::kSlots * sizeof(GrT::Cw)));
      HIP_OK(hipMalloc(&d_ws, ws_bytes));
      gr.GenBatch(d_cws, d_s0s, d_ga, 1);
      std::vector<uint8_t> b[2], all[2];
      for (int p = 0; p < 2; ++p) {
        HIP_OK(hipMalloc(&d_pt[p], GrT::ParityTreeBytes()));
        HIP_OK(hipMalloc(&d_b[p], kN));
        HIP_OK(hipMalloc(&d_all[p], size_t(1) << kG));
        gr.PreprocessDevice(p, d_pt[p], d_s0s + p, d_cws, d_ws, ws_bytes);
        gr.EvalPoints(d_pt[p], d_gx, d_b[p], kN);
        gr.EvalAllBlock(p, d_s0s + p, d_cws, d_all[p], d_ws, ws_bytes);
        b[p].resize(kN); all[p].resize(size_t(1) << kG);
        HIP_OK(hipMemcpy(b[p].data(), d_b[p], kN, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(all[p].data(), d_all[p], size_t(1) << kG, hipMemcpyDeviceToHost));
      }
      for (int i = 0; i < kN; ++i) {
        if ((b[0][i] ^ b[1][i]) != (alpha <= gx[i])) ++errors;
        if (b[0][i] != all[0][gx[i]] || b[1][i] != all[1][gx[i]]) ++errors;  // Preprocess + Eval == EvalAll per party
      }
      for (In x = 0; x < (1u << kG); ++x)
        if ((all[0][x] ^ all[1][x]) != (alpha <= x)) ++errors;
      std::printf("Grotto n=%d: %d points + full domain, %d mismatches\n", kG, kN, errors - before);
      for (int p = 0; p < 2; ++p) { HIP_OK(hipFree(d_pt[p])); HIP_OK(hipFree(d_b[p])); HIP_OK(hipFree(d_all[p])); }
      HIP_OK(hipFree(d_cws)); HIP_OK(hipFree(d_ws)); HIP_OK(hipFree(d_ga)); HIP_OK(hipFree(d_gx));
    }
    {  // ---- VDPF (ChaCha + BLAKE3): shares reconstruct, the two parties' proofs agree, a forged key is caught
      int before = errors;
      using VPrg = fss::prg::ChaCha<2>;
      using VT = fss::Vdpf<kInBits, Group, VPrg, fss::hash::Blake3, fss::hash::Blake3, In>;
      const int nonce[2] = {0x12345678, (int)0x9abcdef0};
      const int4 iv[2] = {{0x11223344, 0x55667788, (int)0x99aabbccu, (int)0xddeeff00u},
                          {0x00112233, 0x44556677, (int)0x8899aabbu, (int)0xccddeeffu}};
      VT vdpf{VPrg(nonce), fss::hash::Blake3(iv), fss::hash::Blake3(iv)};
      int4 *d_cs, *d_ocws, *d_pi0, *d_pi1, *d_pr0, *d_pr1;
      HIP_OK(hipMalloc(&d_cs, 4 * kN * sizeof(int4)));
      HIP_OK(hipMalloc(&d_ocws, kN * sizeof(int4)));
      HIP_OK(hipMalloc(&d_pi0, 4 * kN * sizeof(int4)));
      HIP_OK(hipMalloc(&d_pi1, 4 * kN * sizeof(int4)));
      HIP_OK(hipMalloc(&d_pr0, 4 * kN * sizeof(int4)));
      HIP_OK(hipMalloc(&d_pr1, 4 * kN * sizeof(int4)));
      vdpf.GenLevelMajor(d_cw_s, d_tr, d_cs, d_ocws, d_s0s, d_alphas, d_betas, kN);
      fss::gpu::VdpfEvalPointGpu(false, d_seeds0, d_cw_s, d_tr, d_cs, d_ocws, d_xs, d_y0, d_pi0, kN, vdpf);
      fss::gpu::VdpfEvalPointGpu(true, d_seeds1, d_cw_s, d_tr, d_cs, d_ocws, d_xs, d_y1, d_pi1, kN, vdpf);
      vdpf.ProveKeys(d_pi0, d_cs, kN, d_pr0);
      vdpf.ProveKeys(d_pi1, d_cs, kN, d_pr1);
      std::vector<int4> pr0(4 * kN), pr1(4 * kN);
      HIP_OK(hipMemcpy(y0.data(), d_y0, kN * sizeof(int4), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(y1.data(), d_y1, kN * sizeof(int4), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(pr0.data(), d_pr0, 4 * kN * sizeof(int4), hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(pr1.data(), d_pr1, 4 * kN * sizeof(int4), hipMemcpyDeviceToHost));
      for (int i = 0; i < kN; ++i) {
        uint64_t want = xs[i] == alphas[i] ? Val(betas[i]) : 0;
        if (Val(y0[i]) + Val(y1[i]) != want) ++errors;
        if (!VT::Verify(&pr0[4 * i], &pr1[4 * i])) ++errors;
      }
      // party 1 is handed a key with one flipped seed-correction bit at level 3: every proof must differ
      int4 forged;
      HIP_OK(hipMemcpy(&forged, d_cw_s + (size_t)3 * kN, sizeof(int4), hipMemcpyDeviceToHost));
      forged.y ^= 0x100;
      HIP_OK(hipMemcpy(d_cw_s + (size_t)3 * kN, &forged, sizeof(int4), hipMemcpyHostToDevice));
      fss::gpu::VdpfEvalPointGpu(true, d_seeds1, d_cw_s, d_tr, d_cs, d_ocws, d_xs, d_y1, d_pi1, 1, vdpf);
      vdpf.ProveKeys(d_pi1, d_cs, 1, d_pr1);
      HIP_OK(hipMemcpy(pr1.data(), d_pr1, 4 * sizeof(int4), hipMemcpyDeviceToHost));
      if (VT::Verify(&pr0[0], &pr1[0])) ++errors;
      std::printf("VDPF n=%d (ChaCha+BLAKE3): %d keys, %d mismatches\n", kInBits, kN, errors - before);
      HIP_OK(hipFree(d_cs)); HIP_OK(hipFree(d_ocws)); HIP_OK(hipFree(d_pi0)); HIP_OK(hipFree(d_pi1));
      HIP_OK(hipFree(d_pr0)); HIP_OK(hipFree(d_pr1));
    }
  } catch (const fss::Error& e) {
    std::fprintf(stderr, "fss error: %s\n", e.what());
    return 3;
  }
  std::printf(errors ? "FAILED\n" : "OK\n");
  return errors ? 1 : 0;
}
