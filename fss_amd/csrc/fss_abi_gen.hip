// libfss_amd.so, third unit: key generation on the device (dealer side) and the verifiable DPF.
#include "fss_host.hpp"
#include "fss_gen.hpp"
#include "fss_vdpf.hpp"

// ------------------------------------------------------------------ key generation

struct LevelMajorOut {
  void *cw_s = nullptr, *cw_v = nullptr, *tr_bits = nullptr, *out_cw = nullptr, *extra = nullptr;
};

template <bool LM>
static int gen_common(fss_amd_ctx* ctx, int scheme, int in_bits, int group, int pred, const void* s0s,
                      const void* alphas, int x_bytes, const void* betas, int64_t nkeys, void* cws, void* ocws,
                      const LevelMajorOut& lm, void* stream) {
  if (!ctx || nkeys < 0 || !group_ok(ctx, group) || !x_ok(in_bits, x_bytes) || (pred != 0 && pred != 1)) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!s0s || !alphas || (scheme == S_HT && !ocws)) return FSS_AMD_EINVAL;  // betas may be null: beta = 0
  if (!LM && !cws) return FSS_AMD_EINVAL;
  if (LM && (!lm.cw_s || (scheme == S_DPF && (!lm.tr_bits || !lm.out_cw)) ||
             (scheme == S_DCF && (!lm.cw_v || !lm.out_cw)) || (scheme == S_HT && !lm.extra)))
    return FSS_AMD_EINVAL;
  if (!aligned16(s0s, betas, cws, ocws, lm.cw_s, lm.cw_v, lm.out_cw) || !aligned_to(alphas, x_bytes) ||
      !aligned_to(lm.tr_bits, 8) || !aligned_to(lm.extra, 4))
    return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  GenArgs a{};
  a.in_bits = in_bits; a.group = group; a.pred = pred; a.x_bytes = x_bytes; a.nkeys = nkeys;
  a.s0s = reinterpret_cast<const uint4*>(s0s);
  a.alphas = alphas;
  a.betas = reinterpret_cast<const uint4*>(betas);
  a.cws = reinterpret_cast<uint4*>(cws);
  a.ocws = reinterpret_cast<uint4*>(ocws);
  a.cw_s = reinterpret_cast<uint4*>(lm.cw_s);
  a.cw_v = reinterpret_cast<uint4*>(lm.cw_v);
  a.tr_bits = reinterpret_cast<uint64_t*>(lm.tr_bits);
  a.out_cw = reinterpret_cast<uint4*>(lm.out_cw);
  a.extra = reinterpret_cast<uint32_t*>(lm.extra);
  int bs = nkeys <= 64 ? 64 : (nkeys <= 256 ? 256 : 1024);
  int64_t want = (nkeys + bs - 1) / bs;
  int64_t cap = (int64_t)ctx->num_cu;
  int grid = (int)(want < cap ? want : cap);
  a.dyn = use_dyn(nkeys, 64, grid, bs);
  if (scheme == S_DPF)
    FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((gen_kernel<LT, S_DPF, LM>), dim3(grid), dim3(bs), 0, S(stream), ctx->host, ctx->te0, a));
  else if (scheme == S_DCF)
    FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((gen_kernel<LT, S_DCF, LM>), dim3(grid), dim3(bs), 0, S(stream), ctx->host, ctx->te0, a));
  else
    FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((ht_gen_kernel<LT, LM>), dim3(grid), dim3(bs), 0, S(stream), ctx->host, ctx->te0, a));
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = scheme == S_DPF ? "dpf_gen" : (scheme == S_DCF ? "dcf_gen" : "ht_gen");
  ctx->last_grid = grid;
  ctx->last_block = bs;
  return FSS_AMD_OK;
}

extern "C" {

int fss_amd_dpf_gen(fss_amd_ctx* ctx, int in_bits, int group, const void* s0s, const void* alphas, int x_bytes,
                    const void* betas, int64_t nkeys, void* cws, void* stream) {
  return gen_common<false>(ctx, S_DPF, in_bits, group, 0, s0s, alphas, x_bytes, betas, nkeys, cws, nullptr, {}, stream);
}
int fss_amd_dcf_gen(fss_amd_ctx* ctx, int in_bits, int group, int pred, const void* s0s, const void* alphas,
                    int x_bytes, const void* betas, int64_t nkeys, void* cws, void* stream) {
  return gen_common<false>(ctx, S_DCF, in_bits, group, pred, s0s, alphas, x_bytes, betas, nkeys, cws, nullptr, {}, stream);
}
int fss_amd_ht_gen(fss_amd_ctx* ctx, int in_bits, int group, const void* s0s, const void* alphas, int x_bytes,
                   const void* betas, int64_t nkeys, void* cws, void* ocws, void* stream) {
  return gen_common<false>(ctx, S_HT, in_bits, group, 0, s0s, alphas, x_bytes, betas, nkeys, cws, ocws, {}, stream);
}

int fss_amd_dpf_gen_level_major(fss_amd_ctx* ctx, int in_bits, int group, const void* s0s, const void* alphas,
                                int x_bytes, const void* betas, int64_t nkeys, void* cw_s, void* tr_bits,
                                void* out_cw, void* stream) {
  LevelMajorOut lm;
  lm.cw_s = cw_s; lm.tr_bits = tr_bits; lm.out_cw = out_cw;
  return gen_common<true>(ctx, S_DPF, in_bits, group, 0, s0s, alphas, x_bytes, betas, nkeys, nullptr, nullptr, lm, stream);
}
int fss_amd_dcf_gen_level_major(fss_amd_ctx* ctx, int in_bits, int group, int pred, const void* s0s,
                                const void* alphas, int x_bytes, const void* betas, int64_t nkeys, void* cw_s,
                                void* cw_v, void* out_cw, void* stream) {
  LevelMajorOut lm;
  lm.cw_s = cw_s; lm.cw_v = cw_v; lm.out_cw = out_cw;
  return gen_common<true>(ctx, S_DCF, in_bits, group, pred, s0s, alphas, x_bytes, betas, nkeys, nullptr, nullptr, lm, stream);
}
int fss_amd_ht_gen_level_major(fss_amd_ctx* ctx, int in_bits, int group, const void* s0s, const void* alphas,
                               int x_bytes, const void* betas, int64_t nkeys, void* cw_s, void* extra, void* ocws,
                               void* stream) {
  LevelMajorOut lm;
  lm.cw_s = cw_s; lm.extra = extra;
  return gen_common<true>(ctx, S_HT, in_bits, group, 0, s0s, alphas, x_bytes, betas, nkeys, nullptr, ocws, lm, stream);
}

}  // extern "C"

// ------------------------------------------------------------------ verifiable DPF (include/fss/vdpf.cuh)
extern "C" {

int fss_amd_ctx_set_hash_iv(fss_amd_ctx* ctx, const int32_t iv[8]) {
  if (!ctx || !iv) return FSS_AMD_EINVAL;
  for (int i = 0; i < 8; ++i) ctx->host.blake3_iv[i] = (uint32_t)iv[i];
  ctx->host.hash_kind = 0;
  return FSS_AMD_OK;
}

int fss_amd_ctx_set_hash_sha256(fss_amd_ctx* ctx, const int32_t key[4]) {
  if (!ctx || !key) return FSS_AMD_EINVAL;
  for (int i = 0; i < 8; ++i) ctx->host.blake3_iv[i] = i < 4 ? (uint32_t)key[i] : 0u;
  ctx->host.hash_kind = 1;
  return FSS_AMD_OK;
}

}  // extern "C"

template <typename K>
static int vdpf_launch_k(fss_amd_ctx* ctx, K kernel, const char* name, int ilp, const uint32_t* seed, const uint32_t* cws,
                         const uint32_t* cs, const uint32_t* ocw, VdpfArgs a, hipStream_t st) {
  int grid = 0;
  int rc = persistent_grid(ctx, kernel, 1024, (a.n + ilp - 1) / ilp, &grid);
  if (rc != FSS_AMD_OK) return rc;
  a.dyn = use_dyn(a.n, 64 * ilp, grid, 1024);
  return launch_grid(ctx, kernel, name, grid, 1024, st, ctx->host, ctx->te0, seed, cws, cs, ocw, a);
}

template <bool SHARED>
static int vdpf_launch(fss_amd_ctx* ctx, const char* name, const uint32_t* seed, const uint32_t* cws, const uint32_t* cs,
                       const uint32_t* ocw, const VdpfArgs& a, hipStream_t st) {
  if (a.n == 0) return FSS_AMD_OK;
  if (tune().ilp == 1)
    return FSS_BY_PRG(ctx, (vdpf_launch_k(ctx, vdpf_points_kernel<LT, SHARED, 1, 1024>, name, 1, seed, cws, cs, ocw, a, st)));
  return FSS_BY_PRG(ctx, (vdpf_launch_k(ctx, vdpf_points_kernel<LT, SHARED, 2, 1024>, name, 2, seed, cws, cs, ocw, a, st)));
}

template <bool LM>
static int vdpf_gen_common(fss_amd_ctx* ctx, int in_bits, int group, const void* s0s, const void* alphas, int x_bytes,
                           const void* betas, int64_t nkeys, void* cws, void* cw_s, void* tr_bits, void* cs, void* ocws,
                           int32_t* retry, void* stream) {
  if (!ctx || nkeys < 0 || !group_ok(ctx, group) || !x_ok(in_bits, x_bytes)) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!s0s || !alphas || !cs || !ocws) return FSS_AMD_EINVAL;
  if (LM ? (!cw_s || !tr_bits) : !cws) return FSS_AMD_EINVAL;
  if (!aligned16(s0s, betas, cws, cw_s, cs, ocws) || !aligned_to(alphas, x_bytes) || !aligned_to(tr_bits, 8) ||
      !aligned_to(retry, 4))
    return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  VdpfGenArgs a{};
  a.in_bits = in_bits; a.group = group; a.x_bytes = x_bytes; a.nkeys = nkeys;
  a.s0s = reinterpret_cast<const uint4*>(s0s);
  a.alphas = alphas;
  a.betas = reinterpret_cast<const uint4*>(betas);
  a.cws = reinterpret_cast<uint4*>(cws);
  a.cw_s = reinterpret_cast<uint4*>(cw_s);
  a.tr_bits = reinterpret_cast<uint64_t*>(tr_bits);
  a.cs = reinterpret_cast<uint4*>(cs);
  a.ocws = reinterpret_cast<uint4*>(ocws);
  a.retry = retry;
  int bs = nkeys <= 64 ? 64 : (nkeys <= 256 ? 256 : 1024);
  int64_t want = (nkeys + bs - 1) / bs;
  int grid = (int)(want < (int64_t)ctx->num_cu ? want : (int64_t)ctx->num_cu);
  a.dyn = use_dyn(nkeys, 64, grid, bs);
  FSS_BY_PRG_DO(ctx, hipLaunchKernelGGL((vdpf_gen_kernel<LT, LM>), dim3(grid), dim3(bs), 0, S(stream), ctx->host, ctx->te0, a));
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "vdpf_gen";
  ctx->last_grid = grid;
  ctx->last_block = bs;
  return FSS_AMD_OK;
}

extern "C" {

int fss_amd_vdpf_gen(fss_amd_ctx* ctx, int in_bits, int group, const void* s0s, const void* alphas, int x_bytes,
                     const void* betas, int64_t nkeys, void* cws, void* cs, void* ocws, int32_t* retry, void* stream) {
  return vdpf_gen_common<false>(ctx, in_bits, group, s0s, alphas, x_bytes, betas, nkeys, cws, nullptr, nullptr, cs, ocws,
                                retry, stream);
}
int fss_amd_vdpf_gen_level_major(fss_amd_ctx* ctx, int in_bits, int group, const void* s0s, const void* alphas,
                                 int x_bytes, const void* betas, int64_t nkeys, void* cw_s, void* tr_bits, void* cs,
                                 void* ocws, int32_t* retry, void* stream) {
  return vdpf_gen_common<true>(ctx, in_bits, group, s0s, alphas, x_bytes, betas, nkeys, nullptr, cw_s, tr_bits, cs, ocws,
                               retry, stream);
}

static int vdpf_relayout_common(fss_amd_ctx* ctx, int in_bits, const void* cws, int64_t nkeys, void* cw_s, void* tr_bits,
                                int tr32, void* stream) {
  if (!ctx || nkeys < 0 || in_bits < 1 || in_bits > (tr32 ? 32 : 128)) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!cws || !cw_s || !tr_bits) return FSS_AMD_EINVAL;
  if (!aligned16(cws, cw_s) || !aligned_to(tr_bits, tr32 ? 4 : 8)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  launch_relayout<S_DPF>(ctx, S(stream), in_bits,
                     in_bits, reinterpret_cast<const uint4*>(cws), nkeys, reinterpret_cast<uint4*>(cw_s), (uint4*)nullptr,
                     reinterpret_cast<uint64_t*>(tr_bits), (uint4*)nullptr, (uint32_t*)nullptr, tr32);
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "relayout";
  ctx->last_grid = (int)((nkeys + 63) / 64);
  ctx->last_block = 256;
  return FSS_AMD_OK;
}
int fss_amd_vdpf_relayout(fss_amd_ctx* ctx, int in_bits, const void* cws, int64_t nkeys, void* cw_s, void* tr_bits,
                          void* stream) {
  return vdpf_relayout_common(ctx, in_bits, cws, nkeys, cw_s, tr_bits, 0, stream);
}
int fss_amd_vdpf_relayout_u32(fss_amd_ctx* ctx, int in_bits, const void* cws, int64_t nkeys, void* cw_s, void* extra,
                              void* stream) {
  return vdpf_relayout_common(ctx, in_bits, cws, nkeys, cw_s, extra, 1, stream);
}

int fss_amd_vdpf_eval_points(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seed, const void* cws,
                             const void* cs, const void* ocw, const void* xs, int x_bytes, int64_t npoints, void* ys,
                             void* pi_tildes, void* stream) {
  if (!ctx || !seed || !cws || !cs || !ocw || npoints < 0 || !group_ok(ctx, group) || (party != 0 && party != 1) ||
      !x_ok(in_bits, x_bytes))
    return FSS_AMD_EINVAL;
  if (npoints > 0 && (!xs || !ys || !pi_tildes)) return FSS_AMD_EINVAL;
  if (!aligned16(seed, cws, cs, ocw, ys, pi_tildes) || !aligned_to(xs, x_bytes)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  VdpfArgs a{};
  a.in_bits = in_bits; a.party = party; a.group = group; a.x_bytes = x_bytes; a.n = npoints;
  a.xs = xs; a.ys = reinterpret_cast<uint4*>(ys); a.pis = reinterpret_cast<uint4*>(pi_tildes);
  return vdpf_launch<true>(ctx, "vdpf_eval_points", reinterpret_cast<const uint32_t*>(seed),
                           reinterpret_cast<const uint32_t*>(cws), reinterpret_cast<const uint32_t*>(cs),
                           reinterpret_cast<const uint32_t*>(ocw), a, S(stream));
}

static int vdpf_eval_keys_common(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cw_s,
                                 const void* tr_bits, const void* cs, const void* ocws, const void* xs, int x_bytes,
                                 int64_t nkeys, void* ys, void* pi_tildes, int tr32, void* stream) {
  if (!ctx || nkeys < 0 || !group_ok(ctx, group) || (party != 0 && party != 1) || !x_ok(in_bits, x_bytes)) return FSS_AMD_EINVAL;
  if (tr32 && in_bits > 32) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!seeds || !cw_s || !tr_bits || !cs || !ocws || !xs || !ys || !pi_tildes) return FSS_AMD_EINVAL;
  if (!aligned16(seeds, cw_s, cs, ocws, ys, pi_tildes) || !aligned_to(tr_bits, tr32 ? 4 : 8) || !aligned_to(xs, x_bytes))
    return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  VdpfArgs a{};
  a.in_bits = in_bits; a.party = party; a.group = group; a.x_bytes = x_bytes; a.n = nkeys;
  a.xs = xs; a.ys = reinterpret_cast<uint4*>(ys); a.pis = reinterpret_cast<uint4*>(pi_tildes);
  a.seeds = reinterpret_cast<const uint4*>(seeds);
  a.cw_s = reinterpret_cast<const uint4*>(cw_s);
  a.tr_bits = reinterpret_cast<const uint64_t*>(tr_bits);
  a.tr32 = tr32;
  a.cs = reinterpret_cast<const uint4*>(cs);
  a.ocws = reinterpret_cast<const uint4*>(ocws);
  return vdpf_launch<false>(ctx, "vdpf_eval_keys", nullptr, nullptr, nullptr, nullptr, a, S(stream));
}
int fss_amd_vdpf_eval_keys(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cw_s,
                           const void* tr_bits, const void* cs, const void* ocws, const void* xs, int x_bytes,
                           int64_t nkeys, void* ys, void* pi_tildes, void* stream) {
  return vdpf_eval_keys_common(ctx, in_bits, group, party, seeds, cw_s, tr_bits, cs, ocws, xs, x_bytes, nkeys, ys, pi_tildes, 0, stream);
}
int fss_amd_vdpf_eval_keys_u32(fss_amd_ctx* ctx, int in_bits, int group, int party, const void* seeds, const void* cw_s,
                               const void* extra, const void* cs, const void* ocws, const void* xs, int x_bytes,
                               int64_t nkeys, void* ys, void* pi_tildes, void* stream) {
  return vdpf_eval_keys_common(ctx, in_bits, group, party, seeds, cw_s, extra, cs, ocws, xs, x_bytes, nkeys, ys, pi_tildes, 1, stream);
}

int fss_amd_vdpf_prove(fss_amd_ctx* ctx, const void* pi_tildes, int64_t npoints, const void* cs, void* pi, void* stream) {
  if (!ctx || npoints < 0 || !cs || !pi || (npoints > 0 && !pi_tildes)) return FSS_AMD_EINVAL;
  if (!aligned16(pi_tildes, cs, pi)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  hipLaunchKernelGGL(vdpf_prove_kernel<true>, dim3(1), dim3(64), 0, S(stream), ctx->host,
                     reinterpret_cast<const uint4*>(pi_tildes), reinterpret_cast<const uint4*>(cs), npoints,
                     reinterpret_cast<uint4*>(pi));
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "vdpf_prove";
  ctx->last_grid = 1;
  ctx->last_block = 64;
  return FSS_AMD_OK;
}

int fss_amd_vdpf_prove_keys(fss_amd_ctx* ctx, const void* pi_tildes, const void* cs, int64_t nkeys, void* pis,
                            void* stream) {
  if (!ctx || nkeys < 0) return FSS_AMD_EINVAL;
  if (nkeys == 0) return FSS_AMD_OK;
  if (!pi_tildes || !cs || !pis) return FSS_AMD_EINVAL;
  if (!aligned16(pi_tildes, cs, pis)) return FSS_AMD_EINVAL;
  FSS_ON_DEVICE(ctx);
  int64_t want = (nkeys + 255) / 256, cap = (int64_t)ctx->num_cu * 8;
  hipLaunchKernelGGL(vdpf_prove_kernel<false>, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, S(stream),
                     ctx->host, reinterpret_cast<const uint4*>(pi_tildes), reinterpret_cast<const uint4*>(cs), nkeys,
                     reinterpret_cast<uint4*>(pis));
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "vdpf_prove_keys";
  ctx->last_grid = (int)(want < cap ? want : cap);
  ctx->last_block = 256;
  return FSS_AMD_OK;
}

}  // extern "C"
