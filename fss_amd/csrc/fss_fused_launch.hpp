// Launch side of evalall_fused_kernel (fss_evalall.hpp), shared by the units that instantiate it (fss_abi_fused.hip:
// one key, AES tables; fss_abi_fused_batch.hip: batches of keys; fss_abi_fused_chacha.hip: the ChaCha PRG) so that
// hipcc compiles them in parallel.
#pragma once
#include "fss_host.hpp"
#define FSS_EVALALL_TEMPLATES_ONLY 1
#include "fss_evalall.hpp"

namespace {

template <int SCHEME, int LD, int LE, int OW, bool BATCH, class LT, bool UNI = false>
int launch_one(fss_amd_ctx* ctx, const SharedKey& k, const FusedArgs& f, int grid, hipStream_t st) {
  hipLaunchKernelGGL((evalall_fused_kernel<LT, SCHEME, LD, LE, OW, BATCH, UNI>), dim3(grid), dim3(1024), 0, st, ctx->host, ctx->te0,
                     k.seed, k.cws, k.ocw, f);
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "evalall_fused";
  ctx->last_grid = grid;
  ctx->last_block = 1024;
  return FSS_AMD_OK;
}

// leaf_t levels in the leaf stage: up to DM in registers (DCF 1, the others 2; Half-Tree at least its conversion
// level), then up to two rolled levels above them.
template <int SCHEME, int OW, bool BATCH = false, class LT = Lds, bool UNI = false>
int by_leaf(fss_amd_ctx* ctx, int leaf_t, const SharedKey& k, const FusedArgs& f, int grid, hipStream_t st) {
  constexpr int DM = SCHEME == S_DCF ? 1 : 2;
  if constexpr (SCHEME != S_HT && OW != 0) {  // (node tables, OW = 0, always end on a stage that expands)
    if (leaf_t == 0) return launch_one<SCHEME, 0, 0, 4, BATCH, LT, UNI>(ctx, k, f, grid, st);  // conversion only: no block to narrow
  }
  if (leaf_t == 1) return launch_one<SCHEME, 1, 0, OW, BATCH, LT, UNI>(ctx, k, f, grid, st);
  if constexpr (DM == 2) {
    if (leaf_t == 2) return launch_one<SCHEME, 2, 0, OW, BATCH, LT, UNI>(ctx, k, f, grid, st);
  }
  if (leaf_t == DM + 1) return launch_one<SCHEME, DM, 1, OW, BATCH, LT, UNI>(ctx, k, f, grid, st);
  if (leaf_t == DM + 2) return launch_one<SCHEME, DM, 2, OW, BATCH, LT, UNI>(ctx, k, f, grid, st);
  // three and four rolled levels: one DPF tree with narrow leaves only (fss_abi_evalall.hip: deep_leaf_levels)
  if constexpr (SCHEME == S_DPF && OW == 2 && !BATCH && std::is_same_v<LT, Lds>) {
    if (leaf_t == DM + 3) return launch_one<SCHEME, DM, 3, OW, BATCH, LT, UNI>(ctx, k, f, grid, st);
    if (leaf_t == DM + 4) return launch_one<SCHEME, DM, 4, OW, BATCH, LT, UNI>(ctx, k, f, grid, st);
  }
  return FSS_AMD_EINVAL;
}

}  // namespace

namespace {
// Narrow output groups (Uint<u8 .. u64>) take the leaf-level blocks that compute words 0, 1 and the control bit only
// (expand_kernel's OW).
inline bool fused_narrow(const FusedArgs& f) {
  static const bool narrow_on = [] { const char* e = tune_env("FSS_AMD_LEAF_NARROW"); return !(e && atoi(e) == 0); }();
  return narrow_on && f.a.group >= G_U8 && f.a.group <= G_U64;
}
}  // namespace
