// libfss_amd.so, fourth unit: full-domain evaluation as ONE launch (evalall_fused_kernel, fss_evalall.hpp).
#include "fss_host.hpp"
#define FSS_EVALALL_TEMPLATES_ONLY 1
#include "fss_evalall.hpp"

int launch_fused_evalall(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* seed, const uint32_t* cws,
                         const uint32_t* ocw, const fssamd::FusedArgs& f, int grid, hipStream_t st, bool batch);

namespace {

template <int SCHEME, int LE, int OW, bool BATCH>
int launch_one(fss_amd_ctx* ctx, const SharedKey& k, const FusedArgs& f, int grid, hipStream_t st) {
  constexpr int LD = SCHEME == S_DCF ? 1 : 2;
  hipLaunchKernelGGL((evalall_fused_kernel<Lds, SCHEME, LD, LE, OW, BATCH>), dim3(grid), dim3(1024), 0, st, ctx->host, ctx->te0,
                     k.seed, k.cws, k.ocw, f);
  HIP_TRY(hipGetLastError());
  ctx->last_kernel = "evalall_fused";
  ctx->last_grid = grid;
  ctx->last_block = 1024;
  return FSS_AMD_OK;
}

template <int SCHEME, int OW, bool BATCH = false>
int by_leaf(fss_amd_ctx* ctx, int leaf_t, const SharedKey& k, const FusedArgs& f, int grid, hipStream_t st) {
  constexpr int LD = SCHEME == S_DCF ? 1 : 2;
  if (leaf_t == LD) return launch_one<SCHEME, 0, OW, BATCH>(ctx, k, f, grid, st);
  if constexpr (SCHEME != S_HT) {
    if (leaf_t == LD + 1) return launch_one<SCHEME, 1, OW, BATCH>(ctx, k, f, grid, st);
    if (leaf_t == LD + 2) return launch_one<SCHEME, 2, OW, BATCH>(ctx, k, f, grid, st);
  }
  return FSS_AMD_EINVAL;
}

}  // namespace

// leaf_t: depth of the leaf stage (2 levels in registers -- DCF: 1 -- + 0 .. 2 rolled levels).  Narrow output groups (Uint<u8 .. u64>)
// take the leaf-level blocks that compute words 0, 1 and the control bit only (expand_kernel's OW).
int launch_fused_evalall(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* seed, const uint32_t* cws,
                         const uint32_t* ocw, const fssamd::FusedArgs& f, int grid, hipStream_t st, bool batch) {
  SharedKey k;
  k.seed = seed; k.cws = cws; k.ocw = ocw;
  static const bool narrow_on = [] { const char* e = getenv("FSS_AMD_LEAF_NARROW"); return !(e && atoi(e) == 0); }();
  const bool narrow = FSS_T4 && narrow_on && f.a.group >= G_U8 && f.a.group <= G_U64;
  if (batch) {  // fss_amd_{dpf,ht}_eval_all_batch
    if (scheme == S_DPF) return narrow ? by_leaf<S_DPF, 2, true>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_DPF, 4, true>(ctx, leaf_t, k, f, grid, st);
    if (scheme == S_HT) return narrow ? by_leaf<S_HT, 2, true>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_HT, 4, true>(ctx, leaf_t, k, f, grid, st);
    return FSS_AMD_EINVAL;
  }
  if (scheme == S_DPF) return narrow ? by_leaf<S_DPF, 2>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_DPF, 4>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_HT) return narrow ? by_leaf<S_HT, 2>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_HT, 4>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_GROTTO) return by_leaf<S_GROTTO, 4>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_DCF) return by_leaf<S_DCF, 4>(ctx, leaf_t, k, f, grid, st);
  return FSS_AMD_EINVAL;
}
