// libfss_amd.so, seventh unit: the node table of the shared-prefix point walks (fss_amd_*_eval_points_shared) as ONE
// launch -- evalall_fused_kernel with OW = 0: the tree of the key's top `prefix_bits` levels, its last stage storing
// nodes (DCF: and running values) instead of converted leaves.  AES tables only (ChaCha contexts keep the chain).
#include "fss_fused_launch.hpp"

int launch_fused_nodes(fss_amd_ctx* ctx, int scheme, int last_t, const uint32_t* seed, const uint32_t* cws,
                       const fssamd::FusedArgs& f, int grid, hipStream_t st);

// last_t: depth of the last stage (1 .. DM + 2).
int launch_fused_nodes(fss_amd_ctx* ctx, int scheme, int last_t, const uint32_t* seed, const uint32_t* cws,
                       const fssamd::FusedArgs& f, int grid, hipStream_t st) {
  SharedKey k;
  k.seed = seed; k.cws = cws; k.ocw = nullptr;
  if (last_t < 1) return FSS_AMD_EINVAL;
  int rc = FSS_AMD_EINVAL;
  if (scheme == S_DPF) rc = by_leaf<S_DPF, 0>(ctx, last_t, k, f, grid, st);
  else if (scheme == S_HT) rc = by_leaf<S_HT, 0>(ctx, last_t, k, f, grid, st);
  else if (scheme == S_DCF) rc = by_leaf<S_DCF, 0>(ctx, last_t, k, f, grid, st);
  if (rc == FSS_AMD_OK) ctx->last_kernel = "prefix_nodes_fused";
  return rc;
}
