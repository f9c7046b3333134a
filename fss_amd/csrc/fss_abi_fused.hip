// libfss_amd.so, fourth unit: full-domain evaluation of one key as ONE launch (evalall_fused_kernel, fss_evalall.hpp).
#include "fss_fused_launch.hpp"

int launch_fused_evalall(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* seed, const uint32_t* cws,
                         const uint32_t* ocw, const fssamd::FusedArgs& f, int grid, hipStream_t st);

// leaf_t: depth of the leaf stage.
int launch_fused_evalall(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* seed, const uint32_t* cws,
                         const uint32_t* ocw, const fssamd::FusedArgs& f, int grid, hipStream_t st) {
  SharedKey k;
  k.seed = seed; k.cws = cws; k.ocw = ocw;
  const bool narrow = fused_narrow(f);
  if (scheme == S_DPF) return narrow ? by_leaf<S_DPF, 2>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_DPF, 4>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_HT) return narrow ? by_leaf<S_HT, 2>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_HT, 4>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_GROTTO) return by_leaf<S_GROTTO, 4>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_DCF) return by_leaf<S_DCF, 4>(ctx, leaf_t, k, f, grid, st);
  return FSS_AMD_EINVAL;
}
