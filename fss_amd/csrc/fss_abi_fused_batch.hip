// libfss_amd.so, fifth unit: fss_amd_{dpf,ht}_eval_all_batch as ONE launch (evalall_fused_kernel<..., BATCH>).
#include "fss_fused_launch.hpp"

int launch_fused_evalall_batch(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* cws, const uint32_t* ocws,
                               const fssamd::FusedArgs& f, int grid, hipStream_t st);

int launch_fused_evalall_batch(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* cws, const uint32_t* ocws,
                               const fssamd::FusedArgs& f, int grid, hipStream_t st) {
  SharedKey k;
  k.cws = cws; k.ocw = ocws;  // the seeds are in f.a.seeds
  const bool narrow = fused_narrow(f);
  if (scheme == S_DPF) return narrow ? by_leaf<S_DPF, 2, true>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_DPF, 4, true>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_HT) return narrow ? by_leaf<S_HT, 2, true>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_HT, 4, true>(ctx, leaf_t, k, f, grid, st);
  return FSS_AMD_EINVAL;
}
