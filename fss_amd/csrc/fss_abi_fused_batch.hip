// libfss_amd.so, fifth unit: fss_amd_{dpf,ht}_eval_all_batch as ONE launch (evalall_fused_kernel<..., BATCH>).
#include "fss_fused_launch.hpp"

int launch_fused_evalall_batch(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* cws, const uint32_t* ocws,
                               const fssamd::FusedArgs& f, int grid, hipStream_t st);

int launch_fused_evalall_batch(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* cws, const uint32_t* ocws,
                               const fssamd::FusedArgs& f, int grid, hipStream_t st) {
  SharedKey k;
  k.cws = cws; k.ocw = ocws;  // the seeds are in f.a.seeds
  const bool narrow = fused_narrow(f);
  // UNI kernels (run_stage): every 64-node chunk of every stage lies inside one key -- at least 64 nodes per key at the
  // first stage's input level (top; deeper stages have more), and every workgroup's share of the walks a whole number
  // of them whose first-stage nodes start on a multiple of 64 (ragged shares, FSS_AMD_FUSED_GRID, do not).
  static const bool uni_on = [] { const char* e = tune_env("FSS_AMD_BATCH_UNI"); return !(e && atoi(e) == 0); }();
  const int64_t walks = f.a.nkeys << (f.top - f.tail);
  const bool uni = uni_on && f.top >= 6 && walks % grid == 0 && (((walks / grid) << f.tail) & 63) == 0;
  if (uni) {
    if (scheme == S_DPF) return narrow ? by_leaf<S_DPF, 2, true, Lds, true>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_DPF, 4, true, Lds, true>(ctx, leaf_t, k, f, grid, st);
    if (scheme == S_HT) return narrow ? by_leaf<S_HT, 2, true, Lds, true>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_HT, 4, true, Lds, true>(ctx, leaf_t, k, f, grid, st);
  }
  if (scheme == S_DPF) return narrow ? by_leaf<S_DPF, 2, true>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_DPF, 4, true>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_HT) return narrow ? by_leaf<S_HT, 2, true>(ctx, leaf_t, k, f, grid, st) : by_leaf<S_HT, 4, true>(ctx, leaf_t, k, f, grid, st);
  return FSS_AMD_EINVAL;
}
