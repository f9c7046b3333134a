// libfss_amd.so, sixth unit: the one-launch EvalAll (evalall_fused_kernel) for contexts set to the ChaCha PRG -- no AES
// tables, no narrow leaf blocks; single keys of every scheme (batches keep the chain of launches, measured faster).
#include "fss_fused_launch.hpp"

int launch_fused_evalall_chacha(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* seed, const uint32_t* cws,
                                const uint32_t* ocw, const fssamd::FusedArgs& f, int grid, hipStream_t st, bool batch);

int launch_fused_evalall_chacha(fss_amd_ctx* ctx, int scheme, int leaf_t, const uint32_t* seed, const uint32_t* cws,
                                const uint32_t* ocw, const fssamd::FusedArgs& f, int grid, hipStream_t st, bool batch) {
  SharedKey k;
  k.seed = seed; k.cws = cws; k.ocw = ocw;
  if (batch) return FSS_AMD_EINVAL;  // ChaCha batches keep the chain of launches (fss_abi_evalall.hip)
  if (scheme == S_DPF) return by_leaf<S_DPF, 4, false, ChaChaL>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_HT) return by_leaf<S_HT, 4, false, ChaChaL>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_GROTTO) return by_leaf<S_GROTTO, 4, false, ChaChaL>(ctx, leaf_t, k, f, grid, st);
  if (scheme == S_DCF) return by_leaf<S_DCF, 4, false, ChaChaL>(ctx, leaf_t, k, f, grid, st);
  return FSS_AMD_EINVAL;
}
