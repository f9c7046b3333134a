#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# Interleaved A/B of the EvalAll launch forms on one box: the chain of launches (FSS_AMD_FUSED=0) against the one-launch
# form, and the library without whole-line leaf stores (libfss_amd_nopair.so, -DFSS_LEAF_PAIR_LINES=0) against the product.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
for r in 1 2; do
  EXTRA_ENV="FSS_AMD_FUSED=0" bash tools/ab_evalall.sh main
  EXTRA_ENV="FSS_AMD_FUSED=1" bash tools/ab_evalall.sh main
done > gpurun_out/evalall_fused_ab.txt 2>&1
if [ -f fss_amd/libfss_amd_nopair.so ]; then
  for r in 1 2; do bash tools/ab_evalall.sh nopair main; done > gpurun_out/evalall_pair_lines_ab.txt 2>&1
fi
tail -30 gpurun_out/evalall_fused_ab.txt
