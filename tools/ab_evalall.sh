#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# A/B of EvalAll library variants (FSS_AMD_LIB): steady-state times of the configs[3] block and whole domains.
# usage: tools/ab_evalall.sh <variant tags...>   ("main" = the product library)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
for tag in "$@"; do
  lib="$R/fss_amd/libfss_amd_$tag.so"; [ "$tag" = main ] && lib="$R/fss_amd/libfss_amd.so"
  for cfg in "dpf 28 25" "ht 28 25" "dpf 24 24" "ht 24 24" "dpf 26 26" "dcf 24 24"; do
    set -- $cfg
    echo -n "[$tag${EXTRA_ENV:+ $EXTRA_ENV}] "
    env $EXTRA_ENV FSS_AMD_LIB="$lib" python3 tools/bench_evalall.py --scheme $1 --n $2 --log2-leaves $3 --steady 2>/dev/null | tail -1
  done
done
