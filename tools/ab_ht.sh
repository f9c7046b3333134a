#!/bin/bash
# A/B of Half-Tree EvalAll library variants (FSS_AMD_LIB), interleaved: usage tools/ab_ht.sh <variant tags...>
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
TAGS="$*"
for r in 1 2; do
for tag in $TAGS; do
  lib="$R/fss_amd/libfss_amd_$tag.so"; [ "$tag" = main ] && lib="$R/fss_amd/libfss_amd.so"
  for cfg in "ht 28 25" "ht 24 24" "ht 26 26" "ht 20 20"; do
    set -- $cfg
    echo -n "[$tag] "
    FSS_AMD_LIB="$lib" python3 tools/bench_evalall.py --scheme $1 --n $2 --log2-leaves $3 --steady 2>/dev/null | tail -1
  done
done
done
