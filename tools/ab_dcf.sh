#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
run() { echo -n "[$1] "; env $1 python3 tools/bench_evalall.py --scheme dcf --n $2 --log2-leaves $3 --steady 2>/dev/null | tail -1; }
for rep in 1 2; do
for cfg in "24 24" "26 26" "28 25" "20 20"; do
  set -- $cfg
  for e in "FSS_AMD_FUSED=0" "FSS_AMD_FUSED=1" "FSS_AMD_MID_E=1" "FSS_AMD_MID_E=2"; do run "$e" $1 $2; done
done
done
