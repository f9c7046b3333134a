#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# Small-domain EvalAll: the chain of launches (FSS_AMD_FUSED=0) against the one-launch form, interleaved.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
for rep in 1 2; do
for s in dpf ht dcf; do
for n in 8 12 16 17 18 19 20; do
  for f in 0 1; do
    echo -n "[FUSED=$f] "; FSS_AMD_FUSED=$f python3 tools/bench_evalall.py --scheme $s --n $n --steady 2>&1 | tail -1
  done
done
done
done
