#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# Rolled levels in the leaf stage (FSS_AMD_LEAF_E = 0 / 1 / 2) for Half-Tree and DPF, interleaved.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
for r in 1 2; do
for s in ht dpf; do
for cfg in "28 25" "24 24" "26 26" "22 22"; do
  set -- $cfg
  for e in "FSS_AMD_LEAF_E=0" "FSS_AMD_LEAF_E=1" "FSS_AMD_LEAF_E=2" ""; do
    echo -n "[${e:-default}] "; env $e python3 tools/bench_evalall.py --scheme $s --n $1 --log2-leaves $2 --steady 2>/dev/null | tail -1
  done
done
done
done
for e in "FSS_AMD_LEAF_E=0" "FSS_AMD_LEAF_E=2" ""; do echo "[${e:-default}]"; env $e python3 tools/bench_batch.py --scheme ht --shapes 20x64,14x4096 2>&1 | tail -2; done
