#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# Launch policy of the point walks below the dynamic-distribution threshold: the round-2 policy (dynamic from 16 384 AES
# blocks per wave, fixed shares with rotating priorities below) against fixed shares with progress-ranked priorities
# up to 65 536 blocks per wave (the default), interleaved.  usage: tools/ab_prio.sh
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
OLD="FSS_AMD_DYN_MIN=16384 FSS_AMD_STATIC_PRIO=1"
pts() { for v in "$OLD" ""; do echo -n "[${v:-default}] "; env $v python3 tools/bench_points.py $1 --steady 2>/dev/null | tail -1; done; }
keys() { for v in "$OLD" ""; do echo -n "[${v:-default}] "; env $v python3 tools/bench_keys.py $1 --steady 2>/dev/null | tail -1; done; }
for l in 18 19 20 21 22 23; do
  pts "--scheme dpf --n 32 --log2-points $l"
  pts "--scheme dcf --n 32 --log2-points $l"
  pts "--scheme ht --n 64 --log2-points $l"
  pts "--scheme dpf --n 16 --log2-points $l"
  keys "--scheme dpf --n 32 --log2-keys $l"
  keys "--scheme ht --n 64 --log2-keys $l"
done
