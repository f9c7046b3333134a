#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# Interleaved A/B: ROUNDS rounds over "ENV=... ENV=..." variant strings (use "-" for none) on one workload.
# usage: ROUNDS=3 tools/ab_rounds.sh "<bench_evalall args>" "<variant1 env>" "<variant2 env>" ...
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
ARGS="$1"; shift
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in "$@"; do
    e="$v"; [ "$v" = "-" ] && e=""
    echo -n "[round $r] [$v] "
    env $e python3 tools/bench_evalall.py $ARGS --steady 2>/dev/null | tail -1
  done
done
