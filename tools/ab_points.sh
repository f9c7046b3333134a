#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# Interleaved A/B of launch tunables on a shared-key point batch.
# usage: ROUNDS=2 tools/ab_points.sh "<bench_points args>" "<env variant>" ...   ("-" = defaults)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
ARGS="$1"; shift
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    e="$v"; [ "$v" = "-" ] && e=""
    echo -n "[round $r] [$v] "
    env $e python3 tools/bench_points.py $ARGS --steady 2>/dev/null | tail -1
  done
done
