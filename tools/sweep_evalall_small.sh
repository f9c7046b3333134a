#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# Small domains are latency-bound: sweep the planner knobs (frontier depth, leaf-stage depth) for them.
R="${GRAFT_REPO_ROOT:-/root/repo}"
for sch in ${SCHEMES:-dpf}; do
for n in ${NS:-16 18 20 22}; do
for top in ${TOPS:-9 11 13 15}; do for le in ${LES:-0 1 2 default}; do
  echo -n "$sch n=$n TOPMAX=$top LEAF_E=$le: "
  FSS_AMD_TOPMAX=$top FSS_AMD_LEAF_E=$([ $le = default ] && echo -1 || echo $le) python3 $R/tools/bench_evalall.py --scheme $sch --n $n --steady 2>/dev/null | tail -1 | awk '{print $(NF-4), $(NF-3), $(NF-2), $(NF-1)}'
done; done; done; done
