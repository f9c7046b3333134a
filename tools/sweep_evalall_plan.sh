#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# A/B the EvalAll planner knobs (frontier depth, depth of the stages between frontier and leaf stage).
R="${GRAFT_REPO_ROOT:-/root/repo}"
for sch in ${SCHEMES:-dpf ht dcf}; do
for cfg in "24 24" "26 26" "28 25"; do set -- $cfg
for top in ${TOPS:-13 15 16}; do for mid in ${MIDS:-0 1 2}; do
  echo -n "$sch n=$1 leaves=2^$2 TOPMAX=$top MID_E=$mid: "
  FSS_AMD_TOPMAX=$top FSS_AMD_MID_E=$mid python3 $R/tools/bench_evalall.py --scheme $sch --n $1 --log2-leaves $2 --steady | tail -1
done; done; done; done
