R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
run() { echo -n "[$1] "; env $1 python3 tools/bench_evalall.py --scheme $2 --n $3 --log2-leaves $4 --steady 2>/dev/null | tail -1; }
for rep in 1 2 3; do
for cfg in "dpf 28 25" "ht 28 25" "dpf 24 24" "ht 24 24" "ht 20 20" "dpf 20 20"; do
  set -- $cfg
  for e in "FSS_AMD_TOPMAX=14" "FSS_AMD_TOPMAX=15" "FSS_AMD_TOPMAX=16"; do run "$e" $1 $2 $3; done
done
done
