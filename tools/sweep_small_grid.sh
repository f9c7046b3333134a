R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
for rep in 1 2; do
for s in dpf ht; do
for n in 8 10 12 14 16; do
  for g in "" "FSS_AMD_FUSED=0"; do
    echo -n "[${g:-default}] "; env $g python3 tools/bench_evalall.py --scheme $s --n $n --steady 2>&1 | tail -1
  done
done
done
done
