#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# BASELINE configs[1] (DPF n = 32, one key, 2^20 points): per-slot stamps (diagnostic library) under fixed shares with
# rotating priorities (FSS_AMD_STATIC_PRIO=1) and under the default, fixed shares with progress-ranked priorities.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
OUT="$R/gpurun_out/points_2e20.txt"; : > "$OUT"
for v in "FSS_AMD_STATIC_PRIO=1" "FSS_AMD_STATIC_PRIO=2"; do
  echo "=== $v" >> "$OUT"
  env $v FSS_AMD_LIB="$R/fss_amd/libfss_amd_diag.so" python3 tools/keys_clock.py --mode points --scheme dpf --n 32 --log2 20 2>/dev/null | grep workload >> "$OUT"
done
cat "$OUT"
