#!/bin/bash
export FSS_AMD_TUNE=1  # the library reads its A/B tunables only under this switch (fss_dev.hpp: tune_env)
# In-kernel clocks of the shared-key and independent-key walks and the launch variants that trade occupancy / ILP
# for power.  Output: gpurun_out/keys_clock.txt (copy to profiles/r2/keys_clock.txt).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
OUT="$R/gpurun_out/keys_clock.txt"
mkdir -p "$R/gpurun_out"; : > "$OUT"
DIAG="$R/fss_amd/libfss_amd_diag.so"
run() {  # run <env assignments...> -- <keys_clock.py args>
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" FSS_AMD_LIB="$DIAG" timeout 300 python3 tools/keys_clock.py "$@" 2>>"$OUT.err" | tee -a "$OUT"
}
echo "# diagnostic library (FSS_DIAG_STAMPS=1): clock = d(s_memtime)/d(s_memrealtime) x 100 MHz, median over workgroups, after >= 2 s of launches" >> "$OUT"
run -- --mode points --scheme dpf --n 32
run -- --mode keys --scheme dpf --n 32
run FSS_AMD_ILP=1 -- --mode keys --scheme dpf --n 32
run FSS_AMD_BS=512 -- --mode keys --scheme dpf --n 32
run FSS_AMD_BS=512 FSS_AMD_ILP=1 -- --mode keys --scheme dpf --n 32
run FSS_AMD_ILP=1 -- --mode points --scheme dpf --n 32
run -- --mode points --scheme dcf --n 32
run -- --mode keys --scheme dcf --n 32
run -- --mode points --scheme ht --n 64
run -- --mode keys --scheme ht --n 64
run FSS_AMD_ILP=1 -- --mode keys --scheme ht --n 64
echo "# product library (no stamps), same workloads" >> "$OUT"
for m in "points dpf 32" "keys dpf 32" "keys ht 64"; do
  set -- $m
  timeout 300 python3 tools/keys_clock.py --mode $1 --scheme $2 --n $3 2>>"$OUT.err" | tee -a "$OUT"
done
