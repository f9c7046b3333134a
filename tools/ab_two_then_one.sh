#!/bin/bash
# Round 3, third batch: the point walk "two, then one" (two evaluations per lane for the body of the batch, one per lane
# for its end) against the pure forms, diagnostic library with stamps, interleaved.
export FSS_AMD_TUNE=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
O="$R/gpurun_out/exp"; mkdir -p "$O"
kc() {
  local out="$1" tag="$2"; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "# lib=$tag ${envs[*]} :: $*" >> "$out"
  env "${envs[@]}" FSS_AMD_LIB="$R/fss_amd/libfss_amd_$tag.so" timeout 300 python3 tools/keys_clock.py "$@" 2>>"$out.err" | grep workload >> "$out"
}
B="$O/mixed_ilp.txt"; : > "$B"
ONE="FSS_AMD_ILP=1"; TWO="FSS_AMD_ILP=2 FSS_AMD_MIX_MID_PCT=0 FSS_AMD_MIX_TAIL_UNITS=0"
for rep in 1 2; do
  for w in "points dpf 32 24" "points dpf 32 20" "points dpf 32 22" "keys dpf 32 24" "points ht 64 24" "keys ht 64 24"; do
    set -- $w
    kc "$B" diag -- --mode $1 --scheme $2 --n $3 --log2 $4
    kc "$B" diag $ONE -- --mode $1 --scheme $2 --n $3 --log2 $4
  done
done
kc "$B" diag $TWO -- --mode points --scheme dpf --n 32 --log2 24
for v in "FSS_AMD_MIX_MID_PCT=2" "FSS_AMD_MIX_MID_PCT=12" "FSS_AMD_MIX_MID_PCT=6 FSS_AMD_TAIL_PCT=4"; do
  kc "$B" diag $v -- --mode points --scheme dpf --n 32 --log2 24
done
for v in "FSS_AMD_MIX_TAIL_UNITS=1" "FSS_AMD_MIX_TAIL_UNITS=3"; do
  kc "$B" diag $v -- --mode points --scheme dpf --n 32 --log2 20
  kc "$B" diag $v -- --mode points --scheme dpf --n 32 --log2 22
done
kc "$B" diag -- --mode points --scheme dpf --n 16 --log2 24
kc "$B" diag $ONE -- --mode points --scheme dpf --n 16 --log2 24
echo done
