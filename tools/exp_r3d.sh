#!/bin/bash
# NOTE: libfss_amd_pf.so was built from a working tree with the prefetch (WaveQ::peek); the change was not kept
# (profiles/r3/prefetch_x_raw.txt, tools/ubench/REJECTED_VARIANTS.md).
# Round 3, fourth batch: the next chunk's points requested a chunk ahead (libfss_amd_pf.so) against loading them at the
# start of the chunk (libfss_amd_diag.so, the previous commit), interleaved; both diagnostic builds with stamps.
export FSS_AMD_TUNE=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
O="$R/gpurun_out/exp_r3"; mkdir -p "$O"
kc() {
  local out="$1" tag="$2"; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "# lib=$tag ${envs[*]} :: $*" >> "$out"
  env "${envs[@]}" FSS_AMD_LIB="$R/fss_amd/libfss_amd_$tag.so" timeout 300 python3 tools/keys_clock.py "$@" 2>>"$out.err" | grep workload >> "$out"
}
B="$O/prefetch_x.txt"; : > "$B"
echo "# parity lib=pf" >> "$B"
FSS_AMD_LIB="$R/fss_amd/libfss_amd_pf.so" timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_two_then_one.py -q -x -m gpu -p no:cacheprovider -k "eval_points or eval_keys or reference_kats or ragged" 2>&1 | tail -2 >> "$B"
for rep in 1 2; do
  for w in "points dpf 32 24" "points dpf 16 24" "points dpf 32 20" "points dcf 32 24" "keys dpf 32 24" "points ht 64 24" "keys ht 64 24"; do
    set -- $w
    kc "$B" diag -- --mode $1 --scheme $2 --n $3 --log2 $4
    kc "$B" pf -- --mode $1 --scheme $2 --n $3 --log2 $4
  done
done
echo done
