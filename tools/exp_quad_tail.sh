#!/bin/bash
# Round 4: end a fixed-share launch on quad-lane walks -- each wave's last 64 evaluations as four
# granules of 16, one walk per quad of lanes, taken from the workgroup's LDS counter (FSS_AMD_QUAD_TAIL, walk_points_quad)
# -- against the product plan, at BASELINE configs[1] / [2]'s size and one size up.  Diagnostic library with stamps,
# variants interleaved, two rounds.  Output: gpurun_out/exp_r4/quad_tail.txt
export FSS_AMD_TUNE=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
O="$R/gpurun_out/exp_r4"; mkdir -p "$O"
kc() {
  local out="$1" tag="$2"; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "# lib=$tag ${envs[*]} :: $*" >> "$out"
  env "${envs[@]}" FSS_AMD_LIB="$R/fss_amd/libfss_amd_$tag.so" timeout 300 python3 tools/keys_clock.py "$@" 2>>"$out.err" | grep workload >> "$out"
}
B="$O/quad_tail.txt"; : > "$B"
for rep in 1 2; do
  # DPF 2^20: product plan (128 two per lane + 2 x 64 one per lane) | 128 + 64 + 4 quad granules | 3 x 64 one per lane + 4 quad granules
  kc "$B" diag -- --mode points --scheme dpf --n 32 --log2 20
  kc "$B" diag FSS_AMD_QUAD_TAIL=4 FSS_AMD_MIX_TAIL_UNITS=1 -- --mode points --scheme dpf --n 32 --log2 20
  kc "$B" diag FSS_AMD_QUAD_TAIL=4 FSS_AMD_ILP=1 -- --mode points --scheme dpf --n 32 --log2 20
  # DCF 2^20: 4 x 64 one per lane | 3 x 64 + 4 quad granules
  kc "$B" diag -- --mode points --scheme dcf --n 32 --log2 20
  kc "$B" diag FSS_AMD_QUAD_TAIL=4 -- --mode points --scheme dcf --n 32 --log2 20
  # DPF 2^21 (a GPU's share of the metric's batch split eight ways): product | 4 quad granules in place of the last unit
  kc "$B" diag -- --mode points --scheme dpf --n 32 --log2 21
  kc "$B" diag FSS_AMD_QUAD_TAIL=4 -- --mode points --scheme dpf --n 32 --log2 21
done
echo done
