#!/bin/bash
# NOTE: the dma variant library needs FSS_EXP_CW_LDSDMA, removed from the sources after this run (negative result).
# Round 3, second batch: the LDS-DMA variant of the independent-key walk against the product's
# walk, interleaved; the headline batch with two evaluations per lane (where the time goes, for the mixed-ILP idea).
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
K="$O/keys_clock_dma.txt"; : > "$K"
echo "# parity lib=dma" >> "$K"
FSS_AMD_LIB="$R/fss_amd/libfss_amd_dma.so" timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "eval_keys or reference_kats" 2>&1 | tail -2 >> "$K"
for rep in 1 2; do
  kc "$K" diag -- --mode keys --scheme dpf --n 32
  kc "$K" dma -- --mode keys --scheme dpf --n 32
done
B="$O/points_2e24_ilp.txt"; : > "$B"
for rep in 1 2; do
  kc "$B" diag -- --mode points --scheme dpf --n 32 --log2 24
  kc "$B" diag FSS_AMD_ILP=2 -- --mode points --scheme dpf --n 32 --log2 24
  kc "$B" diag FSS_AMD_ILP=2 FSS_AMD_TAIL_PCT=16 -- --mode points --scheme dpf --n 32 --log2 24
done
kc "$B" diag -- --mode points --scheme ht --n 64 --log2 24
kc "$B" diag FSS_AMD_ILP=2 -- --mode points --scheme ht --n 64 --log2 24
echo done
