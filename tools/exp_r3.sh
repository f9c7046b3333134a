#!/bin/bash
# NOTE: the variant switches (FSS_EXP_CW_LDSDMA, FSS_EXP_BS768) were removed from the sources after this run (negative
# results: profiles/r3/keys_clock.txt, tools/ubench/REJECTED_VARIANTS.md); the script documents what was measured and how.
# Round 3, VERDICT r2 items 5 and 6: (a) where a 2^20-point launch's time goes (entry / fill / loop / tail stamps of the
# diagnostic library), (b) independent keys with the level's correction words staged by LDS-DMA, (c) the same walk
# with 12 waves per CU.  Variant libraries are built beforehand (they travel with the snapshot):
#   python3 -c "from fss_amd import _build as b; b.build_variant('diag', ['FSS_DIAG_STAMPS=1']); \
#               b.build_variant('dma', ['FSS_DIAG_STAMPS=1', 'FSS_EXP_CW_LDSDMA']); b.build_variant('bs768', ['FSS_DIAG_STAMPS=1', 'FSS_EXP_BS768'])"
export FSS_AMD_TUNE=1
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
O="$R/gpurun_out/exp_r3"; mkdir -p "$O"
kc() {  # kc <outfile> <lib tag> <env...> -- <keys_clock args>
  local out="$1" tag="$2"; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "# lib=$tag ${envs[*]} :: $*" >> "$out"
  env "${envs[@]}" FSS_AMD_LIB="$R/fss_amd/libfss_amd_$tag.so" timeout 300 python3 tools/keys_clock.py "$@" 2>>"$out.err" | grep workload >> "$out"
}
# ---- (a) 2^20 points
B="$O/points_2e20_breakdown.txt"; : > "$B"
for rep in 1 2; do
  kc "$B" diag -- --mode points --scheme dpf --n 32 --log2 20
  kc "$B" diag -- --mode points --scheme dcf --n 32 --log2 20
  kc "$B" diag FSS_AMD_DYN=1 FSS_AMD_TAIL_PCT=25 FSS_AMD_TAIL_MINWORK=1 -- --mode points --scheme dpf --n 32 --log2 20
  kc "$B" diag FSS_AMD_DYN=1 FSS_AMD_TAIL_PCT=25 FSS_AMD_TAIL_MINWORK=1 FSS_AMD_ROTPRIO=2 -- --mode points --scheme dpf --n 32 --log2 20
  kc "$B" diag FSS_AMD_ILP=2 -- --mode points --scheme dpf --n 32 --log2 20
done
kc "$B" diag -- --mode points --scheme dpf --n 32 --log2 24
# ---- (b), (c) independent keys
K="$O/keys_clock.txt"; : > "$K"
# parity of the variants first: the oracle comparisons of the independent-key walks
for v in "dma" "bs768 FSS_AMD_BS=768"; do
  set -- $v
  echo "# parity lib=$1 ${2:-}" >> "$K"
  env ${2:-X=1} FSS_AMD_LIB="$R/fss_amd/libfss_amd_$1.so" timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "eval_keys or reference_kats" 2>&1 | tail -2 >> "$K"
done
for rep in 1 2; do
  kc "$K" diag -- --mode keys --scheme dpf --n 32
  kc "$K" dma -- --mode keys --scheme dpf --n 32
  kc "$K" bs768 FSS_AMD_BS=768 -- --mode keys --scheme dpf --n 32
  kc "$K" diag -- --mode keys --scheme ht --n 64
  kc "$K" bs768 FSS_AMD_BS=768 -- --mode keys --scheme ht --n 64
done
echo done
