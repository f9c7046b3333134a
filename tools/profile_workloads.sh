#!/bin/bash
# rocprofv3 evidence for every workload of bench.py's default line (round 3): 
#   (1) the DEFAULT command under --kernel-trace --stats (kernel_stats.csv; tools/trace_phases.py splits it by workload),
#   (2) per workload, `bench.py --workload <w> --no-cpu` under --kernel-trace --stats and under three SEPARATE --pmc
#       passes (LDS / cycles; FETCH_SIZE; WRITE_SIZE -- MI355X_MICROARCH.md: separate passes, FETCH_SIZE x2 on gfx950).
# Output: gpurun_out/prof_<tag>/ with pmc_workloads.json (what bench.py replays as `roofline.traffic` of the
# sub-records, each entry stamped with the digest of the sources its kernel is compiled from), pmc_summary.json (the
# headline's, the format of rounds 1-2) and kernel_stats_by_workload.txt.  Copy the summaries to profiles/r<N>/.
# usage: tools/profile_workloads.sh <tag>
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r4}"
OUT="$R/gpurun_out/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 $R/bench.py --steps 20 --warmup 5 > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/trace_default.log"
python3 $R/tools/trace_phases.py "$OUT/trace_default" > "$OUT/kernel_stats_by_workload.txt"
find "$OUT/trace_default" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
for W in dpf_points dpf_2e20 dcf_2e20 evalall_c4 ht_c5 grotto_c5; do
  EXTRA="--steps 10 --warmup 5"
  [ "$W" = dpf_points ] && EXTRA="--only-headline --steps 20 --warmup 5"
  BENCH="python3 $R/bench.py --workload $W --no-cpu $EXTRA"
  mkdir -p "$OUT/$W"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$W/trace" -- $BENCH > "$OUT/$W/line.json" 2> "$OUT/$W/trace.log"
  i=0
  for PMC in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
             "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/$W/pmc$i" -- $BENCH > "$OUT/$W/pmc$i.log" 2>&1
  done
done
python3 $R/tools/summarize_workload_profiles.py "$OUT"
