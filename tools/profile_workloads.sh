#!/bin/bash
# rocprofv3 evidence for every workload of bench.py's default line, ALL FROM ONE gpurun LEASE (one box, one clock):
#   (0) the box's identity (hostname, GPU serial / unique id) and the DEFAULT command UNPROFILED (bench_line.json);
#   (1) the same command under --kernel-trace --stats (kernel_stats.csv; tools/trace_phases.py splits it by workload:
#       the headline kernel's last-20 average there must agree with (0)'s ms_per_step within 1 %);
#   (2) per workload, `bench.py --workload <w> --no-cpu` under --kernel-trace --stats and under four SEPARATE --pmc
#       passes (LDS / cycles; VALU; FETCH_SIZE; WRITE_SIZE -- MI355X_MICROARCH.md: separate passes, FETCH_SIZE x2 on
#       gfx950);
#   (3) the shared-prefix plan (tools/shared_prefix_sweep.py at the library's own prefix_bits) under a kernel trace and
#       three PMC passes (L2 hit rate and fabric reads of the node gathers; wave / LDS cycles; FETCH_SIZE).
# Output: gpurun_out/prof_<tag>/ with pmc_workloads.json (what bench.py replays as `roofline.traffic` of the
# sub-records, each entry stamped with the digest of the sources its kernel is compiled from), pmc_summary.json (the
# headline's, the format of rounds 1-2) and kernel_stats_by_workload.txt.  Copy the summaries to profiles/r<N>/.
# usage: tools/profile_workloads.sh <tag>
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r5}"
OUT="$R/gpurun_out/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
{ echo "hostname: $(hostname)"; echo "date: $(date -u +%FT%TZ)"; /opt/rocm/bin/rocm-smi --showserial --showuniqueid --showproductname 2>&1 | grep -v "^=\|^$" ; } > "$OUT/box.txt" 2>&1
cd "$R"
timeout 900 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
cp "$R/gpurun_out/bench_detail.json" "$OUT/bench_detail.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 $R/bench.py --steps 20 --warmup 5 > "$OUT/bench_line_under_rocprof.json" 2> "$OUT/trace_default.log"
python3 $R/tools/trace_phases.py "$OUT/trace_default" > "$OUT/kernel_stats_by_workload.txt"
find "$OUT/trace_default" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
for W in dpf_points dpf_2e20 dcf_2e20 evalall_c4 ht_c5 grotto_c5; do
  EXTRA="--steps 10 --warmup 5"
  [ "$W" = dpf_points ] && EXTRA="--only-headline --steps 20 --warmup 5"
  BENCH="python3 $R/bench.py --workload $W --no-cpu --no-clk $EXTRA"
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
# (3) the shared-prefix plan: node-table launch + walk, DPF and DCF, 2^24 and 2^20 points
SWEEP="python3 $R/tools/shared_prefix_sweep.py --auto-k --log2-points 24,20"
mkdir -p "$OUT/shared_prefix"
timeout 600 $SWEEP > "$OUT/shared_prefix/unprofiled.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/shared_prefix/trace" -- $SWEEP > "$OUT/shared_prefix/trace.txt" 2>&1
i=0
for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/shared_prefix/pmc$i" -- python3 $R/tools/shared_prefix_sweep.py --auto-k --schemes dpf --log2-points 24 > "$OUT/shared_prefix/pmc$i.log" 2>&1
done
python3 $R/tools/summarize_shared_prefix.py "$OUT/shared_prefix" > "$OUT/shared_prefix.txt" 2>&1
