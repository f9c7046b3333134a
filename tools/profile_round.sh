#!/bin/bash
# A round's evidence bundle from ONE gpurun lease: tools/profile_workloads.sh <tag> (unprofiled line, kernel trace, PMC
# passes, shared-prefix passes -- all on one box), then the summaries copied to gpurun_out/profiles_<tag>/ (what travels
# back; the raw rocprofv3 directories stay on the box).  Copy that directory to profiles/<tag>/ and commit it.
# usage (from the build container): gpurun --timeout 2400 -- 'bash tools/profile_round.sh r5'
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r5}"
cd "$R"
( time bash tools/profile_workloads.sh "$TAG" ) > "gpurun_out/prof_$TAG.log" 2>&1
OUT="$R/gpurun_out/profiles_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
P="$R/gpurun_out/prof_$TAG"
cp $P/box.txt $P/bench_line.json $P/bench_line_under_rocprof.json $P/bench_detail.json $P/kernel_stats.csv $P/kernel_stats_by_workload.txt \
   $P/pmc_workloads.json $P/pmc_summary.json $P/shared_prefix.txt "$OUT/" 2>/dev/null
for W in dpf_points dpf_2e20 dcf_2e20 evalall_c4 ht_c5 grotto_c5; do
  cp $P/$W/line.json "$OUT/line_$W.json" 2>/dev/null
  find $P/$W/trace -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$W.csv" \; 2>/dev/null
done
tail -3 "gpurun_out/prof_$TAG.log"
rm -rf "$P"
ls -la "$OUT"
