#!/bin/bash
# round 5, third GPU call: launch-plan A/B of the shared-prefix walk (k = 23, 2^24 points), new multi-GPU test, parity subset
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r5_ab3"
mkdir -p "$OUT"
cd "$R"
run() {  # name, env...
  name=$1; shift
  echo "== $name: $*" >> "$OUT/walk_ab.txt"
  env FSS_AMD_TUNE=1 "$@" timeout 300 python tools/shared_prefix_sweep.py --schemes dpf,dcf --kmin 23 --kmax 23 --log2-points 24 2>> "$OUT/err.txt" | grep "k=23" >> "$OUT/walk_ab.txt"
}
for rep in 1 2; do
run default FSS_AMD_NOP=1
run dyn FSS_AMD_DYN=1
run dyn_tail8 FSS_AMD_DYN=1 FSS_AMD_TAIL_MINWORK=8
run dyn_tail8_nomix FSS_AMD_DYN=1 FSS_AMD_TAIL_MINWORK=8 FSS_AMD_MIX_MID_PCT=0 FSS_AMD_MIX_TAIL_UNITS=0
run ilp1 FSS_AMD_ILP=1
run ilp1_dyn_tail8 FSS_AMD_ILP=1 FSS_AMD_DYN=1 FSS_AMD_TAIL_MINWORK=8
run static_noprio FSS_AMD_DYN=0 FSS_AMD_STATIC_PRIO=0
run static_rot FSS_AMD_DYN=0 FSS_AMD_STATIC_PRIO=1
run static_mix1 FSS_AMD_DYN=0 FSS_AMD_TAIL_MINWORK=8
done
cat "$OUT/walk_ab.txt"
timeout 900 python -m pytest tests/test_gpu_threads.py tests/test_gpu_lifecycle.py tests/test_gpu_two_then_one.py tests/test_sharding.py -x -q -m gpu > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
timeout 600 tests/cxx/multi_gpu_test > "$OUT/multi_gpu_test.txt" 2>&1; echo "multi_gpu_test rc=$?"; tail -15 "$OUT/multi_gpu_test.txt"
