#!/bin/bash
# round 5, fifth GPU call: the per-thread drop-in path (rates in dropin_device_test), then the whole -m gpu suite
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r5_ab5"
mkdir -p "$OUT"
cd "$R"
timeout 600 tests/cxx/dropin_device_test > "$OUT/dropin_device_test.txt" 2>&1; echo "dropin_device_test rc=$?"
grep -n "evals/s\|FAILED\|^OK" "$OUT/dropin_device_test.txt"
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > "$OUT/pytest_gpu.txt" 2>&1
tail -8 "$OUT/pytest_gpu.txt"
