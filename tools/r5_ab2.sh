#!/bin/bash
# round 5, second GPU call: parity of the fused node table + AUTO plan, shared-prefix sweep, a full bench line
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r5_ab2"
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shared_prefix or grotto or eval_points" > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
timeout 600 python -m pytest tests/test_gpu_threads.py tests/test_gpu_lifecycle.py tests/test_gpu_host_path.py -x -q -m gpu > "$OUT/pytest2.txt" 2>&1
tail -3 "$OUT/pytest2.txt"
timeout 300 tests/cxx/dropin_test > "$OUT/dropin_test.txt" 2>&1; echo "dropin_test rc=$?"
timeout 900 python tools/shared_prefix_sweep.py --kmin 15 --kmax 25 --log2-points 24,22,20,19,18 > "$OUT/sweep.txt" 2>> "$OUT/err.txt"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_line.json" 2>> "$OUT/err.txt"
cp gpurun_out/bench_detail.json "$OUT/bench_detail.json" 2>/dev/null
tail -c 1500 "$OUT/bench_line.json"
