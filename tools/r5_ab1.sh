#!/bin/bash
# round 5, first GPU call: parity of the two changed kernels, then A/B against the round-4 library (libfss_amd_r4.so)
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r5_ab1"
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grotto or shared_prefix" > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
for rep in 1 2; do
  for LIB in r4 new; do
    [ $LIB = r4 ] && export FSS_AMD_LIB="$R/fss_amd/libfss_amd_r4.so" || unset FSS_AMD_LIB
    timeout 300 python bench.py --workload grotto_c5 --no-cpu --steps 10 --warmup 5 > "$OUT/grotto_${LIB}_$rep.json" 2>> "$OUT/err.txt"
  done
done
for LIB in r4 new; do
  [ $LIB = r4 ] && export FSS_AMD_LIB="$R/fss_amd/libfss_amd_r4.so" || unset FSS_AMD_LIB
  timeout 600 python tools/shared_prefix_sweep.py --kmin 18 --kmax 26 > "$OUT/sweep_$LIB.txt" 2>> "$OUT/err.txt"
done
unset FSS_AMD_LIB
grep -h -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.]*' "$OUT"/grotto_*.json | paste - - 
