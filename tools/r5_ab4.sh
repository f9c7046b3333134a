#!/bin/bash
# round 5, fourth GPU call: batch EvalAll with scalar key addressing (UNI) A/B, what bounds the shared-prefix walk
# (sorted points; PMC: L2 hit rate, fabric reads, wave-parked cycles), parity of the batch forms
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/r5_ab4"
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch or eval_all" > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
for rep in 1 2; do
  for UNI in 1 0; do
    for SCH in dpf ht; do
      FSS_AMD_TUNE=1 FSS_AMD_BATCH_UNI=$UNI timeout 300 python tools/bench_batch.py --scheme $SCH --shapes 20x64,14x4096,16x1024,24x4,10x65536 >> "$OUT/batch_ab.txt" 2>> "$OUT/err.txt"
    done
  done
done
cat "$OUT/batch_ab.txt"
timeout 300 python tools/shared_prefix_sweep.py --schemes dpf --kmin 18 --kmax 24 --log2-points 24 --sorted > "$OUT/sweep_sorted.txt" 2>> "$OUT/err.txt"
cat "$OUT/sweep_sorted.txt"
cd /tmp && export TMPDIR=/tmp
for K in 18 23; do
  i=0
  for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT/pmc_k${K}_$i" -- python3 $R/tools/shared_prefix_sweep.py --schemes dpf --kmin $K --kmax $K --log2-points 24 > "$OUT/pmc_k${K}_$i.log" 2>&1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r5_ab4"
for d in sorted(glob.glob(out + "/pmc_k*_[0-9]")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "eval_points_kernel" in k and "true>" in k.replace(" ", "")[-12:]:
                a = agg[row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    print(os.path.basename(d), {c: (round(v[0] / max(v[1], 1), 1), v[1]) for c, v in agg.items()})
PY
