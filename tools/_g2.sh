set -x
python -m pytest tests/test_gpu_host_path.py -x -q 2>&1 | tail -5
python -m pytest tests -x -q -m gpu -k "dropin or cxx or sample" 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_line2.json 2> gpurun_out/r4_bench_err2.log; echo "rc=$?"
python - <<'P'
import json
d=json.load(open('gpurun_out/r4_bench_line2.json'))
print(d["value"], d["roofline"]["frac"], d["sub"]["e2e_pcie"])
P
