set -x
FSS_AMD_TUNE=1 FSS_AMD_QUAD_TAIL=4 FSS_AMD_MIX_TAIL_UNITS=1 python -m pytest tests/test_gpu_two_then_one.py -x -q -k shared_key 2>&1 | tail -3
python -m pytest tests/test_gpu_errors.py -x -q -m gpu -k "forced" 2>&1 | tail -3
bash tools/exp_quad_tail.sh
python - <<'P'
import json
for ln in open('gpurun_out/exp_r4/quad_tail.txt'):
    if ln.startswith('#'): print(ln.strip()); continue
    d=json.loads(ln); b=d.get('breakdown_us',{})
    print("   kernel_ms %.4f frac2.4 %.4f clk %.3f | loop %.1f floor %.1f above %.1f tail %.1f (xwg %.1f inwg %.1f)"%(d['kernel_ms'],d['lds_frac_at_2.4GHz'],d.get('in_kernel_clock_GHz_median',0),
      b.get('walk_loop_mean (per wave)',0),b.get('lds_floor_at_measured_clock',0),b.get("walk_loop_minus_floor (LDS idle while the wave's workgroup is walking)",0),
      b.get("idle_tail_mean (wave done -> launch's last wave done)",0),b.get('idle_tail_across_workgroups (mean over workgroups of: launch end - workgroup end)',0),b.get('idle_tail_inside_workgroup (mean over waves of: workgroup end - wave end)',0)))
P
