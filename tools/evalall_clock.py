#!/usr/bin/env python3
"""Per-wave stamps of the EvalAll leaf stage (diagnostic library, FSS_AMD_LIB=fss_amd/libfss_amd_diag.so): how evenly
the workgroups of one launch finish and the clock the stage runs at.  The stamps are taken by expand_kernel's leaf stage,
so the plan runs as the chain of launches here (FSS_AMD_FUSED=0), not as the one-launch form the product uses."""
import argparse, ctypes as C, json, os, sys
os.environ["FSS_AMD_TUNE"] = "1"
os.environ["FSS_AMD_FUSED"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, fss_amd
from bench import make_key, time_steps
ap = argparse.ArgumentParser()
ap.add_argument("--scheme", default="dpf"); ap.add_argument("--n", type=int, default=28)
ap.add_argument("--log2-leaves", type=int, default=25); ap.add_argument("--seconds", type=float, default=2.0)
a = ap.parse_args()
eng = fss_amd.default_engine(0)
sid, s0s, cws, ocw, _, _ = make_key(eng, a.scheme, a.n, fss_amd.U64, 7 + a.n)
nl = 1 << a.log2_leaves
sd, cd = s0s[0].cuda(), cws.cuda()
od = None if ocw is None else ocw.cuda()
ys = torch.empty((nl, 4), dtype=torch.int32, device="cuda")
fn = lambda: eng.eval_all(sid, a.n, fss_amd.U64, 0, sd, cd, ocw=od, first_leaf=0, nleaves=nl, out=ys)
_, est = time_steps(fn, 3, 2, None)
_, kms = time_steps(fn, max(10, int(a.seconds * 1e3 / est)), 5, None)
kname, grid, block = eng.last_kernel()
L = eng._L
line = {"workload": f"{a.scheme} EvalAll n={a.n} 2^{a.log2_leaves} leaves", "ms_per_call": round(kms, 4), "G_leaves_per_s": round(nl / kms / 1e6, 2)}
if hasattr(L, "fss_amd_diag_read_stamps"):
    nrec = grid * (block // 64)
    buf = np.zeros((nrec, 4), dtype=np.uint64)
    L.fss_amd_diag_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert L.fss_amd_diag_read_stamps(eng._h, buf.ctypes.data_as(C.c_void_p), nrec) == 0
    r0, r1 = buf[:, 1].astype(np.float64), buf[:, 3].astype(np.float64)
    span = r1.max() - r0.min()
    clk = (buf[:, 2] - buf[:, 0]).astype(np.float64) / np.maximum(r1 - r0, 1) * 0.1
    per_wg_end = r1.reshape(grid, -1).max(axis=1)
    ends = (r1.max() - r1) / 100.0
    line.update({"leaf_stage_span_us": round(span / 100, 1), "mean_wave_busy_frac_of_span": round(float((r1 - r0).mean() / span), 4),
                 "workgroup_end_spread_us": round(float(per_wg_end.max() - per_wg_end.min()) / 100, 1),
                 "waves_done_before_end_us_p50_p90_max": [round(float(np.percentile(ends, q)), 1) for q in (50, 90, 100)],
                 "in_kernel_clock_GHz_median_min_max": [round(float(np.median(clk)), 3), round(float(clk.min()), 3), round(float(clk.max()), 3)]})
print(json.dumps(line), flush=True)
