#!/usr/bin/env python3
"""In-kernel clock of the point walks (MI355X_MICROARCH.md, DVFS give-back item 6): needs the diagnostic library
(fss_amd/libfss_amd_diag.so, built with -DFSS_DIAG_STAMPS=1 by fss_amd._build.build_variant) through FSS_AMD_LIB.
Runs >= 2 s of back-to-back launches of one workload, then reads the per-workgroup stamps of the last launch:
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups.  Launch variants come from the library's
tunables in the environment (FSS_AMD_ILP, FSS_AMD_BS, FSS_AMD_DYN)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fss_amd
from bench import LDS_PEAK_GBS, LOOKUP_BYTES_PER_AES, aes_blocks_per_eval, hbm_bytes_per_eval, make_key, time_steps

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="keys", choices=["points", "keys"])
ap.add_argument("--scheme", default="dpf")
ap.add_argument("--n", type=int, default=32)
ap.add_argument("--log2", type=int, default=24)
ap.add_argument("--seconds", type=float, default=2.0)
a = ap.parse_args()
eng = fss_amd.default_engine(0)
L = eng._L
n, N = a.n, 1 << a.log2
G = fss_amd.U64
sid = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF, "ht": fss_amd.HT}[a.scheme]
gen = torch.Generator(device="cuda").manual_seed(1000 + n)
dt = torch.int32 if n <= 32 else torch.int64
lo, hi = (-2**31, 2**31) if n <= 32 else (-2**63, 2**63 - 1)
xs = torch.randint(lo, hi, (N,), dtype=dt, device="cuda", generator=gen)
if n < 32:
    xs &= (1 << n) - 1
ys = torch.empty((N, 4), dtype=torch.int32, device="cuda")
if a.mode == "points":
    _, s0s, cws, ocw, _, _ = make_key(eng, a.scheme, n, G, 1 + n)
    sd, cd = s0s[0].cuda(), cws.cuda()
    od = None if ocw is None else ocw.cuda()
    fn = lambda: eng.eval_points(sid, n, G, 0, sd, cd, xs, ocw=od, out=ys)
    shared = True
else:
    s0s = torch.randint(-2**31, 2**31, (N, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    betas = torch.randint(-2**31, 2**31, (N, 4), dtype=torch.int32, device="cuda", generator=gen)
    alphas = torch.randint(lo, hi, (N,), dtype=dt, device="cuda", generator=gen)
    keys, ocws = eng.gen_level_major(sid, n, G, s0s, alphas, betas)
    seeds0 = s0s[:, 0].contiguous()
    del s0s, betas, alphas
    fn = lambda: eng.eval_keys(keys, G, 0, seeds0, xs, ocws=ocws, out=ys)
    shared = False

_, est = time_steps(fn, 3, 2, None)
steps = max(10, int(a.seconds * 1e3 / est))
_, kms = time_steps(fn, steps, 5, None)
kname, grid, block = eng.last_kernel()
line = {"workload": f"{a.scheme} n={n} {'1 key x' if shared else ''} 2^{a.log2} {'points' if shared else 'independent keys'}",
        "ilp": os.environ.get("FSS_AMD_ILP", "default"), "bs": os.environ.get("FSS_AMD_BS", "default"),
        "launches": steps, "kernel_ms": round(kms, 4), "G_evals_per_s": round(N / kms / 1e6, 4), "grid": grid, "block": block}
aes = aes_blocks_per_eval(a.scheme, n)
line["lds_frac_at_2.4GHz"] = round(N / (kms * 1e-3) * aes * LOOKUP_BYTES_PER_AES / 1e9 / LDS_PEAK_GBS, 4)
line["hbm_GBs_algorithmic"] = round(N / (kms * 1e-3) * hbm_bytes_per_eval(a.scheme, n, shared) / 1e9, 1)
if hasattr(L, "fss_amd_diag_read_stamps"):
    nrec = grid * (block // 64)  # one record per wave
    buf = np.zeros((nrec, 4), dtype=np.uint64)
    L.fss_amd_diag_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rc = L.fss_amd_diag_read_stamps(eng._h, buf.ctypes.data_as(C.c_void_p), nrec)
    assert rc == 0, rc
    dcyc = (buf[:, 2] - buf[:, 0]).astype(np.float64)
    dreal = (buf[:, 3] - buf[:, 1]).astype(np.float64)
    # the launch on the 100 MHz clock all XCDs share: how full the chip is kept to the end
    r0, r1 = buf[:, 1].astype(np.float64), buf[:, 3].astype(np.float64)
    span = r1.max() - r0.min()
    line["walk_span_us"] = round(span / 100.0, 1)
    line["mean_wave_busy_frac_of_span"] = round(float((r1 - r0).mean() / span), 4)
    ends = np.sort(r1.max() - r1) / 100.0  # us before the last wave's end at which each wave left its loop
    line["waves_done_before_end_us_p50_p90_max"] = [round(float(np.percentile(ends, q)), 1) for q in (50, 90, 100)]
    per_wg_end = r1.reshape(grid, -1).max(axis=1)
    line["workgroup_end_spread_us"] = round(float(per_wg_end.max() - per_wg_end.min()) / 100.0, 1)
    line["starts_spread_us"] = round(float(r0.max() - r0.min()) / 100.0, 1)
    # by wave slot of the workgroup (slots s, s + 4, s + 8, s + 12 share a SIMD): how long before the launch's end
    # the slot's waves leave their loop, averaged over the workgroups
    slot = (r1.max() - r1).reshape(grid, -1).mean(axis=0) / 100.0
    line["slot_done_before_end_us"] = [round(float(v), 1) for v in slot]
    ok = dreal > 0
    clk = dcyc[ok] / dreal[ok] * 0.1  # GHz: realtime ticks at 100 MHz
    line["in_kernel_clock_GHz_median"] = round(float(np.median(clk)), 4)
    line["in_kernel_clock_GHz_min_max"] = [round(float(clk.min()), 4), round(float(clk.max()), 4)]
    line["shader_cycles_per_launch_median"] = int(np.median(dcyc[ok]))
    # LDS-lookup floor in shader cycles: evals x AES x 160 lookups / (32 lookups/clk/CU x 256 CU)
    floor = N * aes * 160 / (32 * 256)
    line["lds_floor_cycles"] = int(floor)
    line["lds_floor_over_cycles"] = round(floor / float(np.median(dcyc[ok])), 4)
    line["lds_frac_at_measured_clock"] = round(line["lds_frac_at_2.4GHz"] * 2.4 / line["in_kernel_clock_GHz_median"], 4)
    if hasattr(L, "fss_amd_diag_read_entry_stamps"):
        # where a launch's time goes (round 3, profiles/r3/points_2e20_breakdown.txt), in us on the 100 MHz clock:
        #   launch -> first wave's first instruction and last wave's last instruction -> completion (what HIP events see
        #   beyond the in-kernel span), entry -> after the table fill, the walk loop, the idle tail of waves that are done
        ent = np.zeros(nrec, dtype=np.uint64)
        L.fss_amd_diag_read_entry_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        assert L.fss_amd_diag_read_entry_stamps(eng._h, ent.ctypes.data_as(C.c_void_p), nrec) == 0
        re_ = ent.astype(np.float64)
        kernel_us = kms * 1e3
        inkernel = (r1.max() - re_.min()) / 100.0
        fill = (r0 - re_) / 100.0
        busy = (r1 - r0) / 100.0
        idle_tail = (r1.max() - r1) / 100.0
        floor_us = floor / (line["in_kernel_clock_GHz_median"] * 1e3)
        line["breakdown_us"] = {
            "kernel_ms_x1000 (HIP events / launches)": round(kernel_us, 2),
            "outside_in_kernel_span (launch, dispatch of 256 workgroups, completion)": round(kernel_us - inkernel, 2),
            "entry_spread (first wave's entry -> last wave's entry)": round(float(re_.max() - re_.min()) / 100.0, 2),
            "table_fill_mean (entry -> loop, per wave)": round(float(fill.mean()), 2),
            "walk_loop_mean (per wave)": round(float(busy.mean()), 2),
            "idle_tail_mean (wave done -> launch's last wave done)": round(float(idle_tail.mean()), 2),
            "idle_tail_across_workgroups (mean over workgroups of: launch end - workgroup end)": round(float((r1.max() - per_wg_end).mean()) / 100.0, 2),
            "idle_tail_inside_workgroup (mean over waves of: workgroup end - wave end)": round(float((per_wg_end.repeat(block // 64) - r1).mean()) / 100.0, 2),
            "lds_floor_at_measured_clock": round(floor_us, 2),
            "walk_loop_minus_floor (LDS idle while the wave's workgroup is walking)": round(float(busy.mean()) - floor_us, 2),
        }
else:
    line["note"] = "product library: no stamps (set FSS_AMD_LIB to the FSS_DIAG_STAMPS build)"
import json
print(json.dumps(line), flush=True)
