#!/usr/bin/env python3
"""Per-phase stamp table of the one-launch EvalAll (evalall_fused_kernel) -- needs the diagnostic library
(FSS_AMD_LIB=fss_amd/libfss_amd_diag.so, -DFSS_DIAG_STAMPS=1).  Every wave stamps s_memrealtime (100 MHz, chip-wide) at
its entry, after the table fill, after its own frontier walks, and at the start (after the workgroup barrier) and end
(its own chunks done) of every stage.  Runs >= --seconds of back-to-back launches and reads the LAST launch's stamps.
Prints, per phase: the mean over workgroups of its span (barrier to barrier), the mean wave-busy time in it, and what
the phase's AES blocks would take at the LDS-lookup floor (32 lookups/clk/CU at the clock measured by the whole
launch's floor fraction is not available per phase: the floor is quoted at 2.4 GHz)."""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, fss_amd
from bench import make_key, time_steps
ap = argparse.ArgumentParser()
ap.add_argument("--scheme", default="dpf"); ap.add_argument("--n", type=int, default=28)
ap.add_argument("--log2-leaves", type=int, default=25); ap.add_argument("--seconds", type=float, default=1.5)
ap.add_argument("--nkeys", type=int, default=0, help="> 0: the batch form (fss_amd_*_eval_all_batch), nkeys whole trees of n levels")
a = ap.parse_args()
eng = fss_amd.default_engine(0)
if a.nkeys:
    sid = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF, "ht": fss_amd.HT}[a.scheme]
    gen = torch.Generator(device="cuda").manual_seed(400 + a.n)
    s0s = torch.randint(-2**31, 2**31, (a.nkeys, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    betas = torch.randint(-2**31, 2**31, (a.nkeys, 4), dtype=torch.int32, device="cuda", generator=gen)
    alphas = torch.randint(0, 1 << a.n, (a.nkeys,), dtype=torch.int32, device="cuda", generator=gen)
    bcws, bocws = eng.gen(sid, a.n, fss_amd.U64, s0s, alphas, betas)
    seeds0 = s0s[:, 0].contiguous()
    nl = a.nkeys << a.n
    ys = torch.empty((a.nkeys, 1 << a.n, 4), dtype=torch.int32, device="cuda")
    fn = lambda: eng.eval_all_batch(sid, a.n, fss_amd.U64, 0, seeds0, bcws, ocws=bocws, out=ys)
else:
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
aes_per_leaf = {"dpf": 2.0, "dcf": 4.0, "ht": 1.5}[a.scheme]
line = {"workload": f"{a.scheme} EvalAll batch n={a.n} x {a.nkeys} keys" if a.nkeys else f"{a.scheme} EvalAll n={a.n} 2^{a.log2_leaves} leaves", "kernel": kname, "grid": grid, "ms_per_call": round(kms, 4),
        "G_leaves_per_s": round(nl / kms / 1e6, 2), "lds_frac_at_2.4GHz": round(nl * aes_per_leaf * 160 / (32 * 256 * 2.4e9) / (kms * 1e-3), 4),
        "env": {k: v for k, v in os.environ.items() if k.startswith("FSS_AMD_") and k not in ("FSS_AMD_LIB", "FSS_AMD_TUNE")}}
SLOTS = 16
if hasattr(L, "fss_amd_diag_read_stamps") and kname == "evalall_fused":
    nw = grid * (block // 64)
    buf = np.zeros((nw * SLOTS,), dtype=np.uint64)
    L.fss_amd_diag_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert L.fss_amd_diag_read_stamps(eng._h, buf.ctypes.data_as(C.c_void_p), nw * SLOTS // 4) == 0
    t = buf.reshape(grid, block // 64, SLOTS).astype(np.float64) / 100.0  # us
    clk = (buf.reshape(grid, block // 64, SLOTS)[:, :, 15] - buf.reshape(grid, block // 64, SLOTS)[:, :, 14]).astype(np.float64) / \
        np.maximum(1.0, (buf.reshape(grid, block // 64, SLOTS)[:, :, 13] - buf.reshape(grid, block // 64, SLOTS)[:, :, 0]).astype(np.float64)) * 0.1
    line["in_kernel_clock_GHz_median"] = round(float(np.median(clk)), 4)
    used = int((t[0, 0, :13] > 0).sum())  # slots 0 .. 12 are the phase stamps (13: realtime at the end, 14 / 15: s_memtime)
    nst = (used - 3) // 2          # stages incl. the leaf stage
    t0 = t[:, :, 0].min()
    end = t[:, :, used - 1].max()
    line["in_kernel_span_us"] = round(float(end - t0), 1)
    line["outside_span_us (launch, dispatch, completion)"] = round(kms * 1e3 - float(end - t0), 1)
    rows = []
    def row(name, start, stop_wave, stop_wg):
        # span: per workgroup, from `start` (max over its waves: the barrier) to stop_wg; busy: mean over waves of stop_wave - start
        span = (stop_wg - start).mean()
        busy = (stop_wave - start[:, None]).mean()
        rows.append({"phase": name, "span_us_mean_over_workgroups": round(float(span), 2), "wave_busy_us_mean": round(float(busy), 2)})
    fill_end = t[:, :, 1]
    rows.append({"phase": "table fill (entry -> tables in LDS)", "span_us_mean_over_workgroups": round(float((fill_end.max(axis=1) - t[:, :, 0].min(axis=1)).mean()), 2),
                 "wave_busy_us_mean": round(float((fill_end - t[:, :, 0]).mean()), 2)})
    row("frontier walks", fill_end.max(axis=1), t[:, :, 2], t[:, :, 2].max(axis=1))
    for s in range(nst):
        st, en = t[:, :, 3 + 2 * s], t[:, :, 4 + 2 * s]
        row("leaf stage" if s == nst - 1 else f"stage {s}", st.max(axis=1), en, en.max(axis=1))
    line["phases"] = rows
    wg_end = t[:, :, used - 1].max(axis=1)
    line["workgroup_end_spread_us (max - min)"] = round(float(wg_end.max() - wg_end.min()), 1)
    line["idle_tail_across_workgroups_us (mean of launch end - workgroup end)"] = round(float((end - wg_end).mean()), 1)
    line["idle_tail_inside_workgroup_us (leaf stage: mean of workgroup end - wave end)"] = round(float((wg_end[:, None] - t[:, :, used - 1]).mean()), 1)
print(json.dumps(line), flush=True)
