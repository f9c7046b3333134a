#!/usr/bin/env python3
"""Time one shared-key point-evaluation workload (used under rocprofv3 for per-kernel counters)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import make_key, time_steady, time_steps
ap = argparse.ArgumentParser()
ap.add_argument("--scheme", default="dpf"); ap.add_argument("--n", type=int, default=32)
ap.add_argument("--log2-points", type=int, default=24); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--prg", default="aes128_mmo")
ap.add_argument("--steady", action="store_true", help="warm the clocks up and time >= 150 ms of launches")
a = ap.parse_args()
eng = fss_amd.Engine(0, prg=a.prg)
sid, s0s, cws, ocw, _, _ = make_key(eng, a.scheme, a.n, fss_amd.U64, 3 + a.n)
batch = 1 << a.log2_points
xs = torch.randint(0, 1 << min(a.n, 31), (batch,), dtype=torch.int32, device="cuda")
sd, cd = s0s[0].cuda(), cws.cuda()
od = None if ocw is None else ocw.cuda()
ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")
fn = lambda: eng.eval_points(sid, a.n, fss_amd.U64, 0, sd, cd, xs, ocw=od, out=ys)
wall, kms = time_steady(fn, a.steps) if a.steady else time_steps(fn, a.steps, 2, None)
print(f"{a.prg} {a.scheme} n={a.n} points=2^{a.log2_points}: {kms:.3f} ms  {batch/kms/1e6:.3f} G evals/s")
