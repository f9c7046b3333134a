#!/usr/bin/env python3
"""Time fss_amd EvalAll for one scheme/size (used under rocprofv3 to see the per-stage split)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import make_key, time_steady, time_steps
ap = argparse.ArgumentParser()
ap.add_argument("--scheme", default="dpf"); ap.add_argument("--n", type=int, default=26)
ap.add_argument("--log2-leaves", type=int, default=None); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--steady", action="store_true", help="warm the clocks up and time >= 150 ms of launches")
ap.add_argument("--group", default="u64", choices=["u64", "bytes", "u128"])
a = ap.parse_args()
eng = fss_amd.default_engine(0)
G = {"u64": fss_amd.U64, "bytes": fss_amd.BYTES, "u128": fss_amd.U128}[a.group]
sid, s0s, cws, ocw, _, _ = make_key(eng, a.scheme, a.n, G, 7 + a.n)
nl = 1 << (a.log2_leaves or a.n)
sd, cd = s0s[0].cuda(), cws.cuda()
od = None if ocw is None else ocw.cuda()
ys = torch.empty((nl, 4), dtype=torch.int32, device="cuda")
fn = lambda: eng.eval_all(sid, a.n, G, 0, sd, cd, ocw=od, first_leaf=0, nleaves=nl, out=ys)
wall, kms = time_steady(fn, a.steps) if a.steady else time_steps(fn, a.steps, 2, None)
print(f"{a.scheme} EvalAll n={a.n} leaves=2^{nl.bit_length()-1}: {kms:.3f} ms  {nl/kms/1e6:.2f} G leaves/s")
