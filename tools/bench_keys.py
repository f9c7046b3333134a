#!/usr/bin/env python3
"""Time one independent-keys workload: Gen (level-major) once, then eval_keys (used under rocprofv3)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import time_steady, time_steps
ap = argparse.ArgumentParser()
ap.add_argument("--scheme", default="dpf"); ap.add_argument("--n", type=int, default=32)
ap.add_argument("--log2-keys", type=int, default=24); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--prg", default="aes128_mmo")
ap.add_argument("--same-key", action="store_true", help="every lane evaluates key 0 (no HBM streaming; compute-only reference)")
ap.add_argument("--steady", action="store_true", help="warm the clocks up and time >= 150 ms of launches")
a = ap.parse_args()
eng = fss_amd.Engine(0, prg=a.prg)
nk, n = 1 << a.log2_keys, a.n
gen = torch.Generator(device="cuda").manual_seed(100 + n)
s0s = torch.randint(-2**31, 2**31, (nk, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
betas = torch.randint(-2**31, 2**31, (nk, 4), dtype=torch.int32, device="cuda", generator=gen)
dt = torch.int32 if n <= 32 else torch.int64
lo, hi = (-2**31, 2**31) if n <= 32 else (-2**63, 2**63 - 1)
alphas = torch.randint(lo, hi, (nk,), dtype=dt, device="cuda", generator=gen)
xs = torch.randint(lo, hi, (nk,), dtype=dt, device="cuda", generator=gen)
sid = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF, "ht": fss_amd.HT, "grotto": fss_amd.GROTTO}[a.scheme]
g = fss_amd.BYTES if a.scheme == "grotto" else fss_amd.U64
keys, ocws = eng.gen_level_major(sid, n, g, s0s, alphas, betas)
seeds0 = s0s[:, 0].contiguous()
ys = (torch.empty((nk,), dtype=torch.uint8, device="cuda") if a.scheme == "grotto"
      else torch.empty((nk, 4), dtype=torch.int32, device="cuda"))
fn = lambda: eng.eval_keys(keys, g, 0, seeds0, xs, ocws=ocws, out=ys)
wall, kms = time_steady(fn, a.steps) if a.steady else time_steps(fn, a.steps, 2, None)
print(f"{a.prg} {a.scheme} n={n} keys=2^{a.log2_keys}: {kms:.3f} ms  {nk/kms/1e6:.3f} G evals/s")
