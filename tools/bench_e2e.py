#!/usr/bin/env python3
"""End-to-end rate of the headline workload when x comes from and y goes back to (pinned) host memory:
H2D of xs, the walk, D2H of ys -- serial on one stream, and pipelined in chunks over three streams.
The kernel-only number is bench.py's; this is the PCIe-inclusive one SURVEY 8(d) asks to state separately."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import make_key
ap = argparse.ArgumentParser()
ap.add_argument("--log2-points", type=int, default=24); ap.add_argument("--chunks", type=int, default=8)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
eng = fss_amd.default_engine(0)
n, N = 32, 1 << a.log2_points
sid, s0s, cws, _, _, _ = make_key(eng, "dpf", n, fss_amd.U64, 35)
sd, cd = s0s[0].cuda(), cws.cuda()
xs_h = torch.randint(0, 2**31 - 1, (N,), dtype=torch.int32).pin_memory()
ys_h = torch.empty((N, 4), dtype=torch.int32).pin_memory()
xs_d = torch.empty((N,), dtype=torch.int32, device="cuda")
ys_d = torch.empty((N, 4), dtype=torch.int32, device="cuda")

def serial():
    xs_d.copy_(xs_h, non_blocking=True)
    eng.eval_points(sid, n, fss_amd.U64, 0, sd, cd, xs_d, out=ys_d)
    ys_h.copy_(ys_d, non_blocking=True)

streams = [torch.cuda.Stream() for _ in range(3)]
def pipelined():
    C = N // a.chunks
    for c in range(a.chunks):
        st = streams[c % 3]
        sl = slice(c * C, (c + 1) * C)
        with torch.cuda.stream(st):
            xs_d[sl].copy_(xs_h[sl], non_blocking=True)
            eng.eval_points(sid, n, fss_amd.U64, 0, sd, cd, xs_d[sl], out=ys_d[sl])
            ys_h[sl].copy_(ys_d[sl], non_blocking=True)

for name, fn in (("serial", serial), (f"pipelined x{a.chunks}", pipelined)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print(f"{name}: {dt*1e3:.2f} ms per 2^{a.log2_points} points = {N/dt/1e9:.2f} G evals/s  ({N*20/dt/1e9:.1f} GB/s over PCIe, both directions summed)")
ref = ys_d.clone(); torch.cuda.synchronize()
assert bool((ys_h.cuda() == ref).all())
