#!/usr/bin/env python3
"""Time batched EvalAll (fss::gpu::*EvalAllGpuBatch) for a few (in_bits, nkeys) shapes."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import time_steady
ap = argparse.ArgumentParser()
ap.add_argument("--scheme", default="dpf"); ap.add_argument("--prg", default="aes128_mmo")
ap.add_argument("--shapes", default="20x64,14x4096,10x1024,16x16,12x256,8x65536")
a = ap.parse_args()
eng = fss_amd.Engine(0, prg=a.prg)
G = fss_amd.U64
sid = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF, "ht": fss_amd.HT}[a.scheme]
for shape in a.shapes.split(","):
    n, nk = (int(v) for v in shape.split("x"))
    gen = torch.Generator(device="cuda").manual_seed(400 + n)
    s0s = torch.randint(-2**31, 2**31, (nk, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    betas = torch.randint(-2**31, 2**31, (nk, 4), dtype=torch.int32, device="cuda", generator=gen)
    alphas = torch.randint(0, 1 << n, (nk,), dtype=torch.int32, device="cuda", generator=gen)
    cws, ocws = eng.gen(sid, n, G, s0s, alphas, betas)
    seeds0 = s0s[:, 0].contiguous()
    yb = torch.empty((nk, 1 << n, 4), dtype=torch.int32, device="cuda")
    _, kms = time_steady(lambda: eng.eval_all_batch(sid, n, G, 0, seeds0, cws, ocws=ocws, out=yb))
    aes = {"dpf": 2.0, "dcf": 4.0, "ht": 1.5}[a.scheme]
    frac = (nk << n) * aes * 640 / (kms * 1e-3) / 1e9 / 78643.2
    print(f"{a.prg} {a.scheme} batch n={n} x {nk}: {kms*1e3:.1f} us  {(nk << n)/kms/1e6:.2f} G out/s  lds_frac={frac:.3f}  "
          f"uni={os.environ.get('FSS_AMD_BATCH_UNI', 'default')}  {eng.last_kernel()}")
