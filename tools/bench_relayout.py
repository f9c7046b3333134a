#!/usr/bin/env python3
"""Time the key-major <-> level-major transpositions (HBM-bound copies; fss::gpu::*RelayoutGpu, point_eval_gpu.cuh:324-375)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import time_steady
ap = argparse.ArgumentParser()
ap.add_argument("--log2-keys", type=int, default=22)
a = ap.parse_args()
eng = fss_amd.default_engine(0)
nk = 1 << a.log2_keys
for name, sid, n in (("dpf", fss_amd.DPF, 32), ("dcf", fss_amd.DCF, 32), ("ht", fss_amd.HT, 64)):
    slots = n if name == "ht" else n + 1
    cws = torch.randint(-2**31, 2**31, (nk, slots, 8), dtype=torch.int32, device="cuda")
    _, kms = time_steady(lambda: eng.relayout(sid, n, cws))
    rd = nk * slots * 32
    wr = nk * (n * (32 if name == "dcf" else 16) + 24)
    print(f"{name} relayout n={n} 2^{a.log2_keys} keys: {kms:.3f} ms  {nk/kms/1e6:.3f} G keys/s  read {rd/kms/1e6:.0f} GB/s + write {wr/kms/1e6:.0f} GB/s = {(rd+wr)/kms/1e6:.0f} GB/s")
    keys = eng.relayout(sid, n, cws)
    _, kms = time_steady(lambda: eng.key_major(keys))
    print(f"{name} key_major n={n}: {kms:.3f} ms  {(rd+wr)/kms/1e6:.0f} GB/s")
    del cws, keys
