#!/usr/bin/env python3
"""Time the raw PRG entry point (fss_amd_prg_gen; reference Aes128Soft<mul>::Gen bench, doc/bench_aes128_soft.md)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import time_steady
ap = argparse.ArgumentParser()
ap.add_argument("--log2-seeds", type=int, default=24); ap.add_argument("--prg", default="aes128_mmo")
a = ap.parse_args()
eng = fss_amd.Engine(0, prg=a.prg)
n = 1 << a.log2_seeds
seeds = torch.randint(-2**31, 2**31, (n, 4), dtype=torch.int32, device="cuda")
for mul in (1, 2, 4):
    _, kms = time_steady(lambda: eng.prg_gen(mul, seeds))
    print(f"{a.prg} prg Gen mul={mul} 2^{a.log2_seeds} seeds: {kms:.3f} ms  {n/kms/1e6:.2f} G Gen/s  lds {n*mul/kms/1e6*640/78643.2:.2f}  hbm {n*16*(1+mul)/kms/1e6:.0f} GB/s")
