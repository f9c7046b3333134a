#!/usr/bin/env python3
"""Time device key generation (level-major, and key-major with --key-major) for the three schemes."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import time_steps
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32); ap.add_argument("--log2-keys", type=int, default=24)
ap.add_argument("--prg", default="aes128_mmo"); ap.add_argument("--schemes", default="dpf,dcf,ht")
ap.add_argument("--key-major", action="store_true", help="fss_amd_*_gen: raw 32-byte Cw slots, (n + 1) per key")
a = ap.parse_args()
eng = fss_amd.Engine(0, prg=a.prg)
nk = 1 << a.log2_keys
g = torch.Generator(device="cuda").manual_seed(1)
s0s = torch.randint(-2**31, 2**31, (nk, 2, 4), dtype=torch.int32, device="cuda", generator=g)
betas = torch.randint(-2**31, 2**31, (nk, 4), dtype=torch.int32, device="cuda", generator=g)
if a.n <= 32:
    al = torch.randint(-2**31, 2**31, (nk,), dtype=torch.int32, device="cuda", generator=g)
else:
    al = torch.randint(-2**63, 2**63 - 1, (nk,), dtype=torch.int64, device="cuda", generator=g)
for name in a.schemes.split(","):
    sch = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF, "ht": fss_amd.HT}[name]
    fn = (lambda: eng.gen(sch, a.n, fss_amd.U64, s0s, al, betas)) if a.key_major else \
         (lambda: eng.gen_level_major(sch, a.n, fss_amd.U64, s0s, al, betas))
    w, k = time_steps(fn, 3, 1, None)
    print(f"{a.prg} {name} gen {'key' if a.key_major else 'level'}-major n={a.n} 2^{a.log2_keys} keys: {k:.2f} ms {nk/k/1e6:.3f} G keys/s")
