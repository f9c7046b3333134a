#!/usr/bin/env python3
"""Wall time of one fss_crypto.Dpf.eval call (the reference binding's per-point API), tensors already on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_crypto
for prg in ("chacha", "aes128_mmo"):
    for n in (20, 32):
        d = fss_crypto.Dpf(n, "uint", prg)
        s0s = torch.randint(-2**31, 2**31, (2, 4), dtype=torch.int32)
        cws = d.gen(s0s, 12345, torch.tensor([7, 0, 0, 0], dtype=torch.int32))
        s0, cw = s0s[0].cuda(), cws.cuda()
        for _ in range(20):
            d.eval(0, s0, cw, 999)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(300):
            y = d.eval(0, s0, cw, 1000 + i)
        torch.cuda.synchronize()
        print(f"fss_crypto.Dpf({n}, uint, {prg}).eval: {(time.perf_counter() - t) / 300 * 1e6:.0f} us per call (wall)")
