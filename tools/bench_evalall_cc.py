#!/usr/bin/env python3
"""ChaCha EvalAll timing for a few domain sizes (planner experiments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import make_key, time_steady
eng = fss_amd.Engine(0, prg="chacha")
for scheme in sys.argv[1:] or ["dpf"]:
    for n in (16, 18, 20, 22, 24):
        sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, fss_amd.U64, 7 + n)
        sd, cd = s0s[0].cuda(), cws.cuda()
        od = None if ocw is None else ocw.cuda()
        ys = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
        _, kms = time_steady(lambda: eng.eval_all(sid, n, fss_amd.U64, 0, sd, cd, ocw=od, out=ys))
        print(f"chacha {scheme} EvalAll n={n}: {kms*1e3:.1f} us  {(1<<n)/kms/1e6:.2f} G leaves/s")
