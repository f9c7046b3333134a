#!/usr/bin/env python3
"""Fixed per-launch cost of the point walk: DPF with very few levels, event-timed back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import make_key, time_steps
eng = fss_amd.default_engine(0)
for lp in (10, 16, 20):
    for n in (1, 2, 4, 8):
        sid, s0s, cws, ocw, _, _ = make_key(eng, "dpf", n, fss_amd.U64, 3 + n)
        batch = 1 << lp
        xs = torch.randint(0, 1 << n, (batch,), dtype=torch.int32, device="cuda")
        sd, cd = s0s[0].cuda(), cws.cuda()
        ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")
        w, k = time_steps(lambda: eng.eval_points(sid, n, fss_amd.U64, 0, sd, cd, xs, out=ys), 50, 5, None)
        print(f"2^{lp} n={n}: {k*1000:.1f} us per launch (event-timed, back-to-back)")
