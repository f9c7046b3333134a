import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import make_key, time_steps
eng = fss_amd.default_engine(0)
G = fss_amd.U64
for n in (20, 22, 24, 26, 28):
    sid, s0s, cws, ocw, _, _ = make_key(eng, "dcf", n, G, 7 + n)
    sd, cd = s0s[0].cuda(), cws.cuda()
    ys = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
    fn = lambda: eng.eval_all(sid, n, G, 0, sd, cd, out=ys)
    _, k = time_steps(fn, 10, 3, None)
    blocks = 4.0 * (1 << n)
    print(f"dcf EvalAll n={n}: {k:.4f} ms  lds_frac {blocks * 640 / (k * 1e-3) / 78.643e12:.4f}  {eng.last_kernel()}  sum={int(ys[::4097].sum())}")
    del ys
