import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import make_key, time_steps
eng = fss_amd.default_engine(0)
for gname, G in (("U64", fss_amd.U64), ("BYTES", fss_amd.BYTES), ("U128", fss_amd.U128)):
    for scheme, bl in (("dpf", 2.0), ("ht", 1.5), ("dcf", 4.0)):
        for n in (24, 26):
            sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, G, 7 + n)
            sd, cd = s0s[0].cuda(), cws.cuda()
            od = None if ocw is None else ocw.cuda()
            ys = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
            fn = lambda: eng.eval_all(sid, n, G, 0, sd, cd, ocw=od, out=ys)
            _, k = time_steps(fn, 10, 3, None)
            print(f"{scheme} EvalAll n={n} {gname}: {k:.4f} ms  lds_frac {bl * (1 << n) * 640 / (k * 1e-3) / 78.643e12:.4f}  {eng.last_kernel()}")
            del ys
