#!/usr/bin/env python3
"""Full-domain evaluation (fss_amd_*_eval_all, the whole 2^n domain of one key) per scheme, output group and size: ms per
call and the fraction of the LDS roof on the blocks a leaf costs (DPF 2, Half-Tree 1.5, DCF 4), with a checksum of the
outputs so that two builds of the library (FSS_AMD_LIB) can be compared on one lease.  --loop N: nothing but N calls of
one configuration (the program to put behind `rocprofv3 --pmc ... --`).
usage: evalall_sizes.py [--schemes dcf,dpf,ht] [--groups U64,BYTES,U128] [--n 20,22,24,26,28] [--loop 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fss_amd  # noqa: E402
from bench import make_key, time_steps  # noqa: E402

BLOCKS = {"dpf": 2.0, "ht": 1.5, "dcf": 4.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--schemes", default="dcf")
    ap.add_argument("--groups", default="U64")
    ap.add_argument("--n", default="20,22,24,26,28")
    ap.add_argument("--loop", type=int, default=0)
    a = ap.parse_args()
    eng = fss_amd.default_engine(0)
    for gname in a.groups.split(","):
        G = getattr(fss_amd, gname)
        for scheme in a.schemes.split(","):
            for n in [int(v) for v in a.n.split(",")]:
                sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, G, 7 + n)
                sd, cd = s0s[0].cuda(), cws.cuda()
                od = None if ocw is None else ocw.cuda()
                ys = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
                fn = lambda: eng.eval_all(sid, n, G, 0, sd, cd, ocw=od, out=ys)  # noqa: E731
                if a.loop:
                    for _ in range(a.loop):
                        fn()
                    torch.cuda.synchronize()
                    continue
                _, k = time_steps(fn, 10, 3, None)
                frac = BLOCKS[scheme] * (1 << n) * 640 / (k * 1e-3) / 78.643e12
                print(f"{scheme} EvalAll n={n} {gname}: {k:.4f} ms  lds_frac {frac:.4f}  {eng.last_kernel()}  sum={int(ys[::4097].sum())}")
                del ys


if __name__ == "__main__":
    main()
