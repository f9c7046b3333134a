#!/usr/bin/env python3
"""Shared-prefix point walk (fss_amd_*_eval_points_shared): time per call against prefix_bits, split into the
expansion of the node table (timed alone through a batch just past the quad-walk threshold, whose walk is negligible)
and the walk.  Outputs are compared with the plain walk's for every row.
usage: shared_prefix_sweep.py [--schemes dpf,dcf] [--log2-points 24,20] [--kmin 16 --kmax 26]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import fss_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--schemes", default="dpf,dcf")
    ap.add_argument("--log2-points", default="24,20")
    ap.add_argument("--in-bits", type=int, default=32)
    ap.add_argument("--kmin", type=int, default=16)
    ap.add_argument("--kmax", type=int, default=26)
    ap.add_argument("--auto-k", action="store_true", help="only the prefix_bits the library picks by itself (fss_amd_points_prefix_bits)")
    ap.add_argument("--sorted", action="store_true", help="evaluate the points in ascending order (the node gathers then walk the table in order: "
                                                          "the locality upper bound of the walk)")
    args = ap.parse_args()
    eng = fss_amd.default_engine(0).set_points_plan("plain")  # eval_points = the plain walk; the plan only through _shared
    G, n = fss_amd.U64, args.in_bits
    print(f"lib={os.environ.get('FSS_AMD_LIB', 'default')} sorted={args.sorted}")
    for sch in args.schemes.split(","):
        sid, s0s, cws, ocw, _, _ = bench.make_key(eng, sch, n, G, 0x5EED0002)
        sd, cd = s0s[0].cuda(), cws.cuda()
        od = None if ocw is None else ocw.cuda()
        for lp in [int(v) for v in args.log2_points.split(",")]:
            N = 1 << lp
            xs = bench._rand_xs(n, N, 0x5EED0002)
            if args.sorted:  # ascending as unsigned values (the int32 / int64 bit patterns sort with the sign bit flipped)
                bias = (1 << 31) if n <= 32 else (1 << 63)
                xs = ((xs ^ -bias).sort().values ^ -bias).contiguous()
            ys, ref = torch.empty((N, 4), dtype=torch.int32, device="cuda"), torch.empty((N, 4), dtype=torch.int32, device="cuda")
            eng.eval_points(sid, n, G, 0, sd, cd, xs, ocw=od, out=ref)
            _, plain = bench.time_steady(lambda: eng.eval_points(sid, n, G, 0, sd, cd, xs, ocw=od, out=ys))
            aes = bench.aes_blocks_per_eval(sch, n)
            print(f"{sch} n={n} 2^{lp} points: plain walk {plain:.4f} ms = {N / plain / 1e6:.3f} G evals/s; "
                  f"auto prefix_bits = {eng._L.fss_amd_points_prefix_bits(sid, n, N)}")
            small = xs[: (1 << 17) + 64].contiguous()
            ysm = torch.empty((small.numel(), 4), dtype=torch.int32, device="cuda")
            auto = eng._L.fss_amd_points_prefix_bits(sid, n, N)
            for k in ([auto] if args.auto_k and auto else range(args.kmin, min(args.kmax, n - (1 if sch == "ht" else 0), 26) + 1)):
                ys.zero_()
                step = lambda: eng.eval_points_shared(sid, n, G, 0, sd, cd, xs, ocw=od, out=ys, prefix_bits=k)
                _, tot = bench.time_steady(step)
                same = bool((ys == ref).all())
                _, exp = bench.time_steady(lambda: eng.eval_points_shared(sid, n, G, 0, sd, cd, small, ocw=od, out=ysm, prefix_bits=k))
                per = aes / n
                blocks = N * (n - k) * per + (2 << k) * per * (1 if sch != "ht" else 0.75)
                fx = blocks * 640 / (tot * 1e-3) / 1e9 / bench.LDS_PEAK_GBS
                print(f"  k={k:2d} total {tot:.4f} ms  expand~{exp:.4f}  walk~{tot - exp:.4f}  {N / tot / 1e6:7.3f} G evals/s  "
                      f"x{plain / tot:.2f}  fx={fx:.3f}  same={same}", flush=True)


if __name__ == "__main__":
    main()
