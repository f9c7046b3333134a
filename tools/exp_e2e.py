#!/usr/bin/env python3
"""Host-memory path of the headline batch (x in pinned host memory, y back to pinned host memory), four ways:
  serial      H2D, walk, D2H on one stream
  rr3         8 chunks round-robin over three streams (round 3's bench.py form)
  lanes       K chunks over three dedicated in-order queues (H2D | walk | D2H) chained by events
  zero-copy   ONE launch of the walk reading x from and writing y to the pinned buffers themselves (mapped host memory)
usage: tools/exp_e2e.py [--log2-points 24] [--only NAME]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import make_key
ap = argparse.ArgumentParser()
ap.add_argument("--log2-points", type=int, default=24); ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--only", default="")
a = ap.parse_args()
eng = fss_amd.default_engine(0)
n, N, G = 32, 1 << a.log2_points, fss_amd.U64
sid, s0s, cws, _, _, _ = make_key(eng, "dpf", n, G, 35)
sd, cd = s0s[0].cuda(), cws.cuda()
xs_h = torch.randint(-2**31, 2**31 - 1, (N,), dtype=torch.int32).pin_memory()
ys_h = torch.empty((N, 4), dtype=torch.int32).pin_memory()
xs_d = torch.empty((N,), dtype=torch.int32, device="cuda")
ys_d = torch.empty((N, 4), dtype=torch.int32, device="cuda")
xs_d.copy_(xs_h); ref = eng.eval_points(sid, n, G, 0, sd, cd, xs_d).clone(); torch.cuda.synchronize()


def raw(xp, yp, cnt, stream):
    rc = eng._L.fss_amd_dpf_eval_points(eng._h, n, G, 0, C.c_void_p(sd.data_ptr()), C.c_void_p(cd.data_ptr()), C.c_void_p(xp), 4, cnt,
                                        C.c_void_p(yp), C.c_void_p(stream.cuda_stream))
    assert rc == 0, rc


def serial():
    xs_d.copy_(xs_h, non_blocking=True)
    eng.eval_points(sid, n, G, 0, sd, cd, xs_d, out=ys_d)
    ys_h.copy_(ys_d, non_blocking=True)


rr = [torch.cuda.Stream() for _ in range(3)]
def rr3(chunks=8):
    Cn = N // chunks
    for c in range(chunks):
        sl = slice(c * Cn, (c + 1) * Cn)
        with torch.cuda.stream(rr[c % 3]):
            xs_d[sl].copy_(xs_h[sl], non_blocking=True)
            eng.eval_points(sid, n, G, 0, sd, cd, xs_d[sl], out=ys_d[sl])
            ys_h[sl].copy_(ys_d[sl], non_blocking=True)


s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def lanes(chunks):
    Cn = N // chunks
    for c in range(chunks):
        sl = slice(c * Cn, (c + 1) * Cn)
        e_in, e_k = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(s_in):
            xs_d[sl].copy_(xs_h[sl], non_blocking=True); e_in.record()
        with torch.cuda.stream(s_k):
            s_k.wait_event(e_in)
            eng.eval_points(sid, n, G, 0, sd, cd, xs_d[sl], out=ys_d[sl]); e_k.record()
        with torch.cuda.stream(s_out):
            s_out.wait_event(e_k)
            ys_h[sl].copy_(ys_d[sl], non_blocking=True)


def zero_copy():
    raw(xs_h.data_ptr(), ys_h.data_ptr(), N, torch.cuda.current_stream())

def zc_x_only():   # x read from host by the kernel, y to device, then one D2H
    raw(xs_h.data_ptr(), ys_d.data_ptr(), N, torch.cuda.current_stream())
    ys_h.copy_(ys_d, non_blocking=True)

def zc_y_only():   # H2D of x, then the kernel writes y to host
    xs_d.copy_(xs_h, non_blocking=True)
    raw(xs_d.data_ptr(), ys_h.data_ptr(), N, torch.cuda.current_stream())

def zc_chunks(chunks=4):   # zero-copy launches on two streams (does a second launch hide the first one's store drain?)
    Cn = N // chunks
    for c in range(chunks):
        with torch.cuda.stream(rr[c % 2]):
            raw(xs_h.data_ptr() + 4 * c * Cn, ys_h.data_ptr() + 16 * c * Cn, Cn, rr[c % 2])

variants = [("serial", serial), ("rr3 x8", rr3), ("lanes x8", lambda: lanes(8)), ("lanes x16", lambda: lanes(16)), ("lanes x32", lambda: lanes(32)),
            ("zero-copy", zero_copy), ("zc x only", zc_x_only), ("zc y only", zc_y_only), ("zc 4 chunks 2 streams", zc_chunks)]
for name, fn in variants:
    if a.only and a.only != name:
        continue
    ys_h.zero_()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    ok = bool((ys_h.cuda() == ref).all())
    print(f"{name:24s} {dt*1e3:8.3f} ms per 2^{a.log2_points} points = {N/dt/1e9:6.3f} G evals/s  ({N*20/dt/1e9:5.1f} GB/s over PCIe)  identical={ok}", flush=True)
