"""Socket power and shader clock while each of the walks runs back to back (hwmon: power1_input, freq1_input of this
process's GPU, found by its PCI address), beside the kernel time.  Backs DESIGN 4.8's statement that the
independent-key walks run at the clock the power budget grants.

    PYTHONPATH=. python tools/power_probe.py [seconds per workload] [substring of the workload names to run]
"""
import glob
import os
import sys
import threading
import time

import torch

import fss_amd
from fss_amd.engine import DPF, DCF, HT

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import groups as G  # noqa: E402


def my_hwmon():
    p = torch.cuda.get_device_properties(0)
    want = None
    if hasattr(p, "pci_bus_id"):
        want = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}.0"
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        bdf = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
        cands.append((bdf, d))
    for bdf, d in cands:
        if want and bdf.lower() == want.lower():
            return d, bdf, cands
    return None, want, cands


def read_int(path):
    try:
        return int(open(path).read())
    except (OSError, ValueError):
        return -1


class Sampler(threading.Thread):
    def __init__(self, dirs, period=0.02):
        super().__init__(daemon=True)
        self.dirs, self.period, self.rows, self.stop = dirs, period, [], False

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            self.rows.append((t, [(read_int(d + "/power1_input"), read_int(d + "/freq1_input")) for d in self.dirs]))
            time.sleep(self.period)


def run_for(fn, seconds):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    k = 0
    e0.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        k += 8
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    eng = fss_amd.default_engine(0)
    mine, bdf, cands = my_hwmon()
    dirs = [d for _, d in cands]
    print(f"gpu {bdf}  hwmon {mine}  cap {read_int((mine or dirs[0]) + '/power1_cap') / 1e6:.0f} W", flush=True)
    gen = torch.Generator(device="cuda").manual_seed(7)
    N = 1 << 24
    ri = lambda *shape: torch.randint(-2**31, 2**31, shape, dtype=torch.int32, device="cuda", generator=gen)
    r64 = lambda n: torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=gen)
    work = []
    # one key, 2^24 points
    s0s, al, be = ri(1, 2, 4), ri(1), ri(1, 4)
    cws, _ = eng.gen(DPF, 32, G.U64, s0s, al, be)
    xs, ys = ri(N), torch.empty((N, 4), dtype=torch.int32, device="cuda")
    seed = s0s[0, 0].contiguous()
    work.append(("dpf n=32, one key x 2^24 points", lambda: eng.eval_points(DPF, 32, G.U64, 0, seed, cws[0], xs, out=ys)))
    cwd, _ = eng.gen(DCF, 32, G.U64, s0s, al, be)
    work.append(("dcf n=32, one key x 2^24 points", lambda: eng.eval_points(DCF, 32, G.U64, 0, seed, cwd[0], xs, out=ys)))
    # 2^24 independent keys
    ks0, kal, kbe = ri(N, 2, 4), ri(N), ri(N, 4)
    keys, _ = eng.gen_level_major(DPF, 32, G.U64, ks0, kal, kbe)
    kseeds = ks0[:, 0].contiguous()
    work.append(("dpf n=32, 2^24 independent keys", lambda: eng.eval_keys(keys, G.U64, 0, kseeds, xs, out=ys)))
    x64 = r64(N)
    hkeys, hocw = eng.gen_level_major(HT, 64, G.U64, ks0, r64(N), kbe)
    work.append(("half-tree n=64, 2^24 independent keys", lambda: eng.eval_keys(hkeys, G.U64, 0, kseeds, x64, ocws=hocw, out=ys)))
    # full domain, n = 28
    c28, _ = eng.gen(DPF, 28, G.U64, s0s, al & ((1 << 28) - 1), be)
    y28 = torch.empty((1 << 28, 4), dtype=torch.int32, device="cuda")
    work.append(("dpf EvalAll n=28", lambda: eng.eval_all(DPF, 28, G.U64, 0, seed, c28[0], out=y28)))
    # a plain copy at HBM rate, for scale
    src = torch.empty((1 << 30,), dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    work.append(("1 GiB device copy", lambda: dst.copy_(src)))

    work = [w for w in work if only in w[0]]
    idx = dirs.index(mine) if mine in dirs else None
    print(f"{'workload':42s} {'ms':>8s} {'W':>7s} {'W max':>7s} {'MHz':>6s}  samples", flush=True)
    time.sleep(1.0)
    s = Sampler(dirs)
    s.start()
    time.sleep(0.5)
    marks = [("idle", time.perf_counter(), None)]
    time.sleep(0.5)
    for name, fn in work:
        t0 = time.perf_counter()
        ms = run_for(fn, secs)
        marks.append((name, t0, (time.perf_counter(), ms)))
        time.sleep(0.3)
    s.stop = True
    s.join()
    if idx is None:  # the card whose power moved most is ours
        spans = [max(r[1][i][0] for r in s.rows) - min(r[1][i][0] for r in s.rows) for i in range(len(dirs))]
        idx = spans.index(max(spans))
        print(f"(no PCI match; taking {cands[idx][0]}, the card whose power moved)")
    for name, t0, rest in marks:
        if rest is None:
            rows = [r for r in s.rows if r[0] < t0 + 0.4]
            ms = float("nan")
        else:
            t1, ms = rest
            rows = [r for r in s.rows if t0 + (t1 - t0) / 3 <= r[0] <= t1]  # the last two thirds: clocks have settled
        pw = [r[1][idx][0] / 1e6 for r in rows if r[1][idx][0] >= 0]
        fq = [r[1][idx][1] / 1e6 for r in rows if r[1][idx][1] >= 0]
        if not pw:
            print(f"{name:42s} {ms:8.3f}   no samples")
            continue
        print(f"{name:42s} {ms:8.3f} {sum(pw) / len(pw):7.0f} {max(pw):7.0f} {sum(fq) / max(len(fq), 1):6.0f}  {len(pw)}",
              flush=True)


if __name__ == "__main__":
    main()
