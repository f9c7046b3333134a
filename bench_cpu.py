"""bench.py's CPU legs: the oracle (oracle/: the restatement of the reference's CPU path, OpenSSL EVP PRG in the
reference's call pattern) timed on bounded samples of the GPU workloads on the host cores, and used as the checker of
the GPU outputs of the timed launches.  Runs after the timed GPU regions; never part of them."""
import os
import time

import numpy as np
import torch

from bench_meta import _u64, make_key

def _timed_cpu(fn, min_s, max_reps=64):
    """fn() repeated until min_s seconds have been spent: (repetitions, seconds)."""
    reps, dt = 0, 0.0
    while dt < min_s and reps < max_reps:
        t0 = time.perf_counter()
        fn()
        dt += time.perf_counter() - t0
        reps += 1
    return reps, dt


def usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs under a 16-CPU quota: 256 threads there run at 8.5 M evals/s, 16 threads at 19.7)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


# ---------------------------------------------------------------------------------------------------------- CPU legs
def cpu_baseline(scheme, n, s0s, cws, ocw, xs_sample, gpu_sample, threads):
    """Oracle = restatement of the reference CPU path with the OpenSSL EVP PRG, timed on a
    bounded sample of the workload's own points; also the parity gate for this run."""
    from oracle import oracle as O
    mul = {"dpf": 2, "dcf": 4, "ht": 1}[scheme]
    sid = {"dpf": O.DPF, "dcf": O.DCF, "ht": O.HT}[scheme]
    p = O.Prg(mul, use_evp=True)
    xs64 = xs_sample.astype(np.uint64)
    oc = None if ocw is None else ocw.numpy()
    O.eval_points(p, sid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), xs64[:1024], ocw=oc, nthreads=1)  # warm
    # the all-threads pass repeated until >= 4 s, a ~1 s single-thread sample, one AES-NI pass (the sub-records'
    # CPU legs add about as much again: ~15 s of CPU work in the whole default run)
    reps, dt = 0, 0.0
    while dt < 4.0 and reps < 64:
        t0 = time.perf_counter()
        ys = O.eval_points(p, sid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), xs64, ocw=oc, nthreads=threads)
        dt += time.perf_counter() - t0
        reps += 1
    n1 = min(len(xs64), max(1024, 1 << 20))
    t1 = time.perf_counter()
    O.eval_points(p, sid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), xs64[:n1], ocw=oc, nthreads=1)
    dt1 = time.perf_counter() - t1
    out = {"value": reps * len(xs64) / dt, "unit": "evals/s", "cores": threads, "kind": "port",
           "sample": f"first {len(xs64)} points of the timed batch x {reps} passes, {dt:.1f} s; t1: {n1} points",
           "t1": n1 / dt1, "ok": bool((ys == gpu_sample).all())}
    if O.have_aesni():
        # "best CPU" line: the same walk with AES-NI intrinsics, as the reference's Aes128MmoRaw (README.md:341)
        pn = O.Prg(mul, aesni=True)
        t2 = time.perf_counter()
        ysn = O.eval_points(pn, sid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), xs64, ocw=oc, nthreads=threads)
        out["aesni"] = len(xs64) / (time.perf_counter() - t2)
        out["ok"] = bool(out["ok"] and (ysn == ys).all())
    return out


def cpu_check_subrecords(checks):
    """The sub-records' parity gates, run with the cpu_baseline leg: a strided sample of every sub-workload's GPU
    outputs against the oracle (party 0, Uint<uint64_t>)."""
    from oracle import oracle as O
    res = {}
    for key, (kind, sch, n, s0s, cws, xs, got) in checks.items():
        sid, mul = {"dpf": (O.DPF, 2), "dcf": (O.DCF, 4)}[sch]
        if kind == "keys":
            want = O.eval_keys(O.Prg(mul), sid, n, O.U64, 0, s0s, cws, _u64(xs), nthreads=4)
        else:
            want = O.eval_points(O.Prg(mul), sid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), _u64(xs), nthreads=4)
        res[key] = bool((want == got).all())
    return res


def cpu_evalall(eng, threads, n_cpu):
    """CPU baseline of the full-domain workload: the oracle's EvalAll in the reference's own
    execution shape -- depth-first, one OpenMP task per child above ceil(log2(threads)) levels (dpf.cuh:232-303,
    util.cuh:40-45), OpenSSL EVP PRG -- on a 2^n_cpu-leaf domain, all threads and one thread; the GPU's EvalAll of
    the same key is compared with it leaf for leaf."""
    import fss_amd
    from oracle import oracle as O
    G = fss_amd.U64
    sid, s0s, cws, _, _, _ = make_key(eng, "dpf", n_cpu, G, 0x5EED0400 + n_cpu)
    got = eng.eval_all(sid, n_cpu, G, 0, s0s[0].cuda(), cws.cuda()).cpu().numpy()
    p = O.Prg(2, use_evp=True)
    out = np.zeros((1 << n_cpu, 4), dtype=np.int32)
    s0, cw = s0s[0].numpy(), cws.numpy()
    reps, dt = _timed_cpu(lambda: O.dpf_eval_all(p, n_cpu, O.U64, 0, s0, cw, task_threads=threads, out=out), 1.5)
    parity = bool((out == got).all())
    t1 = time.perf_counter()
    O.dpf_eval_all(p, n_cpu, O.U64, 0, s0, cw, task_threads=1, out=out)
    dt1 = time.perf_counter() - t1
    return {"cpu": reps * (1 << n_cpu) / dt, "cpu1": (1 << n_cpu) / dt1, "ok": parity and bool((out == got).all()),
            "cpu_sample": f"DPF EvalAll kInBits={n_cpu} x {reps} passes, {threads} threads, {dt:.1f} s; one thread: one pass, {dt1:.1f} s"}


def cpu_keys(eng, scheme, n, keys, seeds0, xs, ocws, ys, threads, m):
    """CPU baseline of the independent-keys workloads (C5): a strided subset of m keys brought back key-major, the
    oracle's per-key Eval (Half-Tree: Aes128Mmo<1> hashes; Grotto: the path walk, both children per level as the
    reference's ExpandTreeRec computes them) with the OpenSSL EVP PRG, all threads and one thread; the GPU's outputs
    for those keys are compared with it."""
    import fss_amd
    from fss_amd.engine import LevelMajorKeys
    from oracle import oracle as O
    nk = keys.nkeys
    m = min(m, nk)
    pick = torch.arange(0, nk, max(1, nk // m), device="cuda")[:m]
    m = int(pick.numel())
    if scheme == "ht":
        sub = LevelMajorKeys(fss_amd.HT, n, m, keys.cw_s[:, pick].contiguous(), extra=keys.extra[pick].contiguous())
        osid, mul, oc = O.HT, 1, ocws[pick].cpu().numpy()
    else:
        sub = LevelMajorKeys(fss_amd.DPF, n, m, keys.cw_s[:, pick].contiguous(), tr_bits=keys.tr_bits[pick].contiguous(),
                             out_cw=keys.out_cw[pick].contiguous())
        osid, mul, oc = O.GROTTO, 2, None
    km = eng.key_major(sub).cpu().numpy()
    sd = seeds0[pick].cpu().numpy()
    xh = xs[pick].cpu().numpy().view(np.uint64)
    got = ys[pick].cpu().numpy()
    p = O.Prg(mul, use_evp=True)
    res = {}

    def all_threads():
        res["y"] = O.eval_keys(p, osid, n, O.U64 if scheme == "ht" else O.BYTES, 0, sd, km, xh, ocws=oc, nthreads=threads)
    reps, dt = _timed_cpu(all_threads, 1.5)
    m1 = min(m, 1 << 16)
    t1 = time.perf_counter()
    O.eval_keys(p, osid, n, O.U64 if scheme == "ht" else O.BYTES, 0, sd[:m1], km[:m1], xh[:m1],
                ocws=None if oc is None else oc[:m1], nthreads=1)
    dt1 = time.perf_counter() - t1
    want = res["y"] if scheme == "ht" else res["y"][:, 0].astype(np.uint8)
    return {"cpu": reps * m / dt, "cpu1": m1 / dt1, "ok": bool((want == got).all()),
            "cpu_sample": f"{m} strided keys x {reps} passes, {threads} threads, {dt:.1f} s; one thread: {m1} keys"}
