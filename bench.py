#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DPF/DCF evaluation engine.

Metric (BASELINE.json): DPF/DCF evals/sec at kInBits=32, batch=2^24.
A "step" is one pass of the hot path over one batch: ONE launch of the shared-key DPF point-evaluation kernel over
2^24 uniformly random points (configs[1]'s shape at the batch the metric is quoted on), party 0, Uint<uint64_t>
outputs, inputs already resident in HBM.  With --gpus N the metric's ONE 2^24-point batch is cut into N slices, one per
rank, no collective on the data path (`value`, "scaling": "strong": SURVEY 8(e), "batch=2^24 on 1/2/4/8"; the same record is
printed as `headline_strong`); what every rank does with a 2^24-point batch of its own is the side record `weak`.  With
one GPU the two are the same launch.  Timed regions are bracketed by barrier + synchronize and the slowest rank's time is
used; `config.ranks` lists what every rank ran on (device UUID / PCI address, process-group backend and size as that rank
sees them) and `config.distinct_devices` how many different GPUs that is.

`python bench.py --gpus N` with no launcher around it starts its own N ranks (one process per GPU, spawned before
anything touches the GPU; a failed or hung rank ends the run non-zero); under `python -m torch.distributed.run` (RANK /
WORLD_SIZE set) it is one of the ranks.

Rank 0 prints ONE JSON line, kept under 7 KB so that every number of it survives in the driver's record: the contract
keys, the three roofs of the headline kernel, the CPU baseline, and under `sub` one short record per remaining cell of
the metric (DCF, independent keys, kInBits 16 / 64, BASELINE configs[0..4], the PCIe-inclusive rate).  The short keys
and every record's workload are spelled out in bench_legend.json (`python bench.py --write-legend` regenerates it from
the tables below; tests/test_bench_meta.py keeps the two in step); the unabridged records of a run go to
gpurun_out/bench_detail.json.  `roofline` is the binding roof -- LDS table-lookup bandwidth (the path is
neither HBM- nor MFMA-bound); `cpu_baseline` times the oracle's restatement of the reference CPU path (OpenSSL EVP
AES-128-ECB, one EVP_EncryptUpdate per block, both children computed -- include/fss/prg/aes128_mmo.cuh:72-93,
include/fss/dpf.cuh:170-214) on a bounded sample of the same points and checks the GPU output against it.

--suite additionally runs tools/bench_suite.py (every other kernel of the library on one GPU; not part of the contract).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_meta import (CLK_HZ, HBM_PEAK_GBS, KEYS, LDS_PEAK_GBS, LOOKUP_BYTES_PER_AES, METRIC, N_CU, RECORDS, SHORT, TOP,  # noqa: E402,F401
                        VALU_ISSUE_PEAK_GINSTR, _headline_pmc, _rand_xs, _u64, aes_blocks_per_eval, compact, hbm_bytes_per_eval, legend, make_key,
                        measured_hbm_traffic, measured_pipe_busy, r5, replayed_counters)
from bench_cpu import _timed_cpu, cpu_baseline, cpu_check_subrecords, cpu_evalall, cpu_keys, usable_cores  # noqa: E402,F401
from bench_launch import launch_ranks, run_dry, visible_gpus_without_hip  # noqa: E402,F401


# ---------------------------------------------------------------------------------------------------------- timing
class Timing(tuple):
    """(wall seconds, mean kernel ms) of a timed region, with `.median`: the median of the K per-launch kernel times (ms)
    when the region recorded an event between its steps, else None."""
    median = None


PER_STEP_EVENTS_MAX = 256  # regions of at most this many steps record one event per step (the median); longer ones two


def time_steps(fn, steps, warmup, dist, per_step=False):
    """W warm-up steps, then exactly K timed steps between barrier+synchronize pairs.
    Returns Timing(wall seconds max over ranks, mean kernel ms from HIP events on the launch stream); with per_step the K
    launches are separated by events on the same stream and `.median` is the median launch ("median of >= 10
    launches", as src/bench_gpu.cu:48-68 times; the mean stays what ms_per_step and every rate are computed from)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    per_step = per_step and 0 < steps <= PER_STEP_EVENTS_MAX
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1 if per_step else 2)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        fn()
        if per_step:
            evs[i + 1].record()
    if not per_step:
        evs[1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = evs[0].elapsed_time(evs[-1]) / max(steps, 1)
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    out = Timing((wall, kernel_ms))
    if per_step:
        out.median = float(np.median([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]))
    return out


def time_steady(fn, steps=10, min_ms=150.0, warm_ms=60.0):
    """Kernel-only rate of a short launch at steady clocks: the chip needs tens of milliseconds of continuous work
    to reach its sustained clock (a 0.3 ms kernel timed over 10 launches runs 15 % below what it sustains), so warm
    up for `warm_ms` of device time and time at least `min_ms` of back-to-back launches (>= `steps`)."""
    _, est = time_steps(fn, 3, 1, None)
    est = max(est, 1e-3)
    warm = min(5000, max(2, int(warm_ms / est) + 1))
    k = min(20000, max(steps, int(min_ms / est) + 1))
    return time_steps(fn, k, warm, None)


def pre_warm(fn, ms=100.0):
    """Sub-records are timed between CPU legs during which the GPU sits idle and its clocks fall: run `ms` of the
    step first (untimed), then the W warm-up and K timed steps follow as for the headline.  The number of repetitions
    depends on this rank's own timing: `fn` must not contain a collective."""
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    est = max(time.perf_counter() - t0, 1e-5)
    for _ in range(min(2000, int(ms * 1e-3 / est))):
        fn()
    torch.cuda.synchronize()


_BENCH_ENGINE = {}


def bench_engine():
    """bench.py's own engine (not the process-wide default one) with FSS_AMD_POINTS_PLAIN: every cell of the metric
    measures the per-point tree walk (in_bits PRG levels per point, north_star); the shared-prefix plan is reported in
    its own records (`shared_*`), through its explicit entry point."""
    import fss_amd
    dev = torch.cuda.current_device()
    if dev not in _BENCH_ENGINE:
        _BENCH_ENGINE[dev] = fss_amd.Engine(dev).set_points_plan("plain")
    return _BENCH_ENGINE[dev]


_DIAG = {}
NO_CLK = False  # --no-clk: skip the in-kernel clock runs (the per-workload profiling passes: one kernel object per name)


def diag_engine():
    """A context of the diagnostic build (fss_amd/libfss_amd_diag.so, -DFSS_DIAG_STAMPS=1), or None without it."""
    if NO_CLK:
        return None
    if "eng" not in _DIAG:
        import fss_amd
        from fss_amd import _lib
        L = _lib.diag_lib()
        _DIAG["eng"] = None if L is None else fss_amd.Engine(torch.cuda.current_device(), lib=L).set_points_plan("plain")
    return _DIAG["eng"]


def in_kernel_clock(mk_step, ms=50.0):
    """Shader clock (GHz) the workload's kernel runs at, measured INSIDE the kernel: `ms` of back-to-back launches of
    mk_step(engine) through the diagnostic build, whose waves record s_memtime (shader clock) and s_memrealtime
    (100 MHz) at entry and exit; median over the waves of the last launch of d(memtime) / d(memrealtime) x 100 MHz.
    Untimed (after the timed region of the record it annotates); None when the diagnostic library is not built."""
    de = diag_engine()
    if de is None:
        return None
    import ctypes as C
    step = mk_step(de)
    pre_warm(step, ms)
    step()
    torch.cuda.synchronize()
    kname, grid, block = de.last_kernel()
    nw = grid * (block // 64)
    fused = kname == "evalall_fused"
    buf = np.zeros((nw, 16 if fused else 4), dtype=np.uint64)
    if de._L.fss_amd_diag_read_stamps(de._h, buf.ctypes.data_as(C.c_void_p), nw * (4 if fused else 1)) != 0:
        return None
    t = buf.astype(np.float64)
    dcyc, dreal = (t[:, 15] - t[:, 14], t[:, 13] - t[:, 0]) if fused else (t[:, 2] - t[:, 0], t[:, 3] - t[:, 1])
    ok = (dreal > 0) & (dcyc > 0)
    return float(np.median(dcyc[ok] / dreal[ok] * 0.1)) if ok.any() else None


def _attach_clock(rec, mk_step):
    """`clk` and `fc` of a record: the in-kernel clock and the roofline fraction at THAT clock instead of the nominal
    2.4 GHz the peak is computed for (what the kernel achieves per cycle, independent of the box's power-capped clock)."""
    clk = in_kernel_clock(mk_step)
    if clk:
        rec["clk"] = clk
        rec["frac_at_clock"] = rec.get("frac_executed", rec["frac"]) * CLK_HZ / (clk * 1e9)
    return rec


def rank_identities(rank, world, dist):
    """`config.ranks`: what every rank ran on, all-gathered -- (rank, local device index, the device's UUID / PCI address as
    the runtime reports them, the process group's backend and size as THIS rank sees them) -- so that a reader of the one
    line can see N distinct GPUs and a process group of N ranks behind an `n_gpus: N` line.  Not timed."""
    i = torch.cuda.current_device()
    pr = torch.cuda.get_device_properties(i)
    ident = {"rank": rank, "pid": os.getpid(), "dev": i, "name": pr.name}
    if getattr(pr, "uuid", None) is not None:
        ident["uuid"] = str(pr.uuid)
    pci = [getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    if all(v is not None for v in pci):
        ident["pci"] = "%04x:%02x:%02x" % tuple(int(v) for v in pci)
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if vis:
        ident["visible"] = vis
    ident["pg"] = None if dist is None else [dist.get_backend(), dist.get_world_size()]
    if dist is None:
        return [ident]
    out = [None] * world
    dist.all_gather_object(out, ident)
    return out


def distinct_devices(ranks):
    """How many different physical devices `config.ranks` names (by UUID when the runtime gives one, else by PCI address)."""
    key = lambda r: r.get("uuid") or (r.get("pci"), r.get("visible"), r["dev"])
    return len({key(r) for r in ranks})


def _walk_record(scheme, n, shared, units, wall, kernel_ms, world, steps, kname):
    """A point-walk workload timed like the headline (same barrier / max-over-ranks protocol)."""
    rate = units / (kernel_ms * 1e-3)
    aes = aes_blocks_per_eval(scheme, n)
    lds = rate * aes * LOOKUP_BYTES_PER_AES / 1e9
    hbm = rate * hbm_bytes_per_eval(scheme, n, shared) / 1e9
    return {"value": units * world * steps / wall, "unit": "evals/s", "ms_per_step": wall / steps * 1e3, "kernel_ms": kernel_ms,
            "units_per_launch": units, "aes_per_unit": aes, "frac": lds / LDS_PEAK_GBS, "lds_gbs": lds, "hbm_gbs": hbm,
            "hbm_frac": hbm / HBM_PEAK_GBS, "kernel": kname}


def _attach_replay(rec, name, units, alg_bytes_per_unit):
    rep = replayed_counters(name, units)
    rec["replay"] = rep["traffic_replayed_from"]
    rec["replay_ok"] = rep["replay_matches_kernel_sources"]
    rec["traffic"] = rep["traffic"]
    if rep["traffic"] is not None:
        rec["traffic_ratio"] = rep["traffic"] / (units * alg_bytes_per_unit)
    if "lds_busy_pmc" in rep:
        rec["lds_busy_pmc"] = rep["lds_busy_pmc"]


# ---------------------------------------------------------------------------------------------------------- GPU legs
def _shared_record(eng, scheme, sid, n, sd, cd, xs, ys_plain, plain_ms, timer=None, world=1, steps=1):
    """The same batch through the shared-prefix plan (fss_amd_*_eval_points_shared with the library's own prefix_bits --
    what fss_amd_*_eval_points picks by itself under FSS_AMD_POINTS_AUTO): the key's top k levels are expanded once per
    call (2^(k+1) blocks), every point walks the remaining in_bits - k.  `a` = AES blocks EXECUTED per evaluation and `f`
    the roofline fraction on those (never above 1); `x` = speed-up over the plain walk of the same batch; `same` =
    outputs bit-identical to it.  Work removed, never `value`.  timer: step -> (wall seconds, kernel ms)."""
    import fss_amd
    G, N = fss_amd.U64, xs.numel()
    k = eng._L.fss_amd_points_prefix_bits(sid, n, N) or max(1, N.bit_length() - 2)
    ys2 = torch.empty_like(ys_plain)
    mk = lambda e: (lambda: e.eval_points_shared(sid, n, G, 0, sd, cd, xs, out=ys2, prefix_bits=k))
    step = mk(eng)
    if timer:
        pre_warm(step, 40.0)  # like every sub-record: the first calls size the per-call workspace, the clocks settle
    wall, kms = timer(step) if timer else time_steady(step)
    aes = (2 if scheme == "dcf" else 1) * ((n - k) + (2 << k) / N)
    rate = N / (kms * 1e-3)
    rec = {"value": N * world * steps / wall if timer else rate, "unit": "evals/s", "ms_per_step": wall / steps * 1e3 if timer else kms,
           "kernel_ms": kms, "units_per_launch": N, "aes_per_unit": aes, "k": k,
           "frac": rate * aes * LOOKUP_BYTES_PER_AES / 1e9 / LDS_PEAK_GBS, "speedup": plain_ms / kms, "kernel": eng.last_kernel()[0],
           "same": bool((ys2 == ys_plain).all())}
    return _attach_clock(rec, mk)


def _small_batch_record(eng, scheme, log2_points, with_cpu):
    """BASELINE configs[1] / configs[2] at their literal size (one key, kInBits=32, 2^20 random points, one GPU): a
    0.3 - 0.6 ms launch, timed at steady clocks (time_steady) like every short launch in this file."""
    import fss_amd
    G, n, batch = fss_amd.U64, 32, 1 << log2_points
    sid, s0s, cws, _, _, _ = make_key(eng, scheme, n, G, 0x5EED0600 + (7 if scheme == "dcf" else 0))
    xs = _rand_xs(n, batch, 0x5EED0700 + (7 if scheme == "dcf" else 0))
    sd, cd = s0s[0].cuda(), cws.cuda()
    ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")
    mk = lambda e: (lambda: e.eval_points(sid, n, G, 0, sd, cd, xs, out=ys))
    step = mk(eng)
    _, kms = time_steady(step)
    rec = _walk_record(scheme, n, True, batch, kms * 1e-3, kms, 1, 1, eng.last_kernel()[0])
    _attach_replay(rec, "dpf_2e20" if scheme == "dpf" else "dcf_2e20", batch, 20)
    _attach_clock(rec, mk)
    shared = _shared_record(eng, scheme, sid, n, sd, cd, xs, ys, kms)
    if with_cpu:
        from oracle import oracle as O
        p = O.Prg(2 if scheme == "dpf" else 4, use_evp=True)
        osid = O.DPF if scheme == "dpf" else O.DCF
        x64 = _u64(xs.cpu().numpy())
        threads = usable_cores()
        res = {}

        def all_threads():
            res["y"] = O.eval_points(p, osid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), x64, nthreads=threads)
        reps, dt = _timed_cpu(all_threads, 1.0)
        m1 = 1 << 17
        t1 = time.perf_counter()
        O.eval_points(p, osid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), x64[:m1], nthreads=1)
        dt1 = time.perf_counter() - t1
        rec.update({"cpu": reps * batch / dt, "cpu1": m1 / dt1, "ok": bool((res["y"] == ys.cpu().numpy()).all()),
                    "cpu_sample": f"all 2^{log2_points} points x {reps} passes, {threads} threads, {dt:.1f} s; one thread: {m1} points"})
    return rec, shared


def _e2e_pcie_record(eng, log2_batch):
    """The PCIe-inclusive rate SURVEY 8(d) asks to state separately (never `value`): x from and y back to pinned host
    memory around the headline kernel -- copy, launch, copy on one stream (`serial`), and the library's host-buffer entry
    point (`pipe`: fss_amd_eval_points_host, one zero-copy launch; profiles/r4/e2e_trace.txt)."""
    import fss_amd
    G, n, N = fss_amd.U64, 32, 1 << log2_batch
    sid, s0s, cws, _, _, _ = make_key(eng, "dpf", n, G, 0x5EED0002)
    sd, cd = s0s[0].cuda(), cws.cuda()
    xs_h = torch.randint(-2**31, 2**31, (N,), dtype=torch.int32).pin_memory()
    ys_h = torch.empty((N, 4), dtype=torch.int32).pin_memory()
    xs_d = torch.empty((N,), dtype=torch.int32, device="cuda")
    ys_d = torch.empty((N, 4), dtype=torch.int32, device="cuda")

    def serial():
        xs_d.copy_(xs_h, non_blocking=True)
        eng.eval_points(sid, n, G, 0, sd, cd, xs_d, out=ys_d)
        ys_h.copy_(ys_d, non_blocking=True)

    def pipelined():  # one launch: the walk reads x from and writes y to the pinned buffers across PCIe itself
        eng.eval_points_host(sid, n, G, 0, sd, cd, xs_h, out=ys_h)

    out = {"unit": "evals/s"}
    reps = 5
    for key, fn in (("serial", serial), ("pipe", pipelined)):
        torch.cuda.synchronize()
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[key] = N / dt
        out[key + "_ms"] = dt * 1e3
    out["value"] = out["pipe"]
    out["gbs"] = out["pipe"] * 20 / 1e9
    xs_d.copy_(xs_h)
    ref = torch.empty_like(ys_d)
    eng.eval_points(sid, n, G, 0, sd, cd, xs_d, out=ref)
    out["same"] = bool((ys_h.cuda() == ref).all())
    return out


def c1_cpu_record(eng, threads):
    """BASELINE configs[0]: DPF kInBits=8, one key, full-domain 256-point evaluation on the CPU with the OpenSSL
    AES-128-MMO PRG (samples/dpf_dcf_cpu.cu:48-89: alpha = 42, beta = {7,0,0,0}, the sample's seeds and AES keys).
    The oracle runs it -- in the reference's task shape on all threads, and sequentially on one -- its outputs are
    checked against the reference's recorded SHA-256, and the GPU's EvalAll of the same key against it."""
    import hashlib
    import fss_amd
    from oracle import oracle as O
    n = 8
    seeds = np.array([[0x11111111, 0x22222222, 0x33333333, 0x44444440], [0x55555555, 0x66666666, 0x77777777, 0x88888880]],
                     dtype=np.uint32).view(np.int32)
    beta = np.array([7, 0, 0, 0], dtype=np.int32)
    p = O.Prg(2, use_evp=True)
    cws = O.dpf_gen(p, n, O.BYTES, seeds, 42, beta)
    want_sha = ("1745bb7f478e5ee9a9ef8709c6d6450d65bd440e145098d6a24515df15860cbd",
                "02067d0b4346ff98b23fff0bc989bf8a4b4f58074426ba15fa77085fb3bd47cc")
    ok_kat, ok_gpu = True, True
    for b in (0, 1):
        y = O.dpf_eval_all(p, n, O.BYTES, b, seeds[b], cws, task_threads=threads)
        ok_kat &= hashlib.sha256(y.tobytes()).hexdigest() == want_sha[b]
        g = eng.eval_all(fss_amd.DPF, n, fss_amd.BYTES, b, torch.from_numpy(seeds[b]), torch.from_numpy(cws)).cpu().numpy()
        ok_gpu &= bool((g == y).all())
    reps = 2000
    t0 = time.perf_counter()
    O.dpf_eval_all_repeat(p, n, O.BYTES, 0, seeds[0], cws, threads, reps)   # per-thread EVP contexts built once, outside the repetitions
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    O.dpf_eval_all_repeat(p, n, O.BYTES, 0, seeds[0], cws, 1, reps)
    dt1 = time.perf_counter() - t1
    sd, cd = torch.from_numpy(seeds[0]).cuda(), torch.from_numpy(cws).cuda()
    yg = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
    _, gms = time_steady(lambda: eng.eval_all(fss_amd.DPF, n, fss_amd.BYTES, 0, sd, cd, out=yg), steps=50, min_ms=50.0, warm_ms=20.0)
    # `cpu` = the better of the two shapes: on a 256-leaf tree a 16-thread OpenMP team costs more than it computes
    return {"value": 256 / (gms * 1e-3), "unit": "leaves/s", "us": gms * 1e3, "cpu": max(reps * 256 / dt, reps * 256 / dt1), "cpu1": reps * 256 / dt1,
            "cpu_us": dt / reps * 1e6, "cpu1_us": dt1 / reps * 1e6, "kat": bool(ok_kat), "ok": bool(ok_gpu and ok_kat)}


def _points_cell(args, rank, world, dist, eng, key, sch, nn, ys, pick, subs, checks):
    """One more shared-key cell of the metric: 2^B points of a (scheme, kInBits) key per GPU, timed like the headline."""
    import fss_amd
    G, batch = fss_amd.U64, ys.shape[0]
    sid, s0s, cws, _, _, _ = make_key(eng, sch, nn, G, 0x5EED0100 + nn + (7 if sch == "dcf" else 0))
    xs = _rand_xs(nn, batch, 0x5EED0200 + nn + rank)
    sd, cd = s0s[0].cuda(), cws.cuda()
    mk = lambda e: (lambda: e.eval_points(sid, nn, G, 0, sd, cd, xs, out=ys))
    step = mk(eng)
    pre_warm(step, 40.0)
    w, k = time_steps(step, args.steps, args.warmup, dist)
    subs[key] = _attach_clock(_walk_record(sch, nn, True, batch, w, k, world, args.steps, eng.last_kernel()[0]), mk)
    checks[key] = ("points", sch, nn, s0s, cws, xs[pick].cpu().numpy(), ys[pick].cpu().numpy())
    if key == "dcf_points":  # the DCF cell of the metric through the shared-prefix plan as well
        subs["shared_dcf"] = _shared_record(eng, sch, sid, nn, sd, cd, xs, ys, k, timer=lambda f: time_steps(f, args.steps, args.warmup, dist),
                                            world=world, steps=args.steps)


def _keys_cell(args, rank, world, dist, eng, key, sch, nn, ys, pick, subs, checks):
    """M(ii): 2^B independent keys x 1 point, keys generated on the device straight into the level-major layout
    (outside the timed region, as the reference excludes Gen and relayout: src/bench_gpu.cu:616-627)."""
    import fss_amd
    from fss_amd.engine import LevelMajorKeys
    G, batch = fss_amd.U64, ys.shape[0]
    sid = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF}[sch]
    gen = torch.Generator(device="cuda").manual_seed(0x5EED0300 + nn + (7 if sch == "dcf" else 0) + 1000 * rank)
    ks0s = torch.randint(-2**31, 2**31, (batch, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    kbetas = torch.randint(-2**31, 2**31, (batch, 4), dtype=torch.int32, device="cuda", generator=gen)
    kalphas = _rand_xs(nn, batch, 0x5EED0380 + nn + 1000 * rank)
    kxs = _rand_xs(nn, batch, 0x5EED0400 + nn + 1000 * rank)
    kxs[::2] = kalphas[::2]  # half of the points hit their key's alpha
    keys, _ = eng.gen_level_major(sid, nn, G, ks0s, kalphas, kbetas)
    kseeds = ks0s[:, 0].contiguous()
    del ks0s, kbetas, kalphas
    mk = lambda e: (lambda: e.eval_keys(keys, G, 0, kseeds, kxs, out=ys))
    step = mk(eng)
    pre_warm(step, 40.0)
    w, k = time_steps(step, args.steps, args.warmup, dist)
    subs[key] = _attach_clock(_walk_record(sch, nn, False, batch, w, k, world, args.steps, eng.last_kernel()[0]), mk)
    if rank == 0 and world == 1 and not args.no_cpu:
        m = int(pick.numel())
        if sch == "dpf":
            sub = LevelMajorKeys(sid, nn, m, keys.cw_s[:, pick].contiguous(), tr_bits=keys.tr_bits[pick].contiguous(),
                                 out_cw=keys.out_cw[pick].contiguous())
        else:
            sub = LevelMajorKeys(sid, nn, m, keys.cw_s[:, pick].contiguous(), cw_v=keys.cw_v[:, pick].contiguous(),
                                 out_cw=keys.out_cw[pick].contiguous())
        checks[key] = ("keys", sch, nn, kseeds[pick].cpu().numpy(), eng.key_major(sub).cpu().numpy(), kxs[pick].cpu().numpy(),
                       ys[pick].cpu().numpy())
    del keys, kseeds, kxs
    torch.cuda.empty_cache()


POINTS_CELLS = (("dcf_points", "dcf", 32), ("dpf_n16", "dpf", 16), ("dpf_n64", "dpf", 64), ("dcf_n16", "dcf", 16), ("dcf_n64", "dcf", 64))
KEYS_CELLS = (("dpf_keys", "dpf", 32), ("dcf_keys", "dcf", 32), ("dpf_keys_n16", "dpf", 16), ("dpf_keys_n64", "dpf", 64))
SHARDED = ("evalall_c4", "ht_c5", "grotto_c5")


def run_headline(args, rank, world, dist):
    import fss_amd
    eng = bench_engine()
    n, scheme = 32, "dpf"
    batch = 1 << args.log2_batch
    G = fss_amd.U64
    sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, G, 0x5EED0002)
    xs = _rand_xs(n, batch, 0x5EED0002 + rank)
    seed_d, cws_d = s0s[0].cuda(), cws.cuda()
    ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")

    ranks = rank_identities(rank, world, dist)
    mk = lambda e: (lambda: e.eval_points(sid, n, G, 0, seed_d, cws_d, xs, out=ys))
    step = mk(eng)
    tm = time_steps(step, args.steps, args.warmup, dist, per_step=True)
    wall, kernel_ms = tm
    kname, grid, block = eng.last_kernel()
    head = _walk_record(scheme, n, True, batch, wall, kernel_ms, world, args.steps, kname)
    head.update({"grid": grid, "block": block, "ms_median": tm.median})
    _attach_clock(head, mk)

    # `headline_strong`: the ONE 2^B-point batch cut into `world` slices.  With one rank that is the launch above.
    if world > 1:
        strong, _ = sharded_record(args, "headline_strong", rank, world, dist, eng, False)
    else:
        strong = dict(head, scaling="strong")

    # Same batch through the shared-prefix plan (identical outputs; the key's top levels are expanded once per step
    # instead of once per point).  Reported beside the headline, never as `value`: it removes AES work.
    shared = _shared_record(eng, scheme, sid, n, seed_d, cws_d, xs, ys, kernel_ms, timer=lambda f: time_steps(f, args.steps, args.warmup, dist),
                            world=world, steps=args.steps)

    # ---- the rest of the metric: DCF, independent keys, kInBits 16 / 64 (north_star), same protocol
    subs, checks = {"shared_prefix": shared}, {}
    m = min(batch, 4096)  # evaluations per sub-record handed to the CPU check (a strided sample)
    pick = torch.arange(0, batch, max(1, batch // m), device="cuda")[:m]
    if not args.only_headline:
        for key, sch, nn in POINTS_CELLS:
            _points_cell(args, rank, world, dist, eng, key, sch, nn, ys, pick, subs, checks)
        for key, sch, nn in KEYS_CELLS:
            _keys_cell(args, rank, world, dist, eng, key, sch, nn, ys, pick, subs, checks)
    eng.eval_points(sid, n, G, 0, seed_d, cws_d, xs, out=ys)  # the headline outputs again, for the CPU comparison below
    # the headline's own CPU leg needs the first cpu_sample points and their outputs only: free the batch before the
    # configs[3] / configs[4] sub-records take their 4 - 17 GiB
    mm = min(batch, args.cpu_sample)
    head_xs, head_ys = xs[:mm].cpu().numpy().view(np.uint32), ys[:mm].cpu().numpy()
    del xs, ys
    torch.cuda.empty_cache()
    if not args.only_headline:
        with_cpu = world == 1 and not args.no_cpu
        if world == 1:
            for key, sch in (("dpf_2e20", "dpf"), ("dcf_2e20", "dcf")):
                subs[key], subs["shared_" + key] = _small_batch_record(eng, sch, 20, with_cpu)
        for name in SHARDED:
            rec, _ = sharded_record(args, name, rank, world, dist, eng, with_cpu)
            subs[name] = rec
            del _
            torch.cuda.empty_cache()
        if world == 1 and rank == 0:
            subs["e2e_pcie"] = _e2e_pcie_record(eng, args.log2_batch)
            if with_cpu:
                subs["c1_cpu"] = c1_cpu_record(eng, usable_cores())
    if rank != 0:
        return
    cpu = None
    if world == 1 and not args.no_cpu:
        cpu = cpu_baseline(scheme, n, s0s, cws, ocw, head_xs, head_ys, usable_cores())
        for key, ok in cpu_check_subrecords(checks).items():  # part of the cpu_baseline leg: the oracle as checker
            subs[key]["ok"] = ok
    line, detail = compose_line(args, world, head, strong, subs, cpu, dist.get_backend() if dist is not None else None, ranks)
    print(json.dumps(line, separators=(",", ":")), flush=True)
    _write_detail(detail)


def compose_line(args, world, head, strong, subs, cpu, process_group, ranks=None):
    """The one JSON line (short; KEYS / RECORDS / TOP describe it) and the unabridged detail record.
    The metric is "batch=2^24 on 1/2/4/8": with more than one rank `value` is the ONE 2^B-point batch cut into `world`
    slices (`strong`, the record also printed as headline_strong) and the line says "scaling": "strong"; what every rank
    does with a 2^B batch of its own is the side record `weak`.  With one rank the two are the same launch."""
    from fss_amd import _build
    B = args.log2_batch
    batch = 1 << B
    prim = strong if world > 1 else head   # the record `value`, `ms_per_step` and the roofs describe
    kernel_ms = prim["kernel_ms"]
    at_profiled_size = prim["units_per_launch"] == (1 << 24)
    pipes = measured_pipe_busy(B) if at_profiled_size else None
    replay_v, replay_src = _headline_pmc(B) if at_profiled_size else (None, None)
    # True only when the profile was captured for exactly the sources this library's point walk is compiled from
    replay_ok = bool(replay_v) and replay_v.get("_captured_for") == _build.headline_digest()
    traffic = measured_hbm_traffic(B) if at_profiled_size else None
    per_rank = f"2^{B} / {world} points of ONE 2^{B}-point batch per GPU per step (shard.batch_slice)" if world > 1 else f"2^{B} random points per step"
    line = {
        "metric": METRIC, "value": prim["value"], "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": prim["ms_per_step"], "ms_median": prim.get("ms_median"), "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"DPF kInBits=32, 1 key, {per_rank} (configs[1] shape at the metric's batch), "
                               "party 0, Uint<uint64_t>",
                   "kernel": prim["kernel"], "grid": prim.get("grid"), "block": prim.get("block"),
                   "parallelism": f"batch-sharded x{world}, no collective", "process_group": process_group, "legend": "bench_legend.json"},
        "roofline": {"bound": "lds", "achieved": prim["lds_gbs"], "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": prim["frac"],
                     "traffic": traffic, "kernel_ms": kernel_ms, "clk": prim.get("clk"), "frac_at_clock": prim.get("frac_at_clock"),
                     "replay": replay_src, "replay_ok": replay_ok},
        "roofline_hbm": {"bound": "hbm", "achieved": prim["hbm_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": prim["hbm_frac"],
                         "traffic": traffic, "algorithmic_bytes": prim["units_per_launch"] * 20},
    }
    if line["ms_median"] is None:
        del line["ms_median"]
    if ranks is not None:
        names = {r.get("name") for r in ranks}
        if len(names) == 1:  # the usual case: say it once
            line["config"]["device_name"] = names.pop()
            ranks = [{k: v for k, v in r.items() if k != "name"} for r in ranks]
        line["config"]["ranks"] = ranks
        line["config"]["distinct_devices"] = distinct_devices(ranks)
    if world > 1:
        line["weak"] = dict(compact(head), scaling="weak")
    if pipes:
        line["roofline"].update({"lds_busy": pipes["lds_busy"], "valu_busy": pipes["valu_busy"]})
        # third roof SURVEY 8(d) asks for: VALU issue.  Peak = 256 CU x 4 SIMD x one wave64 instruction per 2 clk x
        # 2.4 GHz (the fast-class rate); this kernel's instruction mix issues at ~4 clk, so the
        # pipe is ~0.9 busy at less than half of that peak.  Instruction count replayed from the profile, time live.
        valu_rate = pipes["valu_wave_instructions_per_launch"] / (kernel_ms * 1e-3)
        line["roofline_valu"] = {"bound": "valu", "achieved": valu_rate / 1e12, "peak": 1.2288, "unit": "T wave64-instr/s",
                                 "frac": valu_rate / 1.2288e12, "busy": pipes["valu_busy"]}
    if cpu is not None:
        line["cpu_baseline"] = cpu
    line["headline_strong"] = dict(compact(strong), scaling="strong")
    line["sub"] = {k: compact(v) for k, v in subs.items()}
    gates = [v for rec in list(line["sub"].values()) + ([{"ok": cpu["ok"]}] if cpu else []) for k, v in rec.items() if k in ("ok", "same", "kat")]
    if gates:
        line["gates"] = {"all": all(gates), "n": len(gates)}
    detail = {"line": line, "head": head, "strong": strong, "sub": subs, "cpu_baseline": cpu, "legend": legend()}
    return r5(line), detail


def _write_detail(detail):
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_detail.json"), "w") as f:
            json.dump(detail, f, indent=1, default=str)
    except OSError:
        pass


def sharded_record(args, name, rank, world, dist, eng, with_cpu):
    """Workloads cut across the ranks under the same launch contract (barrier + synchronize, max over ranks; strong scaling):
      headline_strong: the ONE 2^B-point batch of the headline key, points [lo, hi) = shard.batch_slice per rank;
      evalall_c4: DPF EvalAll kInBits=--log2-domain (28); the domain is cut into `world` aligned blocks
                  (shard.domain_block), rank r expands block r into its own HBM (no collective);
      ht_c5 / grotto_c5: kInBits=64, 2^--log2-keys (24) independent keys x 1 point, key batch split across the ranks
                  (shard.batch_slice).
    On one GPU with the CPU leg enabled the record also carries the CPU baseline (and its parity gate).  Returns the
    record (meaningful on rank 0) and the outputs (for callers that hash them)."""
    import fss_amd
    from fss_amd import shard
    G = fss_amd.U64
    cpu = None
    if name == "headline_strong":
        n, total = 32, 1 << args.log2_batch
        sid, s0s, cws, _, _, _ = make_key(eng, "dpf", n, G, 0x5EED0002)
        lo, hi = shard.batch_slice(total, rank, world)
        # the batch is a function of the global point index: a rank's slice equals the same slice of a one-GPU run
        xs = _rand_xs(n, total, 0x5EED0002)[lo:hi].contiguous()
        torch.cuda.empty_cache()
        sd, cd = s0s[0].cuda(), cws.cuda()
        ys = torch.empty((hi - lo, 4), dtype=torch.int32, device="cuda")
        mk = lambda e: (lambda: e.eval_points(sid, n, G, 0, sd, cd, xs, out=ys))
        step = mk(eng)
        units_total, units_rank, unit, aes_per_unit, hbm_unit = total, hi - lo, "evals/s", 32.0, 20
    elif name == "evalall_c4":
        n = args.log2_domain
        sid, s0s, cws, ocw, _, _ = make_key(eng, "dpf", n, G, 0x5EED0004)
        first, nleaves = shard.domain_block(n, rank, world)
        sd, cd = s0s[0].cuda(), cws.cuda()
        ys = torch.empty((nleaves, 4), dtype=torch.int32, device="cuda")
        mk = lambda e: (lambda: e.eval_all(sid, n, G, 0, sd, cd, first_leaf=first, nleaves=nleaves, out=ys))
        step = mk(eng)
        units_total, units_rank, unit, aes_per_unit, hbm_unit = 1 << n, nleaves, "leaves/s", 2.0, 16
    elif name == "grotto_evalall":
        # Grotto DCF full domain, domain-sharded: the one workload of the path with an exchange step -- the running XOR
        # crosses block boundaries, one parity byte per rank all-gathered per step (shard.grotto_fix_shard).  Not part
        # of the default line; tests/test_bench_launch.py runs it with 8 ranks on one GPU.
        n = args.log2_domain
        g0 = torch.Generator().manual_seed(0x5EED0006)
        s0s = torch.randint(-2**31, 2**31, (1, 2, 4), dtype=torch.int32, generator=g0)
        s0s[..., 3] &= ~1
        alpha = torch.randint(0, 2**31, (1,), dtype=torch.int32, generator=g0) & ((1 << min(n, 31)) - 1)
        cws1, _ = eng.gen(fss_amd.DPF, n, fss_amd.BYTES, s0s, alpha, torch.zeros((1, 4), dtype=torch.int32))
        first, nleaves = shard.domain_block(n, rank, world)
        sd, cd = s0s[0, 0].cuda(), cws1[0]
        ys = torch.empty((nleaves,), dtype=torch.uint8, device="cuda")

        def warm_step():  # the kernels alone: pre_warm's repetition count differs between ranks, a collective must not be in it
            eng.eval_all(fss_amd.GROTTO, n, fss_amd.BYTES, 0, sd, cd, first_leaf=first, nleaves=nleaves, out=ys)

        def step():
            warm_step()
            shard.grotto_fix_shard(ys, dist)
        mk = None
        units_total, units_rank, unit, aes_per_unit, hbm_unit = 1 << n, nleaves, "leaves/s", 2.0, 1
    else:
        n, total = 64, 1 << args.log2_keys
        scheme = "ht" if name == "ht_c5" else "grotto"
        lo, hi = shard.batch_slice(total, rank, world)
        nk = hi - lo
        # every key of the batch is a function of its global index: a rank's slice equals the same slice of a
        # single-GPU run (tests/test_bench_launch.py hashes both)
        gen = torch.Generator(device="cuda").manual_seed(0x5EED0005)
        s0s = torch.randint(-2**31, 2**31, (total, 2, 4), dtype=torch.int32, device="cuda", generator=gen)[lo:hi].contiguous()
        betas = torch.randint(-2**31, 2**31, (total, 4), dtype=torch.int32, device="cuda", generator=gen)[lo:hi].contiguous()
        alphas = torch.randint(-2**63, 2**63 - 1, (total,), dtype=torch.int64, device="cuda", generator=gen)
        xs = torch.randint(-2**63, 2**63 - 1, (total,), dtype=torch.int64, device="cuda", generator=gen)
        xs[::2] = alphas[::2]  # half of the points hit their key's alpha
        alphas, xs = alphas[lo:hi].contiguous(), xs[lo:hi].contiguous()
        torch.cuda.empty_cache()
        sid = fss_amd.HT if scheme == "ht" else fss_amd.GROTTO
        gg = G if scheme == "ht" else fss_amd.BYTES
        if scheme == "grotto":
            betas.zero_()
        keys, ocws = eng.gen_level_major(sid, n, gg, s0s, alphas, betas)  # outside the timed region (src/bench_gpu.cu:616-627)
        seeds0 = s0s[:, 0].contiguous()
        del s0s, betas, alphas
        ys = (torch.empty((nk, 4), dtype=torch.int32, device="cuda") if scheme == "ht"
              else torch.empty((nk,), dtype=torch.uint8, device="cuda"))
        mk = lambda e: (lambda: e.eval_keys(keys, gg, 0, seeds0, xs, ocws=ocws, out=ys))
        step = mk(eng)
        units_total, units_rank, unit = total, nk, "evals/s"
        aes_per_unit = 64.0 if scheme == "ht" else 96.0
        hbm_unit = 1084 if scheme == "ht" else 1057   # 16 seed + 8 x + 64 x 16 cw_s + (4 extra + 16 ocw + 16 y | 8 tr bits + 1 y)
    pre_warm(warm_step if name == "grotto_evalall" else step)
    tm = time_steps(step, args.steps, args.warmup, dist, per_step=True)
    wall, kernel_ms = tm
    kname, grid, block = eng.last_kernel()
    if with_cpu and rank == 0 and world == 1:
        threads = usable_cores()
        if name == "evalall_c4":
            cpu = cpu_evalall(eng, threads, min(22, n))
            # the timed outputs themselves: a strided sample of the 2^n leaves (and both ends) against the oracle's
            # per-point Eval -- EvalAll's leaf x is Eval(x) (dpf.cuh:232-303 vs :170-214)
            from oracle import oracle as O
            m = min(1 << n, 4096)
            idx = torch.cat([torch.arange(0, 1 << n, max(1, (1 << n) // m), device="cuda")[:m],
                             torch.tensor([(1 << n) - 1], device="cuda")])
            want = O.eval_points(O.Prg(2), O.DPF, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), idx.cpu().numpy().astype(np.uint64), nthreads=4)
            cpu["timed_outputs_ok"] = bool((ys[idx].cpu().numpy() == want).all())
            cpu["ok"] = bool(cpu["ok"] and cpu["timed_outputs_ok"])
        elif name in ("ht_c5", "grotto_c5"):
            cpu = cpu_keys(eng, scheme, n, keys, seeds0, xs, ocws, ys, threads, 1 << 18)
    rate = units_rank / (kernel_ms * 1e-3)
    lds_gbs = rate * aes_per_unit * LOOKUP_BYTES_PER_AES / 1e9
    rec = {"value": units_total * args.steps / wall, "unit": unit, "ms_per_step": wall / args.steps * 1e3, "kernel_ms": kernel_ms,
           "units_per_launch": units_rank, "scaling": "strong", "kernel": kname, "grid": grid, "block": block,
           "aes_per_unit": aes_per_unit, "frac": lds_gbs / LDS_PEAK_GBS, "lds_gbs": lds_gbs, "hbm_gbs": rate * hbm_unit / 1e9,
           "hbm_frac": rate * hbm_unit / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": units_rank * hbm_unit,
           "ms_median": tm.median}
    if name == "grotto_c5":
        # the 32 sibling blocks of a walk are control-bit-only (aes_mmo<.., LSB3>: 133 of 160 lookups); `frac` charges
        # SURVEY 8(d)'s algorithmic 96 whole blocks, `frac_executed` the lookups the kernel issues
        rec["frac_executed"] = rec["frac"] * (64 * 160 + 32 * 133) / (96 * 160.0)
    _attach_replay(rec, name, units_rank, hbm_unit)
    if mk is not None:
        _attach_clock(rec, mk)
    if cpu is not None:
        rec.update(cpu)
    return rec, ys


def run_sharded(args, rank, world, dist):
    """`--workload headline_strong | evalall_c4 | ht_c5 | grotto_c5 | grotto_evalall` as the whole run, one JSON line."""
    eng = bench_engine()
    rec, ys = sharded_record(args, args.workload, rank, world, dist, eng, with_cpu=not args.no_cpu)
    if args.hash_out:
        # tests/test_bench_launch.py: every rank leaves the SHA-256 of its outputs, cut into --hash-blocks equal blocks
        # (an 8-rank run's eight digests must equal the eight block digests of a single-rank run)
        import hashlib
        from fss_amd import shard
        arr = ys.cpu().numpy()
        with open(f"{args.hash_out}.rank{rank}", "w") as f:
            for b in range(max(1, args.hash_blocks)):
                lo, hi = shard.batch_slice(arr.shape[0], b, max(1, args.hash_blocks))
                f.write(hashlib.sha256(arr[lo:hi].tobytes()).hexdigest() + f" {hi - lo}\n")
    if rank != 0:
        return
    print(json.dumps(r5(_workload_line(args, args.workload, rec, world, "strong", dist))), flush=True)


def _workload_line(args, name, rec, world, scaling, dist):
    """The line of a `--workload` run (what tools/profile_workloads.sh profiles): one workload, the contract keys."""
    line = {"metric": f"{name}: {rec['unit'][:-2]} per second", "value": rec["value"], "unit": rec["unit"], "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "ms_median": rec.get("ms_median"),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": RECORDS.get(name, name), "kernel": rec["kernel"], "grid": rec.get("grid"), "block": rec.get("block"),
                       "units_per_launch": rec["units_per_launch"], "parallelism": f"sharded x{world}, no collective",
                       "process_group": dist.get_backend() if dist is not None else None},
            "roofline": {"bound": "lds", "achieved": rec["lds_gbs"], "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": rec["frac"],
                         "traffic": rec.get("traffic"), "kernel_ms": rec["kernel_ms"], "replay": rec.get("replay"),
                         "replay_ok": rec.get("replay_ok"), "lds_busy_pmc": rec.get("lds_busy_pmc")},
            "roofline_hbm": {"bound": "hbm", "achieved": rec["hbm_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rec["hbm_frac"],
                             "algorithmic_bytes_per_launch": rec.get("algorithmic_bytes_per_launch", rec["units_per_launch"] * 20)}}
    if "frac_executed" in rec:
        line["roofline"]["frac_executed"] = rec["frac_executed"]
    if "cpu" in rec:
        line["cpu_baseline"] = {"value": rec["cpu"], "unit": rec["unit"], "cores": usable_cores(), "kind": "port",
                                "sample": rec.get("cpu_sample", ""), "t1": rec.get("cpu1"), "ok": rec.get("ok")}
    return line



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log2-batch", type=int, default=24)
    ap.add_argument("--cpu-sample", type=int, default=(1 << 24) if (os.cpu_count() or 1) >= 8 else (1 << 22), help="points of the batch timed on the CPU")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--suite", action="store_true")
    ap.add_argument("--workload", default="dpf_points",
                    choices=["dpf_points", "headline_strong", "dpf_2e20", "dcf_2e20", "evalall_c4", "ht_c5", "grotto_c5", "grotto_evalall"],
                    help="dpf_points = the headline (BASELINE metric) with every sub-record; the others are one workload as the whole run")
    ap.add_argument("--log2-domain", type=int, default=28, help="evalall_c4: kInBits of the full domain (configs[3]: 28)")
    ap.add_argument("--log2-keys", type=int, default=24, help="ht_c5 / grotto_c5: log2 of the key batch (configs[4]: 24)")
    ap.add_argument("--hash-out", default="", help="--workload runs: every rank writes the SHA-256 of its outputs to <this>.rank<r>")
    ap.add_argument("--hash-blocks", type=int, default=1, help="with --hash-out: digests of this many equal blocks of the rank's outputs")
    ap.add_argument("--only-headline", action="store_true",
                    help="skip the sub-records (DCF, independent keys, kInBits 16 / 64): the PMC passes of tools/profile_workloads.sh "
                         "use this so that one kernel template means one workload")
    ap.add_argument("--no-clk", action="store_true",
                    help="skip the in-kernel clock measurements (clk / fc: ~50 ms of each record's kernel through the diagnostic build, "
                         "after its timed region)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="self-launched multi-GPU runs: seconds before the ranks are stopped and the run fails")
    ap.add_argument("--write-legend", action="store_true", help="write bench_legend.json (the line's key and record descriptions) and exit")
    args = ap.parse_args()
    global NO_CLK
    NO_CLK = args.no_clk
    if args.write_legend:
        with open(os.path.join(ROOT, "bench_legend.json"), "w") as f:
            json.dump(legend(), f, indent=1)
            f.write("\n")
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # Not under a launcher but asked for several GPUs: become the launcher.  This must stay ahead of anything that
    # initialises the GPU (torch.cuda.is_available() does): the ranks are fresh child processes.
    # FSS_AMD_BENCH_SELF_LAUNCH=1: the launcher also for --gpus 1, so that its RCCL branch (rank spawned by launch_ranks,
    # process group "nccl") can run on a one-GPU box (tests/test_bench_launch.py).
    if "RANK" not in os.environ and (args.gpus > 1 or os.environ.get("FSS_AMD_BENCH_SELF_LAUNCH") == "1"):
        sys.exit(launch_ranks(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dry = os.environ.get("FSS_AMD_BENCH_DRY_RUN") == "1"
    # Test hook for boxes with fewer GPUs than ranks (exercises the N > 1 control flow only, not a
    # measurement): ranks share the visible GPUs and rendezvous over gloo, since RCCL refuses two ranks on one device.
    share = os.environ.get("FSS_AMD_BENCH_SHARE_GPU") == "1"
    if not dry:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
        if share:
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
    dist = None
    # Under a launcher (torch.distributed.run sets RANK and MASTER_ADDR) the process group is formed even for one rank:
    # `--nproc-per-node 1` then walks the same RCCL rendezvous, barriers and reduction as an 8-GPU run does.
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import datetime
        import torch.distributed as dist_mod
        # a rank that dies before the first barrier must not leave the others waiting for ever (under
        # torch.distributed.run the launcher ends them; under bench.py's own launcher launch_ranks does; this
        # bounds the remaining cases)
        tmo = datetime.timedelta(seconds=min(args.launch_timeout, 600.0))
        # gloo announces its connections on stdout and RCCL prints a version banner when its communicator comes up:
        # keep stdout for the one JSON line (the first barrier is inside, so that a lazily built communicator is too)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if share or dry:
                dist_mod.init_process_group("gloo", timeout=tmo)
            else:
                dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)
            dist_mod.barrier()
        finally:
            sys.stdout.flush()
            try:  # RCCL's banner goes through C stdio, which buffers when stdout is a pipe: flush it while fd 1 still is stderr
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved, 1)
            os.close(saved)
        dist = dist_mod
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    if dry:
        run_dry(args, rank, world, dist)
    elif args.workload == "dpf_points":
        run_headline(args, rank, world, dist)
    elif args.workload in ("dpf_2e20", "dcf_2e20"):
        # configs[1] / configs[2] at their literal size as the whole run (one GPU; the profiling passes use this)
        if world != 1:
            raise SystemExit(f"--workload {args.workload} is a one-GPU workload")
        import fss_amd
        rec, _ = _small_batch_record(bench_engine(), args.workload[:3], 20, not args.no_cpu)
        print(json.dumps(r5(_workload_line(args, args.workload, rec, 1, "weak", None))), flush=True)
    else:
        run_sharded(args, rank, world, dist)
    if args.suite and world == 1 and not dry:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_suite
        bench_suite.run_suite(args)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
