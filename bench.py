#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X DPF/DCF evaluation engine.

Metric (BASELINE.json): DPF/DCF evals/sec at kInBits=32, batch=2^24.
A "step" is one pass of the hot path over one batch: ONE launch of the shared-key
DPF point-evaluation kernel over 2^24 uniformly random points (configs[1]'s shape --
DPF Boyle'16, kInBits=32, one key -- at the batch the metric is quoted on), party 0,
Uint<uint64_t> outputs, inputs already resident in HBM.  With --gpus N every rank
evaluates its own 2^24-point batch of the same key (the points x keys batch shards
with no collective on the data path; weak scaling), the timed region is bracketed by
barrier + synchronize and the slowest rank's time is used.

`python bench.py --gpus N` with no launcher around it starts its own N ranks (one process per GPU, spawned before
anything touches the GPU; a failed or hung rank ends the run with a non-zero exit code); under
`python -m torch.distributed.run` (RANK / WORLD_SIZE set) it is one of the ranks.

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).  Besides the headline (`value`: DPF, one
key) the line carries the rest of the metric as sub-records timed the same way: `dcf_points` (DCF n=32, one key,
2^24 points), `dpf_keys` (2^24 independent level-major keys, one point each), `dpf_n16` / `dpf_n64`.  `roofline` is
the binding roof of the kernel -- LDS table-lookup bandwidth (the path is
neither HBM- nor MFMA-bound) -- and `roofline_hbm` the HBM figure next to it.
`cpu_baseline` times the oracle's restatement of the reference CPU path (OpenSSL EVP
AES-128-ECB, one EVP_EncryptUpdate per block, both children computed --
include/fss/prg/aes128_mmo.cuh:72-93, include/fss/dpf.cuh:170-214) on a bounded sample
of the same points and checks the GPU output against it.

--suite additionally times the other BASELINE configs on one GPU and writes
gpurun_out/bench_suite.json (not part of the driver contract).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# MI355X constants (/opt/skills/guides/MI355X_MICROARCH.md): 256 CUs, 2.4 GHz max clock,
# ds_read_b32 = 128 B/clk/CU, HBM3E 8 TB/s.
N_CU, CLK_HZ = 256, 2.4e9
LDS_PEAK_GBS = 128 * N_CU * CLK_HZ / 1e9   # 78 643 GB/s of 4-byte table lookups
HBM_PEAK_GBS = 8000.0
LOOKUP_BYTES_PER_AES = 160 * 4              # 144 Te0 + 16 S-box lookups, 4 B each
CHACHA_VALU_PER_WAVE_BLOCK = 1003           # measured wave-instructions per 64 ChaCha20 blocks (960 in the rounds)
VALU_ISSUE_PEAK_GINSTR = N_CU * 4 * CLK_HZ / 2.0 / 1e9   # 1 228.8 G wave64-instr/s: 32 lanes per SIMD and clock (the headline's roofline_valu peak)


def aes_blocks_per_eval(scheme, n):
    """Algorithmic minimum (taken child only), SURVEY.md 8(d)."""
    return {"dpf": n, "dcf": 2 * n, "ht": n}[scheme]


def hbm_bytes_per_eval(scheme, n, shared):
    if shared:
        return (4 if n <= 32 else 8) + 16
    per_level = {"dpf": 16, "dcf": 32, "ht": 16}[scheme]
    fixed = {"dpf": 16 + 8 + 16 + 16, "dcf": 16 + 16 + 16, "ht": 16 + 4 + 16 + 16}[scheme]
    return fixed + per_level * n + (4 if n <= 32 else 8)


def _headline_pmc(log2_batch):
    """Counters of the headline kernel (plain DPF walk, shared key) REPLAYED from the committed rocprofv3 PMC passes
    (profiles/r<N>/pmc_summary.json, highest N first): bench.py cannot read PMCs itself.  Only valid for the batch
    profiled (2^24).  Returns (values, path relative to the repo) or (None, None)."""
    import glob
    import re
    if log2_batch != 24:
        return None, None
    files = glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json"))
    rnd = lambda f: int(re.search(r"profiles[/\\]r(\d+)", f).group(1))
    for f in sorted(files, key=rnd, reverse=True):
        try:
            d = json.load(open(f))
            for k, v in d.items():
                name = k.replace("fssamd::Lds, ", "")
                if "eval_points_kernel<0, true" in name and name.rstrip().endswith("false>") and "FETCH_SIZE" in v:
                    v = dict(v, _captured_for=d.get("_captured_for", {}).get("headline_sources_sha256"))
                    return v, os.path.relpath(f, ROOT)
        except Exception:
            pass
    return None, None


def measured_hbm_traffic(log2_batch):
    """HBM bytes per launch: FETCH_SIZE and WRITE_SIZE in KiB, collected in separate passes; FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950.  Replayed from profiles/ (see `traffic_replayed_from`)."""
    v, _ = _headline_pmc(log2_batch)
    if not v or "WRITE_SIZE" not in v:
        return None
    return (2 * v["FETCH_SIZE"]["mean_per_launch"] + v["WRITE_SIZE"]["mean_per_launch"]) * 1024.0


def measured_pipe_busy(log2_batch):
    """LDS- and VALU-busy fractions of the profiled launch (north_star: "VALU-busy against the chip's peak"):
    SQ_LDS_IDX_ACTIVE per CU, and SQ_ACTIVE_INST_VALU (counted in 4-cycle issue slots) per SIMD, over the
    kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs).  Ratios of counters of one profiled launch -- not of this run."""
    v, src = _headline_pmc(log2_batch)
    need = ("GRBM_GUI_ACTIVE", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU")
    if not v or any(k not in v for k in need):
        return None
    cyc = v["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0
    return {"lds_busy": v["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"] / 256.0 / cyc,
            "valu_busy": v["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] * 4.0 / 1024.0 / cyc,
            "valu_wave_instructions_per_launch": v["SQ_INSTS_VALU"]["mean_per_launch"],
            "replayed_from": src}


def time_steps(fn, steps, warmup, dist):
    """W warm-up steps, then exactly K timed steps between barrier+synchronize pairs.
    Returns (wall seconds max over ranks, mean kernel ms from HIP events on the launch stream)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / steps
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall, kernel_ms


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


def make_key(eng, scheme, n, group_id, seed_val):
    """One valid key from fixed seeds, generated by the product's device Gen."""
    import fss_amd
    g = torch.Generator().manual_seed(seed_val)
    s0s = torch.randint(-2**31, 2**31, (1, 2, 4), dtype=torch.int32, generator=g)
    s0s[..., 3] &= ~1
    beta = torch.randint(-2**31, 2**31, (1, 4), dtype=torch.int32, generator=g)
    beta[..., 3] &= ~1
    if n <= 32:
        alpha = torch.randint(0, 2**31, (1,), dtype=torch.int32, generator=g) & ((1 << min(n, 31)) - 1)
    else:
        alpha = torch.randint(0, 2**62, (1,), dtype=torch.int64, generator=g)
    sid = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF, "ht": fss_amd.HT}[scheme]
    cws, ocws = eng.gen(sid, n, group_id, s0s, alpha, beta)
    return sid, s0s[0].cpu(), cws[0].cpu(), (None if ocws is None else ocws[0].cpu()), alpha, beta


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
    parity = bool((ys == gpu_sample).all())
    out = {
        "value": reps * len(xs64) / dt, "unit": "evals/s", "cores": threads, "kind": "port",
        "sample": f"first {len(xs64)} points of the timed batch x {reps} passes, party 0, oracle with OpenSSL EVP AES-128-ECB "
                  f"PRG (reference call pattern), {threads} OpenMP threads, {dt:.1f} s; single thread: {n1} points",
        "single_thread_value": n1 / dt1,
        "gpu_matches_oracle_on_sample": parity,
    }
    if O.have_aesni():
        # "best CPU" line: the same walk with AES-NI intrinsics, as the reference's Aes128MmoRaw (README.md:341)
        pn = O.Prg(mul, aesni=True)
        t2 = time.perf_counter()
        ysn = O.eval_points(pn, sid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), xs64, ocw=oc, nthreads=threads)
        dt2 = time.perf_counter() - t2
        out["aesni_value"] = len(xs64) / dt2
        out["aesni_matches"] = bool((ysn == ys).all())
    return out


def _rand_xs(n, batch, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    if n <= 32:
        xs = torch.randint(-2**31, 2**31, (batch,), dtype=torch.int32, device="cuda", generator=gen)
        return xs & ((1 << n) - 1) if n < 32 else xs
    return torch.randint(-2**63, 2**63 - 1, (batch,), dtype=torch.int64, device="cuda", generator=gen)


def _sub_record(desc, scheme, n, shared, units, wall, kernel_ms, world, steps, kname):
    """One more workload of the metric, timed like the headline (same barrier / max-over-ranks protocol)."""
    rate = units / (kernel_ms * 1e-3)
    aes = aes_blocks_per_eval(scheme, n)
    lds = rate * aes * LOOKUP_BYTES_PER_AES / 1e9
    hbm = rate * hbm_bytes_per_eval(scheme, n, shared) / 1e9
    return {"workload": desc, "value": units * world * steps / wall, "unit": "evals/s", "ms_per_step": wall / steps * 1e3,
            "kernel_ms": kernel_ms, "per_gpu_evals_per_s": rate, "aes_blocks_per_eval": aes,
            "roofline": {"bound": "lds", "achieved": lds, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": lds / LDS_PEAK_GBS},
            "hbm_gbs_algorithmic": hbm, "hbm_frac": hbm / HBM_PEAK_GBS, "kernel": kname}


def _small_batch_record(eng, scheme, log2_points, with_cpu):
    """BASELINE configs[1] / configs[2] at their literal size (one key, kInBits=32, 2^20 random points, one GPU): a
    0.3 - 0.6 ms launch, timed at steady clocks (time_steady) like every short launch in this file."""
    import fss_amd
    G, n, batch = fss_amd.U64, 32, 1 << log2_points
    sid, s0s, cws, _, _, _ = make_key(eng, scheme, n, G, 0x5EED0600 + (7 if scheme == "dcf" else 0))
    xs = _rand_xs(n, batch, 0x5EED0700 + (7 if scheme == "dcf" else 0))
    sd, cd = s0s[0].cuda(), cws.cuda()
    ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")
    step = lambda: eng.eval_points(sid, n, G, 0, sd, cd, xs, out=ys)
    _, kms = time_steady(step)
    rec = _sub_record(f"{'DCF (Boyle 21, mul=4 PRG, kLt)' if scheme == 'dcf' else 'DPF (Boyle 16)'} kInBits=32, 1 key, "
                      f"2^{log2_points} random eval points, 1 GPU (BASELINE configs[{2 if scheme == 'dcf' else 1}] at its "
                      f"literal size), party 0, Uint<uint64_t>; >= 150 ms of back-to-back launches after 60 ms of warm-up",
                      scheme, n, True, batch, kms * 1e-3, kms, 1, 1, eng.last_kernel()[0])
    rec["units_per_launch"] = batch
    rep = replayed_counters("dpf_2e20" if scheme == "dpf" else "dcf_2e20", batch)
    rec["roofline"].update({"traffic": rep["traffic"], "traffic_replayed_from": rep["traffic_replayed_from"],
                            "replay_matches_kernel_sources": rep["replay_matches_kernel_sources"]})
    if "lds_busy_pmc" in rep:
        rec["roofline"]["lds_busy_pmc"] = rep["lds_busy_pmc"]
    if with_cpu:
        from oracle import oracle as O
        p = O.Prg(2 if scheme == "dpf" else 4, use_evp=True)
        osid = O.DPF if scheme == "dpf" else O.DCF
        x64 = xs.cpu().numpy().view(np.uint32).astype(np.uint64)
        threads = usable_cores()
        res = {}

        def all_threads():
            res["y"] = O.eval_points(p, osid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), x64, nthreads=threads)
        reps, dt = _timed_cpu(all_threads, 1.0)
        m1 = 1 << 17
        t1 = time.perf_counter()
        O.eval_points(p, osid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), x64[:m1], nthreads=1)
        dt1 = time.perf_counter() - t1
        ok = bool((res["y"] == ys.cpu().numpy()).all())
        rec["cpu_baseline"] = {"value": reps * batch / dt, "unit": "evals/s", "cores": threads, "kind": "port",
                               "sample": f"all 2^{log2_points} points of the timed batch x {reps} passes, party 0, oracle with OpenSSL "
                                         f"EVP AES-128-ECB PRG (reference call pattern), {threads} OpenMP threads, {dt:.1f} s; "
                                         f"single thread: {m1} points",
                               "single_thread_value": m1 / dt1, "gpu_matches_oracle_on_sample": ok}
        rec["gpu_matches_oracle_on_sample"] = ok
    return rec


def _e2e_pcie_record(eng, log2_batch):
    """The PCIe-inclusive rate SURVEY 8(d) asks to state separately (never `value`): x from and y back to pinned host
    memory around the headline kernel -- serial on one stream, and pipelined in 8 chunks over three streams."""
    import fss_amd
    G, n, N, chunks = fss_amd.U64, 32, 1 << log2_batch, 8
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

    streams = [torch.cuda.Stream() for _ in range(3)]

    def pipelined():
        c_sz = N // chunks
        for c in range(chunks):
            sl = slice(c * c_sz, (c + 1) * c_sz)
            with torch.cuda.stream(streams[c % 3]):
                xs_d[sl].copy_(xs_h[sl], non_blocking=True)
                eng.eval_points(sid, n, G, 0, sd, cd, xs_d[sl], out=ys_d[sl])
                ys_h[sl].copy_(ys_d[sl], non_blocking=True)

    out = {"workload": f"the headline batch (DPF kInBits=32, 1 key, 2^{log2_batch} points) with x read from and y written back to "
                       "pinned host memory: 4 B up + 16 B down per evaluation; NOT the metric's value (inputs are not resident)",
           "unit": "evals/s"}
    reps = 5
    for key, fn in (("serial_one_stream", serial), (f"pipelined_{chunks}_chunks_3_streams", pipelined)):
        torch.cuda.synchronize()
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[key] = {"value": N / dt, "ms_per_batch": dt * 1e3, "pcie_GBs_both_directions": N * 20 / dt / 1e9}
    ref = torch.empty_like(ys_d)
    eng.eval_points(sid, n, G, 0, sd, cd, xs_d, out=ref)
    out["outputs_identical_to_resident_run"] = bool((ys_h.cuda() == ref).all())
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
    O.dpf_eval_all_repeat(p, n, O.BYTES, 0, seeds[0], cws, threads, reps)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    O.dpf_eval_all_repeat(p, n, O.BYTES, 0, seeds[0], cws, 1, reps)
    dt1 = time.perf_counter() - t1
    sd, cd = torch.from_numpy(seeds[0]).cuda(), torch.from_numpy(cws).cuda()
    yg = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
    _, gms = time_steady(lambda: eng.eval_all(fss_amd.DPF, n, fss_amd.BYTES, 0, sd, cd, out=yg), steps=50, min_ms=50.0, warm_ms=20.0)
    return {"workload": "DPF kInBits=8, 1 key, full-domain 256-point EvalAll, Bytes, party 0 (BASELINE configs[0]: the reference's "
                        "CPU-runnable case, samples/dpf_dcf_cpu.cu constants)",
            "cpu_baseline": {"value": reps * 256 / dt, "unit": "leaves/s", "cores": threads, "kind": "port",
                             "us_per_evalall": dt / reps * 1e6,
                             "sample": f"{reps} EvalAll calls: oracle in the reference's task-recursion shape, OpenSSL EVP PRG, "
                                       f"{threads} threads; single thread: {reps} calls",
                             "single_thread_value": reps * 256 / dt1, "single_thread_us_per_evalall": dt1 / reps * 1e6,
                             "oracle_matches_reference_sha256": bool(ok_kat), "gpu_matches_oracle_on_sample": bool(ok_gpu and ok_kat)},
            "gpu_us_per_evalall": gms * 1e3, "value": 256 / (gms * 1e-3), "unit": "leaves/s",
            "gpu_matches_oracle_on_sample": bool(ok_gpu and ok_kat)}


def run_headline(args, rank, world, dist):
    import fss_amd
    eng = fss_amd.default_engine(torch.cuda.current_device())
    n, scheme = 32, "dpf"
    batch = 1 << args.log2_batch
    G = fss_amd.U64
    sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, G, 0x5EED0002)
    xs = _rand_xs(n, batch, 0x5EED0002 + rank)
    seed_d, cws_d = s0s[0].cuda(), cws.cuda()
    ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")

    def step():
        eng.eval_points(sid, n, G, 0, seed_d, cws_d, xs, out=ys)

    wall, kernel_ms = time_steps(step, args.steps, args.warmup, dist)
    kname, grid, block = eng.last_kernel()

    # Same batch through the shared-prefix entry point (identical outputs; the top log2(batch) levels
    # of the one key's tree are expanded once per step instead of once per point).  Reported beside
    # the headline, never as `value`: it removes AES work rather than doing it faster.
    ys2 = torch.empty_like(ys)

    def step_shared():
        eng.eval_points_shared(sid, n, G, 0, seed_d, cws_d, xs, out=ys2)

    wall_sp, kernel_ms_sp = time_steps(step_shared, args.steps, args.warmup, dist)
    same = bool((ys == ys2).all())
    del ys2

    # ---- the rest of the metric: DCF, independent keys, kInBits 16 / 64 (north_star), same protocol
    subs, checks = {}, {}
    m = min(batch, 4096)  # evaluations per sub-record handed to the CPU check (a strided sample)
    pick = torch.arange(0, batch, max(1, batch // m), device="cuda")[:m]
    for key, sch, nn in (() if args.only_headline else (("dcf_points", "dcf", 32), ("dpf_n16", "dpf", 16), ("dpf_n64", "dpf", 64))):
        sid2, s0s2, cws2, ocw2, _, _ = make_key(eng, sch, nn, G, 0x5EED0100 + nn + (7 if sch == "dcf" else 0))
        xs2 = _rand_xs(nn, batch, 0x5EED0200 + nn + rank)
        sd2, cd2 = s0s2[0].cuda(), cws2.cuda()
        step2 = lambda: eng.eval_points(sid2, nn, G, 0, sd2, cd2, xs2, out=ys)
        w2, k2 = time_steps(step2, args.steps, args.warmup, dist)
        subs[key] = _sub_record(f"{'DCF (Boyle 21, kLt)' if sch == 'dcf' else 'DPF'} kInBits={nn}, 1 key, 2^{args.log2_batch} random "
                                f"points per GPU per step, party 0, Uint<uint64_t>", sch, nn, True, batch, w2, k2, world,
                                args.steps, eng.last_kernel()[0])
        checks[key] = (sch, nn, s0s2, cws2, None, xs2[pick].cpu().numpy(), ys[pick].cpu().numpy())
        del xs2
    if not args.only_headline:
        _keys_subrecord(args, rank, world, dist, eng, G, batch, ys, pick, subs, checks)
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
                subs[key] = _small_batch_record(eng, sch, 20, with_cpu)
        for name in ("evalall_c4", "ht_c5", "grotto_c5"):
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
    xs, ys = head_xs, head_ys
    _print_headline(args, world, batch, scheme, n, s0s, cws, ocw, xs, ys, wall, kernel_ms, wall_sp, kernel_ms_sp, same,
                    kname, grid, block, subs, checks, dist.get_backend() if dist is not None else None)


def _keys_subrecord(args, rank, world, dist, eng, G, batch, ys, pick, subs, checks):
    import fss_amd
    # M(ii): 2^batch independent keys x 1 point, keys generated on the device straight into the level-major layout
    # (outside the timed region, as the reference excludes Gen and relayout: src/bench_gpu.cu:616-627)
    gen = torch.Generator(device="cuda").manual_seed(0x5EED0300 + rank)
    ks0s = torch.randint(-2**31, 2**31, (batch, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    kbetas = torch.randint(-2**31, 2**31, (batch, 4), dtype=torch.int32, device="cuda", generator=gen)
    kalphas = torch.randint(-2**31, 2**31, (batch,), dtype=torch.int32, device="cuda", generator=gen)
    kxs = _rand_xs(32, batch, 0x5EED0400 + rank)
    kxs[::2] = kalphas[::2]  # half of the points hit their key's alpha
    keys, _ = eng.gen_level_major(fss_amd.DPF, 32, G, ks0s, kalphas, kbetas)
    kseeds = ks0s[:, 0].contiguous()
    step_k = lambda: eng.eval_keys(keys, G, 0, kseeds, kxs, out=ys)
    w3, k3 = time_steps(step_k, args.steps, args.warmup, dist)
    subs["dpf_keys"] = _sub_record(f"DPF kInBits=32, 2^{args.log2_batch} independent level-major keys x 1 point each per GPU per "
                                   f"step (SURVEY 8d M(ii)), party 0, Uint<uint64_t>", "dpf", 32, False, batch, w3, k3, world,
                                   args.steps, eng.last_kernel()[0])
    if rank == 0 and world == 1 and not args.no_cpu:
        from fss_amd.engine import LevelMajorKeys
        sub = LevelMajorKeys(fss_amd.DPF, 32, int(pick.numel()), keys.cw_s[:, pick].contiguous(),
                             tr_bits=keys.tr_bits[pick].contiguous(), out_cw=keys.out_cw[pick].contiguous())
        checks["dpf_keys"] = ("dpf_keys", 32, kseeds[pick].cpu().numpy(), eng.key_major(sub).cpu().numpy(), None,
                              kxs[pick].cpu().numpy(), ys[pick].cpu().numpy())
    del keys, ks0s, kbetas, kalphas, kseeds


def _print_headline(args, world, batch, scheme, n, s0s, cws, ocw, xs, ys, wall, kernel_ms, wall_sp, kernel_ms_sp, same,
                    kname, grid, block, subs, checks, process_group=None):
    evals = batch * world * args.steps
    value = evals / wall
    per_gpu_kernel_rate = batch / (kernel_ms * 1e-3)
    aes = aes_blocks_per_eval(scheme, n)
    lds_bytes = batch * aes * LOOKUP_BYTES_PER_AES
    lds_gbs = lds_bytes / (kernel_ms * 1e-3) / 1e9
    hbm_bytes = batch * hbm_bytes_per_eval(scheme, n, True)
    hbm_gbs = hbm_bytes / (kernel_ms * 1e-3) / 1e9
    pipes = measured_pipe_busy(args.log2_batch)
    replay_v, replay_src = _headline_pmc(args.log2_batch)
    from fss_amd import _build
    # True only when the profile was captured for exactly the sources this library's point walk is compiled from
    replay_ok = bool(replay_v) and replay_v.get("_captured_for") == _build.headline_digest()
    out = {
        "metric": "DPF/DCF evals/sec at kInBits=32, batch=2^24", "value": value, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"DPF (Boyle'16) kInBits=32, 1 key, 2^{args.log2_batch} uniformly random eval points "
                               "per GPU per step (configs[1] shape at the metric's batch), party 0, Uint<uint64_t>",
                   "kernel": kname, "grid": grid, "block": block, "parallelism": f"batch-sharded x{world}, no collective",
                   "process_group": process_group},
        "roofline": {"bound": "lds", "achieved": lds_gbs, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                     "frac": lds_gbs / LDS_PEAK_GBS, "traffic": measured_hbm_traffic(args.log2_batch),
                     "traffic_replayed_from": replay_src, "replay_matches_kernel_sources": replay_ok,
                     "kernel_ms": kernel_ms, "per_gpu_evals_per_s": per_gpu_kernel_rate,
                     "algorithmic": f"{aes} AES blocks/eval x 160 LDS lookups x 4 B; peak = 128 B/clk/CU x 256 CU x 2.4 GHz",
                     "aes_blocks_per_s": per_gpu_kernel_rate * aes, "pipes": pipes},
        "roofline_hbm": {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": measured_hbm_traffic(args.log2_batch),
                         "traffic_replayed_from": replay_src,
                         "algorithmic_bytes_per_launch": hbm_bytes,
                         "algorithmic": "4 B x + 16 B y per eval (key resident on chip); traffic = PMC bytes per "
                                        "launch of the committed profile named in traffic_replayed_from "
                                        "(FETCH_SIZE x2 + WRITE_SIZE), not of this run"},
    }
    if pipes:
        # third roof SURVEY 8(d) asks for: VALU issue.  Peak = 256 CU x 4 SIMD x one wave64 instruction per 2 clk x
        # 2.4 GHz (the fast-class rate); this kernel's instruction mix issues at ~4 clk, so the
        # pipe is ~0.9 busy at less than half of that peak.  Instruction count replayed from the profile, time live.
        valu_rate = pipes["valu_wave_instructions_per_launch"] / (kernel_ms * 1e-3)
        out["roofline_valu"] = {"bound": "valu", "achieved": valu_rate / 1e12, "peak": 1.2288, "unit": "T wave64-instr/s",
                                "frac": valu_rate / 1.2288e12, "busy_frac_pmc": pipes["valu_busy"],
                                "instructions_per_launch": pipes["valu_wave_instructions_per_launch"],
                                "replayed_from": pipes["replayed_from"]}
    out["shared_prefix"] = {
        "value": batch * world * args.steps / wall_sp, "unit": "evals/s", "ms_per_step": wall_sp / args.steps * 1e3,
        "kernel_ms": kernel_ms_sp, "identical_to_plain_walk": same,
        "note": "fss_amd_dpf_eval_points_shared: top log2(batch) tree levels expanded once per step (EvalAll-style), "
                "then (in_bits - log2(batch)) levels per point; ~10 AES blocks executed per eval instead of 32",
    }
    out.update(subs)
    if world == 1 and not args.no_cpu:
        threads = usable_cores()
        mm = min(batch, args.cpu_sample)
        out["cpu_baseline"] = cpu_baseline(scheme, n, s0s, cws, ocw, xs[:mm], ys[:mm], threads)
        for key, ok in cpu_check_subrecords(checks).items():  # part of the cpu_baseline leg: the oracle as checker
            out[key]["gpu_matches_oracle_on_sample"] = ok
        gates = {k: v["gpu_matches_oracle_on_sample"] for k, v in out.items()
                 if isinstance(v, dict) and "gpu_matches_oracle_on_sample" in v}
        out["parity_gates"] = {"all_true": all(gates.values()), "gates": gates}
    print(json.dumps(out), flush=True)


def cpu_check_subrecords(checks):
    """The sub-records' parity gates, run with the cpu_baseline leg: a strided sample of every sub-workload's GPU
    outputs against the oracle (party 0, Uint<uint64_t>)."""
    from oracle import oracle as O
    res = {}
    for key, (sch, n, s0s, cws, ocw, xs, got) in checks.items():
        xs64 = (xs.view(np.uint32) if xs.dtype == np.int32 else xs.view(np.uint64)).astype(np.uint64)
        if sch == "dpf_keys":
            want = O.eval_keys(O.Prg(2), O.DPF, n, O.U64, 0, s0s, cws, xs64, nthreads=4)
        else:
            sid = {"dpf": O.DPF, "dcf": O.DCF}[sch]
            want = O.eval_points(O.Prg(2 if sch == "dpf" else 4), sid, n, O.U64, 0, s0s[0].numpy(), cws.numpy(), xs64, nthreads=4)
        res[key] = bool((want == got).all())
    return res


def _workload_pmc(name):
    """Counters of one workload's dominant kernel REPLAYED from profiles/r<N>/pmc_workloads.json (highest N first; written
    by tools/profile_workloads.sh from separate rocprofv3 --pmc passes of `bench.py --workload <name>`): bench.py cannot
    read PMCs itself.  Returns (record, path relative to the repo) or (None, None)."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_workloads.json"))
    rnd = lambda f: int(re.search(r"profiles[/\\]r(\d+)", f).group(1))
    for f in sorted(files, key=rnd, reverse=True):
        try:
            d = json.load(open(f))
            if name in d:
                return d[name], os.path.relpath(f, ROOT)
        except Exception:
            pass
    return None, None


def replayed_counters(name, units_per_launch):
    """HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes, MI355X_MICROARCH.md's gfx950 correction) and LDS busy
    fraction per launch of the committed profile of this workload -- valid only for the launch size profiled and
    labelled with whether the profile was captured for the sources this library's kernel is compiled from."""
    rec, src = _workload_pmc(name)
    out = {"traffic": None, "traffic_replayed_from": src, "replay_matches_kernel_sources": False}
    if not rec or rec.get("units_per_launch") != units_per_launch:
        return out
    from fss_amd import _build
    out["replay_matches_kernel_sources"] = rec.get("_captured_for") == _build.workload_digest(name)
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        out["traffic"] = (2 * rec["FETCH_SIZE"]["mean_per_launch"] + rec["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    if "GRBM_GUI_ACTIVE" in rec and "SQ_LDS_IDX_ACTIVE" in rec:
        out["lds_busy_pmc"] = rec["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"] / 256.0 / (rec["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0)
    return out


def _timed_cpu(fn, min_s, max_reps=64):
    """fn() repeated until min_s seconds have been spent: (repetitions, seconds)."""
    reps, dt = 0, 0.0
    while dt < min_s and reps < max_reps:
        t0 = time.perf_counter()
        fn()
        dt += time.perf_counter() - t0
        reps += 1
    return reps, dt


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
    return {"value": reps * (1 << n_cpu) / dt, "unit": "leaves/s", "cores": threads, "kind": "port",
            "sample": f"DPF EvalAll kInBits={n_cpu} (2^{n_cpu} leaves) x {reps} passes, party 0, Uint<uint64_t>: oracle in the "
                      f"reference's task-recursion shape (one OpenMP task per child above ceil(log2(threads)) levels), OpenSSL "
                      f"EVP AES-128-ECB PRG, {threads} threads, {dt:.1f} s; single thread: one pass, {dt1:.1f} s",
            "single_thread_value": (1 << n_cpu) / dt1,
            "gpu_matches_oracle_on_sample": parity and bool((out == got).all()),
            "gpu_sample": f"the GPU's EvalAll of the same kInBits={n_cpu} key, all 2^{n_cpu} leaves"}


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
    return {"value": reps * m / dt, "unit": "evals/s", "cores": threads, "kind": "port",
            "sample": f"{m} of the timed batch's keys (every {max(1, nk // m)}th, brought back key-major) x {reps} passes, party 0: "
                      f"oracle {'HalfTreeDpf::Eval (Aes128Mmo<1>)' if scheme == 'ht' else 'Grotto path walk (both children per level)'}"
                      f" with the OpenSSL EVP AES-128-ECB PRG, {threads} OpenMP threads, {dt:.1f} s; single thread: {m1} keys",
            "single_thread_value": m1 / dt1,
            "gpu_matches_oracle_on_sample": bool((want == got).all())}


def sharded_record(args, name, rank, world, dist, eng, with_cpu):
    """BASELINE.json configs[3] / configs[4] under the same launch contract (barrier + synchronize, max over ranks):
      evalall_c4: DPF EvalAll kInBits=--log2-domain (28); the domain is cut into `world` aligned blocks
                  (shard.domain_block), rank r expands block r into its own HBM (strong scaling, no collective);
      ht_c5 / grotto_c5: kInBits=64, 2^--log2-keys (24) independent keys x 1 point, key batch split across the ranks
                  (shard.batch_slice; strong).
    On one GPU with the CPU leg enabled the record also carries cpu_baseline (and its parity gate).  Returns the
    record (meaningful on rank 0) and the outputs (for callers that hash them)."""
    import fss_amd
    from fss_amd import shard
    G = fss_amd.U64
    cpu = None
    if name == "evalall_c4":
        n = args.log2_domain
        sid, s0s, cws, ocw, _, _ = make_key(eng, "dpf", n, G, 0x5EED0004)
        first, nleaves = shard.domain_block(n, rank, world)
        sd, cd = s0s[0].cuda(), cws.cuda()
        ys = torch.empty((nleaves, 4), dtype=torch.int32, device="cuda")
        step = lambda: eng.eval_all(sid, n, G, 0, sd, cd, first_leaf=first, nleaves=nleaves, out=ys)
        units_total, units_rank, unit, aes_per_unit = 1 << n, nleaves, "leaves/s", 2.0
        desc = (f"DPF full-domain EvalAll kInBits={n} (2^{n} outputs), domain-sharded x{world}: "
                f"2^{nleaves.bit_length() - 1} leaves per GPU, Uint<uint64_t>, party 0")
        hbm_unit, hbm_note = 16, "16 B written per leaf (SURVEY 8d); the PMC traffic also holds the frontier nodes written and read back"
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
        units_total, units_rank, unit, aes_per_unit = 1 << n, nleaves, "leaves/s", 2.0
        desc = (f"Grotto DCF full-domain EvalAll kInBits={n}, domain-sharded x{world} with the cross-block parity carry "
                f"(one byte per rank all-gathered per step), 1 B per leaf, party 0")
        hbm_unit, hbm_note = 1, "1 B written per leaf"
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
        step = lambda: eng.eval_keys(keys, gg, 0, seeds0, xs, ocws=ocws, out=ys)
        units_total, units_rank, unit = total, nk, "evals/s"
        aes_per_unit = 64.0 if scheme == "ht" else 96.0
        desc = (f"{'Half-Tree DPF' if scheme == 'ht' else 'Grotto DCF (path walk)'} kInBits=64, 2^{args.log2_keys} independent "
                f"level-major keys x 1 point each, key batch sharded x{world} ({nk} keys per GPU), party 0")
        hbm_unit = 1084 if scheme == "ht" else 1057
        hbm_note = ("16 seed + 8 x + 64 x 16 cw_s + 4 extra + 16 ocw + 16 y per key" if scheme == "ht"
                    else "16 seed + 8 x + 64 x 16 cw_s + 8 tr bits + 1 y per key")
    pre_warm(warm_step if name == "grotto_evalall" else step)
    wall, kernel_ms = time_steps(step, args.steps, args.warmup, dist)
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
            cpu["timed_outputs_match_oracle_on_strided_sample"] = bool((ys[idx].cpu().numpy() == want).all())
            cpu["gpu_matches_oracle_on_sample"] = bool(cpu["gpu_matches_oracle_on_sample"] and cpu["timed_outputs_match_oracle_on_strided_sample"])
        elif name in ("ht_c5", "grotto_c5"):
            cpu = cpu_keys(eng, scheme, n, keys, seeds0, xs, ocws, ys, threads, 1 << 18)
    rate = units_rank / (kernel_ms * 1e-3)
    lds_gbs = rate * aes_per_unit * LOOKUP_BYTES_PER_AES / 1e9
    rep = replayed_counters(name, units_rank)
    rec = {
        "workload": desc, "value": units_total * args.steps / wall, "unit": unit, "ms_per_step": wall / args.steps * 1e3,
        "kernel_ms": kernel_ms, "per_gpu_units_per_s": rate, "units_per_launch": units_rank, "scaling": "strong",
        "kernel": kname, "grid": grid, "block": block,
        "aes_blocks_per_unit": aes_per_unit,
        "roofline": {"bound": "lds", "achieved": lds_gbs, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": lds_gbs / LDS_PEAK_GBS,
                     "traffic": rep["traffic"], "traffic_replayed_from": rep["traffic_replayed_from"],
                     "replay_matches_kernel_sources": rep["replay_matches_kernel_sources"], "kernel_ms": kernel_ms,
                     "algorithmic": f"{aes_per_unit:g} AES blocks per unit x 160 LDS lookups x 4 B (rank 0's share of {units_rank} units)"},
        "roofline_hbm": {"bound": "hbm", "achieved": rate * hbm_unit / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": rate * hbm_unit / 1e9 / HBM_PEAK_GBS, "traffic": rep["traffic"],
                         "traffic_replayed_from": rep["traffic_replayed_from"],
                         "algorithmic_bytes_per_launch": units_rank * hbm_unit, "algorithmic": hbm_note},
    }
    if "lds_busy_pmc" in rep:
        rec["roofline"]["lds_busy_pmc"] = rep["lds_busy_pmc"]
    if cpu is not None:
        rec["cpu_baseline"] = cpu
        rec["gpu_matches_oracle_on_sample"] = cpu["gpu_matches_oracle_on_sample"]
    return rec, ys


def run_sharded(args, rank, world, dist):
    """`--workload evalall_c4 | ht_c5 | grotto_c5`: one of configs[3] / configs[4] as the whole run, one JSON line."""
    import fss_amd
    eng = fss_amd.default_engine(torch.cuda.current_device())
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
    line = {"metric": f"{args.workload}: {rec['unit'][:-2]} per second", "value": rec["value"], "unit": rec["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": rec["workload"], "kernel": rec["kernel"], "grid": rec["grid"], "block": rec["block"],
                       "units_per_launch": rec["units_per_launch"], "parallelism": f"sharded x{world}, no collective",
                       "process_group": dist.get_backend() if dist is not None else None},
            "roofline": rec["roofline"], "roofline_hbm": rec["roofline_hbm"]}
    if "cpu_baseline" in rec:
        line["cpu_baseline"] = rec["cpu_baseline"]
    print(json.dumps(line), flush=True)


def run_suite(args):
    """Other BASELINE configs on one GPU (kernel-only rates, HIP events)."""
    import fss_amd
    eng = fss_amd.default_engine(0)
    res = []

    def record(name, units, fn, aes_per_unit, hbm_per_unit, steps=10, e=None, chacha_per_unit=0):
        wall, kms = time_steady(fn, steps)
        rate = units / (kms * 1e-3)
        res.append({"config": name, "units_per_launch": units, "kernel_ms": kms, "units_per_s": rate,
                    "aes_blocks_per_s": rate * aes_per_unit,
                    "lds_frac": rate * aes_per_unit * LOOKUP_BYTES_PER_AES / 1e9 / LDS_PEAK_GBS,
                    "hbm_gbs_algorithmic": rate * hbm_per_unit / 1e9, "kernel": (e or eng).last_kernel()})
        if chacha_per_unit:
            # ChaCha lines: neither LDS (unused) nor HBM (< 1 %) binds them but VALU issue: a ChaCha20
            # block is 10 double rounds x 96 instructions (32 add, 32 xor, 32 rotate), 1 003 wave-instructions per
            # wave-block with the feed-forward and the walk (SQ_INSTS_VALU, profiles/r1/chacha_dpf_n32_counters.txt).
            # peak = the chip's VALU rate, 256 CU x 4 SIMD x 32 lanes per clock = one wave64 instruction per 2 clocks
            # and SIMD (MI355X_MICROARCH.md).  The rotates are slow-class instructions and drag the whole stream to
            # ~3.8 clocks per instruction, so ~0.5 of this peak is where an add-rotate-xor cipher ends.
            wave_instr = rate * chacha_per_unit / 64.0 * CHACHA_VALU_PER_WAVE_BLOCK
            res[-1].pop("lds_frac")
            res[-1]["chacha_blocks_per_s"] = rate * chacha_per_unit
            res[-1]["roofline"] = {"bound": "valu_issue", "achieved": wave_instr / 1e9, "peak": VALU_ISSUE_PEAK_GINSTR,
                                   "unit": "G wave64-instr/s", "frac": wave_instr / 1e9 / VALU_ISSUE_PEAK_GINSTR,
                                   "algorithmic": f"{chacha_per_unit:g} ChaCha20 blocks per unit x {CHACHA_VALU_PER_WAVE_BLOCK} "
                                                  "wave-instructions per 64 blocks; peak = 256 CU x 4 SIMD x 2.4 GHz / 2 clk",
                                   "clk_per_instruction_and_simd": N_CU * 4 * CLK_HZ / wave_instr}
        print(json.dumps(res[-1]), file=sys.stderr, flush=True)

    G = fss_amd.U64
    for scheme, n, lb in (("dpf", 32, 10), ("dpf", 32, 16), ("dcf", 32, 16), ("dpf", 32, 20), ("dcf", 32, 20), ("dpf", 16, 24), ("dpf", 64, 24), ("dcf", 16, 24),
                          ("dcf", 32, 24), ("dcf", 64, 24), ("ht", 64, 24)):
        sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, G, 1 + n)
        batch = 1 << lb
        gen = torch.Generator(device="cuda").manual_seed(n)
        if n <= 32:
            xs = torch.randint(-2**31, 2**31, (batch,), dtype=torch.int32, device="cuda", generator=gen)
            if n < 32:
                xs &= (1 << n) - 1
        else:
            xs = torch.randint(-2**63, 2**63 - 1, (batch,), dtype=torch.int64, device="cuda", generator=gen)
        sd, cd = s0s[0].cuda(), cws.cuda()
        od = None if ocw is None else ocw.cuda()
        ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")
        record(f"{scheme} n={n} 1 key x 2^{lb} points", batch,
               lambda: eng.eval_points(sid, n, G, 0, sd, cd, xs, ocw=od, out=ys),
               aes_blocks_per_eval(scheme, n), hbm_bytes_per_eval(scheme, n, True))

    # one key, shared top of the tree expanded once per launch (same outputs, fewer AES blocks executed;
    # fractions here are ALGORITHMIC work / time and may exceed 1: the work is removed, not sped up)
    for scheme, n, lb in (("dpf", 32, 24), ("dpf", 32, 20), ("dcf", 32, 24), ("ht", 64, 24)):
        sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, G, 1 + n)
        batch = 1 << lb
        gen = torch.Generator(device="cuda").manual_seed(n)
        if n <= 32:
            xs = torch.randint(-2**31, 2**31, (batch,), dtype=torch.int32, device="cuda", generator=gen)
        else:
            xs = torch.randint(-2**63, 2**63 - 1, (batch,), dtype=torch.int64, device="cuda", generator=gen)
        sd, cd = s0s[0].cuda(), cws.cuda()
        od = None if ocw is None else ocw.cuda()
        ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")
        record(f"{scheme} n={n} 1 key x 2^{lb} points, shared prefix", batch,
               lambda: eng.eval_points_shared(sid, n, G, 0, sd, cd, xs, ocw=od, out=ys),
               aes_blocks_per_eval(scheme, n), hbm_bytes_per_eval(scheme, n, True))

    # independent keys, one point each: keys generated on the device straight into the
    # level-major layout (no key-major intermediate, no relayout); Gen is timed as its own line
    for scheme, n, lk in (("dpf", 32, 24), ("dcf", 32, 24), ("ht", 64, 24), ("grotto", 64, 24)):
        nk = 1 << lk
        gen = torch.Generator(device="cuda").manual_seed(100 + n)
        s0s = torch.randint(-2**31, 2**31, (nk, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
        betas = torch.randint(-2**31, 2**31, (nk, 4), dtype=torch.int32, device="cuda", generator=gen)
        if scheme == "grotto":
            betas.zero_()
        if n <= 32:
            alphas = torch.randint(-2**31, 2**31, (nk,), dtype=torch.int32, device="cuda", generator=gen)
            xs = torch.randint(-2**31, 2**31, (nk,), dtype=torch.int32, device="cuda", generator=gen)
        else:
            alphas = torch.randint(-2**63, 2**63 - 1, (nk,), dtype=torch.int64, device="cuda", generator=gen)
            xs = torch.randint(-2**63, 2**63 - 1, (nk,), dtype=torch.int64, device="cuda", generator=gen)
        sid = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF, "ht": fss_amd.HT, "grotto": fss_amd.GROTTO}[scheme]
        gg = fss_amd.BYTES if scheme == "grotto" else G
        keys, ocws = eng.gen_level_major(sid, n, gg, s0s, alphas, betas)
        gen_blocks = {"dpf": 4, "dcf": 8, "ht": 2, "grotto": 4}[scheme] * n
        gen_bytes = 48 + (4 if n <= 32 else 8) + n * (32 if scheme == "dcf" else 16) + 24
        record(f"{scheme} Gen n={n} 2^{lk} keys, level-major", nk,
               lambda: eng.gen_level_major(sid, n, gg, s0s, alphas, betas), gen_blocks, gen_bytes, steps=3)
        seeds0 = s0s[:, 0].contiguous()
        del s0s, betas, alphas
        if scheme == "grotto":
            ys = torch.empty((nk,), dtype=torch.uint8, device="cuda")
            record(f"grotto n={n} 2^{lk} independent keys x 1 point (path walk)", nk,
                   lambda: eng.eval_keys(keys, gg, 0, seeds0, xs, out=ys), 1.5 * n, 16 + 8 + n * 16 + 8 + 1)
        else:
            ys = torch.empty((nk, 4), dtype=torch.int32, device="cuda")
            record(f"{scheme} n={n} 2^{lk} independent keys x 1 point", nk,
                   lambda: eng.eval_keys(keys, G, 0, seeds0, xs, ocws=ocws, out=ys),
                   aes_blocks_per_eval(scheme, n), hbm_bytes_per_eval(scheme, n, False))
        del keys, ys, seeds0, xs
        torch.cuda.empty_cache()

    # ChaCha PRG (the reference's published GPU numbers use it: README.md:384-388, n = 20)
    ceng = fss_amd.Engine(0, prg="chacha")
    for scheme, n, lb in (("dpf", 20, 24), ("dcf", 20, 24), ("ht", 20, 24), ("dpf", 32, 24)):
        sid, s0s, cws, ocw, _, _ = make_key(ceng, scheme, n, G, 3 + n)
        batch = 1 << lb
        gen = torch.Generator(device="cuda").manual_seed(n)
        xs = torch.randint(0, 1 << min(n, 31), (batch,), dtype=torch.int32, device="cuda", generator=gen)
        sd, cd = s0s[0].cuda(), cws.cuda()
        od = None if ocw is None else ocw.cuda()
        ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")
        record(f"chacha {scheme} n={n} 1 key x 2^{lb} points", batch,
               lambda: ceng.eval_points(sid, n, G, 0, sd, cd, xs, ocw=od, out=ys), 0, hbm_bytes_per_eval(scheme, n, True), e=ceng,
               chacha_per_unit=n)
    for scheme, n, lk in (("dpf", 20, 22), ("dcf", 20, 21), ("ht", 20, 22)):
        nk = 1 << lk
        gen = torch.Generator(device="cuda").manual_seed(200 + n)
        s0s = torch.randint(-2**31, 2**31, (nk, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
        betas = torch.randint(-2**31, 2**31, (nk, 4), dtype=torch.int32, device="cuda", generator=gen)
        alphas = torch.randint(0, 1 << n, (nk,), dtype=torch.int32, device="cuda", generator=gen)
        xs = torch.randint(0, 1 << n, (nk,), dtype=torch.int32, device="cuda", generator=gen)
        sid = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF, "ht": fss_amd.HT}[scheme]
        keys, ocws = ceng.gen_level_major(sid, n, G, s0s, alphas, betas)
        seeds0 = s0s[:, 0].contiguous()
        ys = torch.empty((nk, 4), dtype=torch.int32, device="cuda")
        record(f"chacha {scheme} n={n} 2^{lk} independent keys x 1 point", nk,
               lambda: ceng.eval_keys(keys, G, 0, seeds0, xs, ocws=ocws, out=ys), 0, hbm_bytes_per_eval(scheme, n, False), e=ceng,
               chacha_per_unit=n)
        del keys, ys
    # verifiable DPF: walk + one keyed-BLAKE3 XorHash per evaluation; the reference's GPU number
    # is ChaCha + BLAKE3 at n = 20 (README.md:389, BM_VdpfEvalPointGpu: 945 M eval/s on RTX PRO 5000)
    seng = fss_amd.Engine(0)  # AES-128-MMO + SHA-256: the pair of the reference's CPU VDPF rows (README.md:354: 2499 ns/eval at n=20)
    seng.set_hash_sha256([0x11111111, 0x22222222, 0x33333333, 0x44444444])
    for veng, tag, n, lk in ((ceng, "chacha+blake3", 20, 22), (eng, "aes+blake3", 32, 22), (seng, "aes+sha256", 20, 22)):
        nk = 1 << lk
        gen = torch.Generator(device="cuda").manual_seed(300 + n)
        s0s = torch.randint(-2**31, 2**31, (nk, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
        betas = torch.randint(-2**31, 2**31, (nk, 4), dtype=torch.int32, device="cuda", generator=gen)
        alphas = torch.randint(0, 1 << min(n, 31), (nk,), dtype=torch.int32, device="cuda", generator=gen)
        xs = torch.randint(0, 1 << min(n, 31), (nk,), dtype=torch.int32, device="cuda", generator=gen)
        keys, cs, ocws, _ = veng.vdpf_gen(n, G, s0s, alphas, betas, level_major=True)
        seeds0 = s0s[:, 0].contiguous()
        outs = (torch.empty((nk, 4), dtype=torch.int32, device="cuda"), torch.empty((nk, 4, 4), dtype=torch.int32, device="cuda"))
        record(f"{tag} vdpf n={n} 2^{lk} independent keys x 1 point (y + pi~)", nk,
               lambda: veng.vdpf_eval_keys(keys, G, 0, seeds0, cs, ocws, xs, out=outs),
               n if tag.startswith("aes") else 0, 16 + 4 + n * 16 + 8 + 64 + 16 + 16 + 64, e=veng)
        del keys, outs
    sid, s0s, cws, ocw, _, _ = make_key(ceng, "dpf", 24, G, 5)
    sd, cd = s0s[0].cuda(), cws.cuda()
    ys = torch.empty((1 << 24, 4), dtype=torch.int32, device="cuda")
    record("chacha dpf EvalAll n=24", 1 << 24, lambda: ceng.eval_all(sid, 24, G, 0, sd, cd, out=ys), 0, 16, steps=5, e=ceng,
           chacha_per_unit=1.0)  # one permutation per internal node = per leaf
    del ys

    # Grotto DCF full domain: the DPF tree over Bytes with beta = 0, leaf control bits, then the running-XOR scan
    # (GrottoDcf::EvalAll, grotto_dcf.cuh:151-163); one byte per leaf
    for n in (24, 26):
        gen = torch.Generator().manual_seed(77 + n)
        s0s = torch.randint(-2**31, 2**31, (1, 2, 4), dtype=torch.int32, generator=gen)
        alpha = torch.randint(0, 2**31, (1,), dtype=torch.int32, generator=gen) & ((1 << n) - 1)
        cws, _ = eng.gen(fss_amd.DPF, n, fss_amd.BYTES, s0s, alpha, torch.zeros((1, 4), dtype=torch.int32))
        sd, cd = s0s[0, 0].cuda(), cws[0]
        yb = torch.empty((1 << n,), dtype=torch.uint8, device="cuda")
        record(f"grotto EvalAll n={n}", 1 << n,
               lambda: eng.eval_all(fss_amd.GROTTO, n, fss_amd.BYTES, 0, sd, cd, out=yb), 2, 1, steps=5)
        del yb

    # raw PRG (Aes128Mmo<mul>::Gen, aes128_mmo.cuh:72-93; the reference's Aes128Soft<2>::Gen: 425.7 M Gen/s on A30,
    # doc/bench_aes128_soft.md:35-47): 16 B seed in, mul x 16 B out
    seeds = torch.randint(-2**31, 2**31, (1 << 24, 4), dtype=torch.int32, device="cuda")
    for beng, tag in ((eng, "aes"), (ceng, "chacha")):
        for mul in (1, 2, 4):
            record(f"{tag} prg Gen mul={mul}, 2^24 seeds", 1 << 24, lambda: beng.prg_gen(mul, seeds),
                   mul if tag == "aes" else 0, 16 + 16 * mul, steps=5, e=beng, chacha_per_unit=0 if tag == "aes" else 1.0)
    del seeds

    # key-major <-> level-major transpositions (fss::gpu::*RelayoutGpu, point_eval_gpu.cuh:324-375, and the inverse):
    # pure copies, HBM-bound (hbm_gbs_algorithmic / 8000 is their roofline fraction); one-time work in the
    # reference's benches (src/bench_gpu.cu:618-619)
    for scheme, n in (("dpf", 32), ("dcf", 32), ("ht", 64)):
        nk = 1 << 22
        sid = {"dpf": fss_amd.DPF, "dcf": fss_amd.DCF, "ht": fss_amd.HT}[scheme]
        slots = n if scheme == "ht" else n + 1
        km = torch.randint(-2**31, 2**31, (nk, slots, 8), dtype=torch.int32, device="cuda")
        moved = slots * 32 + n * (32 if scheme == "dcf" else 16) + 24
        record(f"{scheme} relayout n={n}, 2^22 keys (key-major -> level-major)", nk, lambda: eng.relayout(sid, n, km), 0, moved, steps=5)
        lm = eng.relayout(sid, n, km)
        record(f"{scheme} key_major n={n}, 2^22 keys (level-major -> key-major)", nk, lambda: eng.key_major(lm), 0, moved, steps=5)
        del km, lm

    # batched full domain (fss::gpu::*EvalAllGpuBatch; the reference's n = 20 rows, README.md:384-385: ChaCha,
    # DPF 15.43 and Half-Tree 12.09 G outputs/s on RTX PRO 5000)
    for beng, tag in ((eng, "aes"), (ceng, "chacha")):
        for scheme, n, nk in (("dpf", 20, 64), ("ht", 20, 64), ("dpf", 14, 4096)):
            gen = torch.Generator(device="cuda").manual_seed(400 + n)
            s0s = torch.randint(-2**31, 2**31, (nk, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
            betas = torch.randint(-2**31, 2**31, (nk, 4), dtype=torch.int32, device="cuda", generator=gen)
            alphas = torch.randint(0, 1 << n, (nk,), dtype=torch.int32, device="cuda", generator=gen)
            sid = {"dpf": fss_amd.DPF, "ht": fss_amd.HT}[scheme]
            cws, ocws = beng.gen(sid, n, G, s0s, alphas, betas)
            seeds0 = s0s[:, 0].contiguous()
            yb = torch.empty((nk, 1 << n, 4), dtype=torch.int32, device="cuda")
            record(f"{tag} {scheme} EvalAll batch n={n} x {nk} keys", nk << n,
                   lambda: beng.eval_all_batch(sid, n, G, 0, seeds0, cws, ocws=ocws, out=yb),
                   (2 if scheme == "dpf" else 1.5) if tag == "aes" else 0, 16, steps=5, e=beng,
                   chacha_per_unit=0 if tag == "aes" else (1.0 if scheme == "dpf" else 1.5))
            del yb

    # full domain
    for scheme, n in (("dpf", 24), ("dpf", 26), ("ht", 24)):
        sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, G, 7 + n)
        sd, cd = s0s[0].cuda(), cws.cuda()
        od = None if ocw is None else ocw.cuda()
        ys = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
        record(f"{scheme} EvalAll n={n}", 1 << n, lambda: eng.eval_all(sid, n, G, 0, sd, cd, ocw=od, out=ys),
               2.0 if scheme == "dpf" else 1.5, 16, steps=5)
        del ys
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_suite.json"), "w") as f:
        json.dump(res, f, indent=1)


def visible_gpus_without_hip():
    """GPUs this process could use, counted WITHOUT the HIP runtime (the launcher must not initialise it before it
    forks its ranks; torch.cuda.device_count() only avoids HIP while amdsmi discovery works): KFD topology nodes with
    SIMDs, cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES lists.  None = cannot tell
    (no sysfs): then there is no pre-check and a rank without a device fails by itself, which ends the run."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for f in nodes:
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except (OSError, ValueError):
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def _die_with_parent():
    """preexec_fn of the ranks: SIGTERM when the launcher dies, however it dies (PR_SET_PDEATHSIG)."""
    try:
        import ctypes
        import signal
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)  # 1 = PR_SET_PDEATHSIG
    except Exception:
        pass


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (fresh processes, spawned before
    this one has touched the GPU -- nothing here calls into HIP: the devices are counted from sysfs), one per GPU,
    rendezvous on 127.0.0.1.  Rank 0's
    stdout (the one JSON line) passes through.  If a rank exits non-zero, or the run outlives --launch-timeout, the
    remaining ranks are killed and the exit code is non-zero: a failed rank can never leave the others in a barrier."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    share = os.environ.get("FSS_AMD_BENCH_SHARE_GPU") == "1"
    dry = os.environ.get("FSS_AMD_BENCH_DRY_RUN") == "1"
    if not (share or dry):
        have = visible_gpus_without_hip()
        if have is not None and have < n:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible")

    # SIGTERM / SIGHUP / SIGINT to the launcher (`timeout N python bench.py --gpus 8`, a harness kill) must unwind
    # into the `finally` below: the ranks sit in their own sessions and would otherwise outlive it on the GPUs.
    def _on_signal(signum, _frame):
        raise SystemExit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
        signal.signal(sg, _on_signal)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, start_new_session=True,
                                      preexec_fn=_die_with_parent))
    deadline = time.monotonic() + args.launch_timeout
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            for pr in list(live):
                code = pr.poll()
                if code is None:
                    continue
                live.remove(pr)
                if code != 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {procs.index(pr)} exited with code {code}; stopping the other ranks", file=sys.stderr)
            if live and rc == 0:
                if time.monotonic() > deadline:
                    rc = 124
                    print(f"bench.py: ranks still running after {args.launch_timeout} s; stopping them", file=sys.stderr)
                else:
                    time.sleep(0.05)
    finally:
        for pr in procs:  # exactly the process groups started above
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, 15)
                except ProcessLookupError:
                    pass
        t_end = time.monotonic() + 5
        for pr in procs:
            try:
                pr.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, 9)
                except ProcessLookupError:
                    pass
    return rc


def run_dry(args, rank, world, dist):
    """FSS_AMD_BENCH_DRY_RUN=1: the launch / rendezvous / barrier / max-over-ranks / one-JSON-line control flow with an
    empty step and no GPU -- a test hook for boxes without one (tests/test_bench_launch.py).  Its line says
    "dry_run": true and measures nothing."""
    print(f"bench.py dry rank {rank} pid {os.getpid()}", file=sys.stderr, flush=True)
    fail = os.environ.get("FSS_AMD_BENCH_FAIL_RANK")
    if fail is not None and int(fail) == rank:
        raise SystemExit(f"rank {rank}: failure injected by FSS_AMD_BENCH_FAIL_RANK")
    hang = os.environ.get("FSS_AMD_BENCH_HANG_RANK")
    if hang is not None and int(hang) == rank:
        time.sleep(3600)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "DPF/DCF evals/sec at kInBits=32, batch=2^24", "value": 0.0, "unit": "evals/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / max(args.steps, 1) * 1e3,
                          "dry_run": True, "config": {"workload": "none (control-flow test hook, no GPU work)"}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log2-batch", type=int, default=24)
    ap.add_argument("--cpu-sample", type=int, default=(1 << 24) if (os.cpu_count() or 1) >= 8 else (1 << 22), help="points of the batch timed on the CPU")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--suite", action="store_true")
    ap.add_argument("--workload", default="dpf_points", choices=["dpf_points", "dpf_2e20", "dcf_2e20", "evalall_c4", "ht_c5", "grotto_c5", "grotto_evalall"],
                    help="dpf_points = the headline (BASELINE metric); the others are configs[3] / configs[4], sharded")
    ap.add_argument("--log2-domain", type=int, default=28, help="evalall_c4: kInBits of the full domain (configs[3]: 28)")
    ap.add_argument("--log2-keys", type=int, default=24, help="ht_c5 / grotto_c5: log2 of the key batch (configs[4]: 24)")
    ap.add_argument("--hash-out", default="", help="--workload runs: every rank writes the SHA-256 of its outputs to <this>.rank<r>")
    ap.add_argument("--hash-blocks", type=int, default=1, help="with --hash-out: digests of this many equal blocks of the rank's outputs")
    ap.add_argument("--only-headline", action="store_true",
                    help="skip the sub-records (DCF, independent keys, kInBits 16 / 64): the PMC passes of tools/profile.sh "
                         "use this so that one kernel template means one workload")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="self-launched multi-GPU runs: seconds before the ranks are stopped and the run fails")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # Not under a launcher but asked for several GPUs: become the launcher.  This must stay ahead of anything that
    # initialises the GPU (torch.cuda.is_available() does): the ranks are fresh child processes.
    if args.gpus > 1 and "RANK" not in os.environ:
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
        rec = _small_batch_record(fss_amd.default_engine(torch.cuda.current_device()), args.workload[:3], 20, not args.no_cpu)
        print(json.dumps({"metric": f"{args.workload}: evals per second", "value": rec["value"], "unit": "evals/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                          "config": {"workload": rec["workload"], "kernel": rec["kernel"], "units_per_launch": rec["units_per_launch"]},
                          "roofline": rec["roofline"], **({"cpu_baseline": rec["cpu_baseline"]} if "cpu_baseline" in rec else {})}),
              flush=True)
    else:
        run_sharded(args, rank, world, dist)
    if args.suite and world == 1 and not dry:
        run_suite(args)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
