"""bench.py's launcher: `python bench.py --gpus N` without torch.distributed.run around it starts its own N ranks (one
process per GPU, spawned before anything touches the GPU; the devices are counted from sysfs, never through HIP), stops
them all if one fails or hangs, and passes rank 0's one JSON line through.  run_dry is the GPU-less control-flow hook of
tests/test_bench_launch.py.  Nothing here is timed."""
import json
import os
import sys
import time

import torch

from bench_meta import METRIC

BENCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench.py")


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
        procs.append(subprocess.Popen([sys.executable, BENCH] + argv, env=env,
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
    # config.ranks as the real run gathers it (bench.rank_identities), minus the device: one entry per rank, each with the
    # process group as THAT rank sees it
    ident = {"rank": rank, "dev": None, "pid": os.getpid(), "pg": None if dist is None else [dist.get_backend(), dist.get_world_size()]}
    ranks = [ident]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, ident)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "evals/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / max(args.steps, 1) * 1e3,
                          "scaling": "strong" if world > 1 else "weak", "dry_run": True,
                          "config": {"workload": "none (control-flow test hook, no GPU work)", "ranks": ranks}}), flush=True)


