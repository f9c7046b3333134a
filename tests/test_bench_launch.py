"""bench.py --gpus N started bare (no torch.distributed.run around it): the launcher spawns the ranks before
anything touches the GPU, passes rank 0's single JSON line through, and ends the run with a non-zero exit code when
a rank fails or hangs.  The CPU tests use the FSS_AMD_BENCH_DRY_RUN hook (empty step, gloo rendezvous): control
flow only, nothing is measured.  The GPU test runs the real kernels with two ranks sharing the one GPU of the box
(FSS_AMD_BENCH_SHARE_GPU: gloo instead of RCCL, which refuses two ranks on one device)."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env, timeout=240):
    e = dict(os.environ, **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)
    return r, time.monotonic() - t0


def json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def test_bare_gpus_2_prints_one_rank0_line():
    r, _ = run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"FSS_AMD_BENCH_DRY_RUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry_run"] is True


def test_under_a_launcher_env_it_is_a_rank_not_a_launcher():
    """With RANK / WORLD_SIZE set (torch.distributed.run's contract) bench.py must not spawn anything."""
    r, _ = run(["--gpus", "1", "--steps", "2"], {"FSS_AMD_BENCH_DRY_RUN": "1"})
    assert r.returncode == 0 and json_lines(r.stdout)[0]["n_gpus"] == 1


def test_a_failing_rank_ends_the_run_nonzero_and_fast():
    r, dt = run(["--gpus", "3", "--steps", "2"], {"FSS_AMD_BENCH_DRY_RUN": "1", "FSS_AMD_BENCH_FAIL_RANK": "2"})
    assert r.returncode != 0
    assert "rank 2 exited" in r.stderr
    assert not json_lines(r.stdout)
    assert dt < 120


def test_a_hung_rank_is_stopped_by_the_launch_timeout():
    r, dt = run(["--gpus", "2", "--steps", "2", "--launch-timeout", "8"],
                {"FSS_AMD_BENCH_DRY_RUN": "1", "FSS_AMD_BENCH_HANG_RANK": "1"}, timeout=200)
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert dt < 90


def test_bare_gpus_8_dry_run():
    """The driver's largest launch (8 ranks) through the launcher, rendezvous, barriers and the max-over-ranks
    reduction, without a GPU."""
    r, _ = run(["--gpus", "8", "--steps", "2", "--warmup", "1"], {"FSS_AMD_BENCH_DRY_RUN": "1"}, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["dry_run"] is True
    # config.ranks: eight entries, one per rank, eight different processes, each of which saw a process group of eight
    ranks = lines[0]["config"]["ranks"]
    assert sorted(r["rank"] for r in ranks) == list(range(8)) and len({r["pid"] for r in ranks}) == 8
    assert all(r["pg"] == ["gloo", 8] for r in ranks)
    assert lines[0]["scaling"] == "strong"


def test_sigterm_to_the_launcher_takes_the_ranks_down():
    """`timeout N python bench.py --gpus 8` sends SIGTERM to the launcher only; its ranks live in their own sessions.
    The launcher's handler must unwind into the clean-up that kills them (and PR_SET_PDEATHSIG covers SIGKILL)."""
    import re
    import signal
    e = dict(os.environ, FSS_AMD_BENCH_DRY_RUN="1", FSS_AMD_BENCH_HANG_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "2"], env=e, stdout=subprocess.DEVNULL,
                         stderr=subprocess.PIPE)
    pids, t0, buf = [], time.monotonic(), ""
    os.set_blocking(p.stderr.fileno(), False)
    while len(pids) < 2 and time.monotonic() - t0 < 120:
        try:
            buf += os.read(p.stderr.fileno(), 65536).decode(errors="replace")
        except BlockingIOError:
            pass
        pids = [int(m) for m in re.findall(r"dry rank \d+ pid (\d+)", buf)]
        time.sleep(0.1)
    assert len(pids) == 2, buf[-2000:]
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) != 0

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        try:  # a zombie of an already-reaped session still answers kill(0): look at its state
            return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
        except OSError:
            return False
    t1 = time.monotonic()
    while any(alive(q) for q in pids) and time.monotonic() - t1 < 20:
        time.sleep(0.2)
    assert not any(alive(q) for q in pids), pids


def test_device_count_does_not_need_hip():
    """The launcher counts GPUs from sysfs (or cannot tell), never through the HIP runtime."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    n = m.visible_gpus_without_hip()
    assert n is None or n >= 0
    import inspect
    assert "torch.cuda" not in inspect.getsource(m.launch_ranks)


@pytest.mark.gpu
def test_bare_gpus_2_on_one_gpu_with_the_share_hook():
    r, _ = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-batch", "20", "--no-cpu"],
               {"FSS_AMD_BENCH_SHARE_GPU": "1"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1
    d = lines[0]
    # the metric's ONE batch cut in two is `value`; every rank's own batch is the side record
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["value"] == d["headline_strong"]["v"] and d["ms_per_step"] == d["headline_strong"]["ms"]
    assert d["weak"]["v"] > 0 and d["weak"]["n"] == (1 << 20) and d["weak"]["scaling"] == "weak"
    assert d["config"]["kernel"] and "ONE 2^20-point batch" in d["config"]["workload"]
    assert d["roofline"]["kernel_ms"] == d["headline_strong"]["kms"]
    # what the ranks ran on: two entries, two ranks, a process group of two each (gloo under the share hook) -- and ONE GPU
    ranks = d["config"]["ranks"]
    assert len(ranks) == 2 and sorted(r["rank"] for r in ranks) == [0, 1]
    assert len({json.dumps(r, sort_keys=True) for r in ranks}) == 2
    assert all(r["pg"] == ["gloo", 2] for r in ranks) and all(r["dev"] == 0 for r in ranks)
    assert d["config"]["distinct_devices"] == 1
    for key in ("dcf_points", "dpf_keys", "dcf_keys", "dpf_n16", "dpf_n64", "dcf_n16", "dcf_n64", "dpf_keys_n16", "dpf_keys_n64"):
        assert d["sub"][key]["v"] > 0 and 0 < d["sub"][key]["f"] < 1.2, key
    # the one batch cut in two: half the points per rank, a rate of its own
    assert d["headline_strong"]["n"] == (1 << 19) and d["headline_strong"]["v"] > 0 and d["headline_strong"]["scaling"] == "strong"


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--workload", "evalall_c4", "--log2-domain", "22"],
                                   ["--workload", "grotto_evalall", "--log2-domain", "20"]])
def test_one_rank_under_torch_distributed_run_goes_through_rccl(extra):
    """The driver's multi-GPU command line with --nproc-per-node 1: the rank forms its process group over RCCL
    (backend "nccl"), passes the barriers either side of the timed region and the max-over-ranks reduction on a device
    tensor, and tears the group down -- the calls an 8-GPU run makes, on the one GPU of this box."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu",
           "--log2-batch", "20"] + extra
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # RCCL's banner and the launcher's notes stay off stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["process_group"] == "nccl"


@pytest.mark.gpu
def test_self_launcher_rccl_branch_with_one_rank():
    """bench.py's OWN launcher (bench_launch.launch_ranks, not torch.distributed.run) with one rank: the rank it spawns
    forms its process group over RCCL (backend "nccl"), gathers config.ranks through it, passes the barriers and the
    device-tensor reduction -- the branch an 8-GPU `python bench.py --gpus 8` takes, on the one GPU of this box."""
    r, _ = run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu", "--log2-batch", "20", "--only-headline"],
               {"FSS_AMD_BENCH_SELF_LAUNCH": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["process_group"] == "nccl"
    assert d["config"]["ranks"][0]["pg"] == ["nccl", 1] and d["config"]["ranks"][0]["dev"] == 0
    assert d["config"]["distinct_devices"] == 1 and d["ms_median"] > 0
    assert d["headline_strong"]["v"] == d["value"] and d["scaling"] == "weak" and "weak" not in d


@pytest.mark.gpu
def test_default_line_carries_the_whole_metric():
    """The driver's command at a small batch: the headline, every cell of the north-star grid (DCF, independent keys,
    kInBits 16 / 64 for both), BASELINE configs[0..4] and the PCIe-inclusive figure -- each with its own kernel time and
    LDS fraction, each checked against the oracle inside the cpu_baseline leg -- in ONE line short enough for the
    driver's record (< 7 KB), whose every key bench_legend.json explains."""
    r, _ = run(["--steps", "3", "--warmup", "1", "--log2-batch", "18", "--cpu-sample", "65536", "--log2-domain", "24",
                "--log2-keys", "20"], {}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    raw = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(raw) == 1 and len(raw[0]) < 7000, len(raw[0])
    d = json.loads(raw[0])
    sub = d["sub"]
    legend = json.load(open(os.path.join(ROOT, "bench_legend.json")))
    for name, rec in sub.items():
        assert name in legend["records"], name
        assert set(rec) <= set(legend["sub_record_keys"]), (name, set(rec) - set(legend["sub_record_keys"]))
    walks = ("dcf_points", "dpf_n16", "dpf_n64", "dcf_n16", "dcf_n64", "dpf_keys", "dcf_keys", "dpf_keys_n16", "dpf_keys_n64")
    for key in walks + ("dpf_2e20", "dcf_2e20", "evalall_c4", "ht_c5", "grotto_c5"):
        rec = sub[key]
        assert rec["ok"] is True, key
        assert rec["kms"] > 0 and 0 < rec["f"] < 1.2 and rec["v"] > 0, key
    for key in ("dpf_2e20", "dcf_2e20", "evalall_c4", "ht_c5", "grotto_c5", "c1_cpu"):
        assert sub[key]["cpu"] > 0 and sub[key]["cpu1"] > 0, key
    assert sub["dcf_points"]["a"] == 64 and sub["dpf_n16"]["a"] == 16 and sub["dcf_n64"]["a"] == 128 and sub["dcf_keys"]["a"] == 64
    assert sub["grotto_c5"]["fx"] < sub["grotto_c5"]["f"]
    assert sub["c1_cpu"]["kat"] is True and sub["c1_cpu"]["ok"] is True
    assert sub["e2e_pcie"]["same"] is True and sub["e2e_pcie"]["serial"] > 0 and sub["e2e_pcie"]["pipe"] > 0
    assert sub["shared_prefix"]["same"] is True
    assert d["gates"]["all"] is True and d["gates"]["n"] >= 18
    assert d["n_gpus"] == 1 and d["cpu_baseline"]["ok"] is True
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    # one GPU: the strong-scaled form of the metric is the headline launch itself
    assert d["headline_strong"]["v"] == d["value"] and d["headline_strong"]["n"] == 1 << 18
    assert d["scaling"] == "weak" and d["ms_median"] > 0 and abs(d["ms_median"] - d["roofline"]["kernel_ms"]) < 0.5 * d["ms_median"]
    assert d["config"]["ranks"][0]["rank"] == 0 and d["config"]["ranks"][0]["pg"] is None and d["config"]["distinct_devices"] == 1
    assert os.path.exists(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))


def _digests(prefix, ranks):
    out = []
    for r in range(ranks):
        with open(f"{prefix}.rank{r}") as f:
            out += [ln.split()[0] for ln in f if ln.strip()]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("workload,size", [("headline_strong", ["--log2-batch", "21"]),
                                           ("evalall_c4", ["--log2-domain", "22"]), ("ht_c5", ["--log2-keys", "18"]),
                                           ("grotto_c5", ["--log2-keys", "18"]), ("grotto_evalall", ["--log2-domain", "20"])])
def test_sharded_workloads_with_8_ranks_on_one_gpu(workload, size, tmp_path):
    """The metric's own batch split eight ways (headline_strong: 2^B / 8 points of the one key per rank) and
    BASELINE configs[3] / configs[4] as the driver's 8-GPU box would run them -- `bench.py --gpus 8 --workload ...`,
    the domain cut by shard.domain_block, the point and key batches by shard.batch_slice -- with the eight ranks sharing this box's
    one GPU (FSS_AMD_BENCH_SHARE_GPU: gloo instead of RCCL) at a reduced size.  Every rank writes the SHA-256 of its
    block of outputs; the eight digests must be the digests of the eight blocks of a single-rank run (for the Grotto
    full domain that includes the cross-block parity carry of shard.grotto_fix_shard)."""
    common = ["--workload", workload, *size, "--steps", "2", "--warmup", "1", "--no-cpu"]
    one = str(tmp_path / "one")
    r, _ = run(common + ["--hash-out", one, "--hash-blocks", "8"], {}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    want = _digests(one, 1)
    assert len(want) == 8 and len(set(want)) > 1
    eight = str(tmp_path / "eight")
    r, _ = run(["--gpus", "8"] + common + ["--hash-out", eight], {"FSS_AMD_BENCH_SHARE_GPU": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json_lines(r.stdout)
    assert len(d) == 1 and d[0]["n_gpus"] == 8 and d[0]["value"] > 0 and d[0]["scaling"] == "strong"
    assert _digests(eight, 8) == want
