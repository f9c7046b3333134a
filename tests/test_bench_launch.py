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
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    for key in ("dcf_points", "dpf_keys", "dpf_n16", "dpf_n64"):
        assert d[key]["value"] > 0 and 0 < d[key]["roofline"]["frac"] < 1.2


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
def test_default_line_carries_the_whole_metric():
    """The driver's command at a small batch: headline + DCF / independent keys / kInBits 16, 64, each with its own
    kernel time and LDS fraction, each checked against the oracle on a sample inside the cpu_baseline leg."""
    r, _ = run(["--steps", "3", "--warmup", "1", "--log2-batch", "18", "--cpu-sample", "65536", "--log2-domain", "24",
                "--log2-keys", "20"], {}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json_lines(r.stdout)[0]
    # round 3: BASELINE configs[0], [1], [2] at their literal sizes and configs[3], [4] (here at reduced sizes), each
    # with its own CPU leg and parity gate, and the PCIe-inclusive figure, all in the one line
    for key in ("dpf_2e20", "dcf_2e20", "evalall_c4", "ht_c5", "grotto_c5", "c1_cpu"):
        rec = d[key]
        assert rec["gpu_matches_oracle_on_sample"] is True, key
        cb = rec["cpu_baseline"]
        assert cb["value"] > 0 and cb["single_thread_value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port", key
    for key in ("dpf_2e20", "dcf_2e20", "evalall_c4", "ht_c5", "grotto_c5"):
        assert d[key]["kernel_ms"] > 0 and 0 < d[key]["roofline"]["frac"] < 1.2, key
        assert "traffic" in d[key]["roofline"] and "replay_matches_kernel_sources" in d[key]["roofline"], key
    assert d["evalall_c4"]["cpu_baseline"]["timed_outputs_match_oracle_on_strided_sample"] is True
    assert d["c1_cpu"]["cpu_baseline"]["oracle_matches_reference_sha256"] is True
    assert d["e2e_pcie"]["outputs_identical_to_resident_run"] is True and d["e2e_pcie"]["serial_one_stream"]["value"] > 0
    assert d["parity_gates"]["all_true"] is True and len(d["parity_gates"]["gates"]) >= 11
    assert d["n_gpus"] == 1 and d["cpu_baseline"]["gpu_matches_oracle_on_sample"] is True
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    for key in ("dcf_points", "dpf_keys", "dpf_n16", "dpf_n64"):
        rec = d[key]
        assert rec["gpu_matches_oracle_on_sample"] is True, key
        assert rec["kernel_ms"] > 0 and rec["roofline"]["bound"] == "lds"
    assert d["dcf_points"]["aes_blocks_per_eval"] == 64 and d["dpf_n16"]["aes_blocks_per_eval"] == 16
    assert d["shared_prefix"]["identical_to_plain_walk"] is True
    assert "clock_ghz_in_pmc_run" not in json.dumps(d)


def _digests(prefix, ranks):
    out = []
    for r in range(ranks):
        with open(f"{prefix}.rank{r}") as f:
            out += [ln.split()[0] for ln in f if ln.strip()]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("workload,size", [("evalall_c4", ["--log2-domain", "22"]), ("ht_c5", ["--log2-keys", "18"]),
                                           ("grotto_c5", ["--log2-keys", "18"]), ("grotto_evalall", ["--log2-domain", "20"])])
def test_sharded_workloads_with_8_ranks_on_one_gpu(workload, size, tmp_path):
    """BASELINE configs[3] / configs[4] as the driver's 8-GPU box would run them -- `bench.py --gpus 8 --workload ...`,
    the domain cut by shard.domain_block, the key batch by shard.batch_slice -- with the eight ranks sharing this box's
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
