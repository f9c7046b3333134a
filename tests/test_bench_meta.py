"""bench.py's bookkeeping that needs no GPU: the committed rocprofv3 PMC summary is found for the headline kernel
(so `roofline.traffic` is not silently null after a kernel is renamed) and agrees with the algorithmic bytes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_headline_pmc_summary_is_found():
    import bench
    t = bench.measured_hbm_traffic(24)
    assert t is not None
    assert abs(t / ((1 << 24) * 20) - 1.0) < 0.02      # 4 B x + 16 B y per evaluation, no wasted re-reads
    p = bench.measured_pipe_busy(24)
    assert p is not None and 0.5 < p["lds_busy"] <= 1.0 and 0.5 < p["valu_busy"] <= 1.0
    assert bench.measured_hbm_traffic(20) is None       # only the profiled batch has counters


def test_algorithmic_work_tables():
    import bench
    assert bench.aes_blocks_per_eval("dpf", 32) == 32 and bench.aes_blocks_per_eval("dcf", 32) == 64
    assert bench.hbm_bytes_per_eval("dpf", 32, True) == 20


def test_workload_pmc_summaries_are_found():
    """Round 3: every sub-record of the default line that claims HBM traffic finds its committed rocprofv3 counters
    (profiles/r<N>/pmc_workloads.json) for the launch size it runs, and the traffic is the algorithmic bytes give or
    take the frontier nodes of the full-domain evaluation."""
    import bench
    alg = {"dpf_2e20": (1 << 20) * 20, "dcf_2e20": (1 << 20) * 20, "evalall_c4": (1 << 28) * 16, "ht_c5": (1 << 24) * 1084,
           "grotto_c5": (1 << 24) * 1057}
    units = {"dpf_2e20": 1 << 20, "dcf_2e20": 1 << 20, "evalall_c4": 1 << 28, "ht_c5": 1 << 24, "grotto_c5": 1 << 24}
    for name, bytes_ in alg.items():
        r = bench.replayed_counters(name, units[name])
        assert r["traffic"] is not None and r["traffic_replayed_from"], name
        assert 0.98 < r["traffic"] / bytes_ < (1.25 if name == "evalall_c4" else 1.05), (name, r["traffic"] / bytes_)
        assert 0.5 < r["lds_busy_pmc"] <= 1.0
        assert bench.replayed_counters(name, units[name] // 2)["traffic"] is None   # only the profiled launch size


def test_committed_legend_is_what_bench_generates():
    """bench_legend.json (where the line's prose lives) is generated from bench.py's own tables."""
    import json
    import bench
    have = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_legend.json")))
    assert have == json.loads(json.dumps(bench.legend()))
    cells = [c[0] for c in bench.POINTS_CELLS + bench.KEYS_CELLS] + list(bench.SHARDED)
    for name in cells + ["headline_strong", "shared_prefix", "shared_dcf", "shared_dpf_2e20", "shared_dcf_2e20", "dpf_2e20", "dcf_2e20",
                         "e2e_pcie", "c1_cpu"]:
        assert name in have["records"], name
    assert {s for _, s in bench.SHORT} | {"u"} == set(have["sub_record_keys"])


def test_a_full_line_fits_the_drivers_record():
    """The driver keeps about 8 KB of the line's tail: a line with every sub-record at full width (5 significant digits,
    every optional key present) must stay under 7 000 bytes, and every number of it must be in it."""
    import argparse
    import json
    import bench
    walk = {"value": 3.5912e9, "unit": "evals/s", "ms_per_step": 4.6597, "kernel_ms": 4.6561, "units_per_launch": 1 << 24,
            "aes_per_unit": 32, "frac": 0.93831, "lds_gbs": 73792.1, "hbm_gbs": 72.012, "hbm_frac": 0.0090015,
            "kernel": "eval_points_kernel<Lds,0,true,2,1024,false>", "grid": 256, "block": 1024, "traffic_ratio": 1.0041,
            "lds_busy_pmc": 0.95123, "replay_ok": True, "ok": True, "cpu": 2.0712e7, "cpu1": 1.3512e6, "clk": 2.3712, "frac_at_clock": 0.94971}
    subs = {c[0]: dict(walk) for c in bench.POINTS_CELLS + bench.KEYS_CELLS}
    for name in ("dpf_2e20", "dcf_2e20", "ht_c5"):
        subs[name] = dict(walk)
    subs["evalall_c4"] = dict(walk, unit="leaves/s")
    subs["grotto_c5"] = dict(walk, frac_executed=0.84123)
    shared = {"value": 1.0812e10, "unit": "evals/s", "ms_per_step": 1.5521, "kernel_ms": 1.5512, "units_per_launch": 1 << 24, "aes_per_unit": 10.0,
              "k": 23, "frac": 0.88123, "speedup": 2.9712, "kernel": "eval_points_kernel<Lds,0,true,2,1024,true>", "same": True, "clk": 2.3512,
              "frac_at_clock": 0.89951}
    for name in ("shared_prefix", "shared_dcf", "shared_dpf_2e20", "shared_dcf_2e20"):
        subs[name] = dict(shared)
    subs["e2e_pcie"] = {"value": 2.7e9, "unit": "evals/s", "serial": 1.4475e9, "pipe": 2.7123e9, "gbs": 54.246, "same": True}
    subs["c1_cpu"] = {"value": 2.1e7, "unit": "leaves/s", "us": 12.345, "cpu": 2.7e7, "cpu1": 2.75e7, "cpu_us": 9.4123, "cpu1_us": 9.3123,
                      "kat": True, "ok": True}
    cpu = {"value": 2.0712e7, "unit": "evals/s", "cores": 16, "kind": "port",
           "sample": "first 16777216 points of the timed batch x 5 passes, 4.1 s; t1: 1048576 points", "t1": 1.3512e6, "ok": True,
           "aesni": 2.9512e7}
    args = argparse.Namespace(log2_batch=24, steps=20, warmup=5)
    line, detail = bench.compose_line(args, 1, dict(walk), dict(walk, scaling="strong"), subs, cpu, "nccl")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 7000, len(text)
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["config"]["workload"] and back["roofline"]["bound"] == "lds" and back["cpu_baseline"]["sample"]
    assert set(back["sub"]) == set(subs) and back["gates"]["all"] is True
    assert back["sub"]["grotto_c5"]["fx"] == 0.84123 and back["sub"]["evalall_c4"]["u"] == "leaves/s"
    assert back["sub"]["shared_prefix"]["f"] == 0.88123 and back["sub"]["shared_prefix"]["k"] == 23 and back["sub"]["shared_prefix"]["x"] == 2.9712
    assert back["sub"]["dpf_keys"]["clk"] == 2.3712 and back["sub"]["dpf_keys"]["fc"] == 0.94971 and back["roofline"]["clk"] == 2.3712
    assert back["headline_strong"]["scaling"] == "strong" and back["headline_strong"]["v"] == back["value"]


def test_an_eight_rank_line_is_the_metrics_one_batch_and_names_its_devices():
    """With N > 1 ranks `value` is headline_strong (the ONE 2^24-point batch cut N ways), the line says "strong", the
    per-rank-batch number moves to `weak`, the roofs describe the strong launch (no PMC replay: another launch size),
    config.ranks names eight devices -- and the line still fits the driver's record."""
    import argparse
    import json
    import bench
    walk = {"value": 2.8712e10, "unit": "evals/s", "ms_per_step": 4.6597, "kernel_ms": 4.6561, "units_per_launch": 1 << 24,
            "aes_per_unit": 32, "frac": 0.93831, "lds_gbs": 73792.1, "hbm_gbs": 72.012, "hbm_frac": 0.0090015,
            "kernel": "eval_points_kernel<Lds,0,true,2,1024,false>", "grid": 256, "block": 1024, "clk": 2.3712, "frac_at_clock": 0.94971,
            "ms_median": 4.6512}
    strong = dict(walk, value=2.7512e10, ms_per_step=0.60981, kernel_ms=0.60012, units_per_launch=1 << 21, frac=0.90123, scaling="strong",
                  ms_median=0.59871)
    subs = {c[0]: dict(walk) for c in bench.POINTS_CELLS + bench.KEYS_CELLS}
    for name in ("ht_c5", "grotto_c5"):
        subs[name] = dict(walk, ms_median=1.2345)
    subs["evalall_c4"] = dict(walk, unit="leaves/s")
    for name in ("shared_prefix", "shared_dcf"):
        subs[name] = dict(walk, k=23, speedup=2.9712, same=True)
    ranks = [{"rank": r, "pid": 100000 + r, "dev": r, "name": "AMD Instinct MI355X", "uuid": "GPU-%032x" % (0xabcdef0123456789 * (r + 1)),
              "pci": "0000:%02x:00" % (5 + 16 * r), "pg": ["nccl", 8]} for r in range(8)]
    args = argparse.Namespace(log2_batch=24, steps=20, warmup=5)
    line, _ = bench.compose_line(args, 8, dict(walk), strong, subs, None, "nccl", ranks)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 7000, len(text)
    back = json.loads(text)
    assert back["n_gpus"] == 8 and back["scaling"] == "strong"
    assert back["value"] == back["headline_strong"]["v"] == 2.7512e10 and back["ms_per_step"] == back["headline_strong"]["ms"]
    assert back["ms_median"] == 0.59871 and back["roofline"]["kernel_ms"] == 0.60012 and back["roofline"]["frac"] == 0.90123
    assert back["roofline"]["traffic"] is None and back["roofline"]["replay_ok"] is False
    assert back["roofline_hbm"]["algorithmic_bytes"] == (1 << 21) * 20
    assert back["weak"]["v"] == 2.8712e10 and back["weak"]["scaling"] == "weak" and back["weak"]["n"] == 1 << 24
    assert len(back["config"]["ranks"]) == 8 and back["config"]["distinct_devices"] == 8 and back["config"]["device_name"] == "AMD Instinct MI355X"
    assert all(r["pg"] == ["nccl", 8] for r in back["config"]["ranks"]) and "name" not in back["config"]["ranks"][0]
    assert bench.distinct_devices([dict(r, uuid=ranks[0]["uuid"]) for r in ranks]) == 1   # eight ranks sharing one GPU
    # one rank: the same launch under both names, "weak", no side record
    one, _ = bench.compose_line(args, 1, dict(walk), dict(walk, scaling="strong"), subs, None, None, ranks[:1])
    assert one["scaling"] == "weak" and "weak" not in one and one["value"] == one["headline_strong"]["v"] and one["ms_median"] == 4.6512
