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
