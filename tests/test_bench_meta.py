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
