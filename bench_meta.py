"""bench.py's tables: the MI355X constants its roofs are computed from, the legend of the one JSON line (short keys,
records; bench_legend.json is generated from them), the algorithmic work per unit, the replay of the
committed rocprofv3 counters, and the synthetic inputs (keys from the product's device Gen, SplitMix-free torch
generators seeded per workload).  Nothing here is timed."""
import json
import math
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))

# MI355X constants (/opt/skills/guides/MI355X_MICROARCH.md): 256 CUs, 2.4 GHz max clock,
# ds_read_b32 = 128 B/clk/CU, HBM3E 8 TB/s.
N_CU, CLK_HZ = 256, 2.4e9
LDS_PEAK_GBS = 128 * N_CU * CLK_HZ / 1e9   # 78 643 GB/s of 4-byte table lookups
HBM_PEAK_GBS = 8000.0
LOOKUP_BYTES_PER_AES = 160 * 4              # 144 Te0 + 16 S-box lookups, 4 B each
VALU_ISSUE_PEAK_GINSTR = N_CU * 4 * CLK_HZ / 2.0 / 1e9   # 1 228.8 G wave64-instr/s: 32 lanes per SIMD and clock
METRIC = "DPF/DCF evals/sec at kInBits=32, batch=2^24"

# ---------------------------------------------------------------------------------------------------------- legend
# Short keys of the sub-records (and of `headline_strong`); bench_legend.json is generated from these two tables.
KEYS = {
    "v": "whole-job rate over all GPUs: units x steps / wall time of the timed region (max over ranks); evals/s unless `u` says otherwise",
    "u": "unit of `v` when it is not evals/s",
    "ms": "wall milliseconds per step (barrier + synchronize either side, max over ranks)",
    "kms": "rank 0's kernel milliseconds per launch, HIP events on the launch stream around the timed steps",
    "med": "rank 0's MEDIAN launch of the K timed steps, ms (an event between the steps; SURVEY 8(d) asks for the median, `kms` is the mean)",
    "n": "units (evaluations, keys or leaves) one launch of rank 0 processes",
    "a": "algorithmic AES blocks per unit (taken child only; SURVEY 8d)",
    "f": "LDS roofline fraction: n x a x 160 lookups x 4 B / kms / 78 643.2 GB/s (128 B/clk/CU x 256 CU x 2.4 GHz)",
    "clk": "shader clock in GHz measured INSIDE this record's kernel: ~50 ms of its launches through the diagnostic build of the same "
           "sources (libfss_amd_diag.so), median over the waves of the last launch of d(s_memtime) / d(s_memrealtime) x 100 MHz; "
           "untimed, after the record's timed region",
    "fc": "the fraction at THAT clock: (fx if present, else f) x 2.4 / clk -- lookups per LDS cycle achieved against the pipe's "
          "128 B/clk/CU, independent of the clock the box's power cap grants",
    "k": "shared_* records: prefix_bits (levels of the key's tree expanded once per call)",
    "x": "shared_* records: speed-up over the plain walk of the same batch (kms of the plain walk / kms)",
    "fx": "the same fraction counting only the lookups the kernel executes (Grotto: the 32 sibling blocks per walk are "
          "control-bit-only, 133 of 160 lookups)",
    "h": "algorithmic HBM bytes per second / 8 000 GB/s",
    "tr": "HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate passes) / "
          "algorithmic bytes per launch; replayed from profiles/, only for the launch size profiled",
    "lb": "SQ_LDS_IDX_ACTIVE / CU / kernel cycles of the committed PMC pass (LDS busy fraction)",
    "rp": "true when the replayed counters were captured for exactly the sources this library's kernel is compiled from",
    "ok": "parity gate: the GPU outputs of the timed launch equal the oracle's on the sample named in the legend",
    "cpu": "the oracle (OpenSSL EVP PRG in the reference's call pattern) on all usable host cores, units/s",
    "cpu1": "the same on one thread",
    "serial": "e2e_pcie: evals/s with H2D of x, the walk and D2H of y one after the other on one stream",
    "pipe": "e2e_pcie: evals/s of the host-buffer entry point fss_amd_eval_points_host: ONE launch whose walk reads x from and "
            "writes y to the pinned host buffers across PCIe (no staging copies)",
    "gbs": "e2e_pcie: bytes over PCIe per second of `pipe` (4 B up + 16 B down per evaluation), GB/s",
    "same": "outputs bit-identical to the resident-input run of the same batch",
    "us": "c1_cpu: GPU microseconds per 256-leaf EvalAll",
    "cpu_us": "c1_cpu: oracle microseconds per EvalAll on all threads",
    "cpu1_us": "c1_cpu: oracle microseconds per EvalAll on one thread",
    "kat": "c1_cpu: the oracle's EvalAll reproduces the reference's recorded SHA-256 for both parties (SURVEY 8c)",
}
PTS = "party 0, Uint<uint64_t>, 2^B uniformly random points per GPU per step (B = --log2-batch, 24)"
RECORDS = {
    "value": f"DPF (Boyle'16) kInBits=32, 1 key, {PTS}; configs[1]'s shape at the metric's batch. ok: the first --cpu-sample outputs. "
             "With N > 1 ranks `value` is headline_strong (the metric's ONE batch cut N ways, \"scaling\": \"strong\") and the per-rank-batch "
             "number is the side record `weak`",
    "weak": "N > 1 only: every rank its own 2^B-point batch of the headline key (weak scaling), whole-job rate",
    "headline_strong": "the headline's ONE 2^B-point batch cut by shard.batch_slice: 2^B / N points per rank, same key, max-over-ranks time "
              "(SURVEY 8e); with one GPU it is the headline itself",
    "shared_prefix": "the headline batch through the shared-prefix plan (fss_amd_dpf_eval_points_shared with the library's own prefix_bits k = "
                     "what fss_amd_dpf_eval_points picks by itself under FSS_AMD_POINTS_AUTO; every other record runs FSS_AMD_POINTS_PLAIN): "
                     "the key's top k levels are expanded once per step, a = (32 - k) + 2^(k+1) / n AES blocks EXECUTED per evaluation and f "
                     "counts those; x = speed-up over the headline (work removed, never `value`). same: outputs identical to the plain walk",
    "shared_dcf": "the dcf_points batch through the shared-prefix plan (fss_amd_dcf_eval_points_shared), as shared_prefix; a = 2 x ((32 - k) + 2^(k+1) / n)",
    "shared_dpf_2e20": "the dpf_2e20 batch (configs[1]) through the shared-prefix plan, timed like dpf_2e20",
    "shared_dcf_2e20": "the dcf_2e20 batch (configs[2]) through the shared-prefix plan, timed like dcf_2e20",
    "dcf_points": f"DCF (Boyle'21, mul=4 PRG, kLt) kInBits=32, 1 key, {PTS}. ok: 4096 strided outputs",
    "dpf_n16": f"DPF kInBits=16, 1 key, {PTS}. ok: 4096 strided outputs",
    "dpf_n64": f"DPF kInBits=64, 1 key, {PTS}. ok: 4096 strided outputs",
    "dcf_n16": f"DCF kInBits=16, 1 key, {PTS}. ok: 4096 strided outputs",
    "dcf_n64": f"DCF kInBits=64, 1 key, {PTS}. ok: 4096 strided outputs",
    "dpf_keys": "DPF kInBits=32, 2^B independent level-major keys x 1 point each per GPU per step (SURVEY 8d M(ii); "
                "src/bench_gpu.cu:604-633), keys generated on the device outside the timed region. ok: 4096 strided keys brought back key-major",
    "dcf_keys": "DCF kInBits=32, 2^B independent level-major keys x 1 point (src/bench_gpu.cu:637-661). ok: 4096 strided keys",
    "dpf_keys_n16": "DPF kInBits=16, 2^B independent level-major keys x 1 point. ok: 4096 strided keys",
    "dpf_keys_n64": "DPF kInBits=64, 2^B independent level-major keys x 1 point. ok: 4096 strided keys",
    "dpf_2e20": "BASELINE configs[1] at its literal size: DPF kInBits=32, 1 key, 2^20 points, 1 GPU; >= 150 ms of back-to-back "
                "launches after 60 ms of warm-up. ok / cpu: all 2^20 points",
    "dcf_2e20": "BASELINE configs[2] at its literal size: DCF kInBits=32, 1 key, 2^20 points, 1 GPU; timed like dpf_2e20",
    "evalall_c4": "BASELINE configs[3]: DPF full-domain EvalAll kInBits=28 (--log2-domain), the domain cut into N aligned blocks "
                  "(shard.domain_block), leaves/s, strong scaling. ok: a kInBits=22 domain leaf for leaf + 4097 strided leaves of "
                  "the timed outputs. cpu: the oracle in the reference's task-recursion shape on 2^22 leaves",
    "ht_c5": "BASELINE configs[4]: Half-Tree DPF kInBits=64, 2^24 (--log2-keys) independent keys x 1 point, key batch cut by "
             "shard.batch_slice, strong scaling. ok / cpu: 2^18 strided keys",
    "grotto_c5": "BASELINE configs[4]: Grotto DCF kInBits=64 path walk, 2^24 independent keys x 1 point, 1 B per output; a = 96 "
                 "(64 taken children + 32 siblings on average); fx counts the siblings at the 133 lookups they execute. ok / cpu: 2^18 strided keys",
    "e2e_pcie": "the headline batch with x read from and y written to pinned host memory (4 B up + 16 B down per evaluation); "
                "the PCIe-inclusive rate SURVEY 8(d) asks to state separately, never `value`",
    "c1_cpu": "BASELINE configs[0]: DPF kInBits=8, 1 key, 256-leaf EvalAll, Bytes (samples/dpf_dcf_cpu.cu constants); the CPU-runnable "
              "case: v = GPU leaves/s; cpu = the oracle's better shape (all threads in the reference's task recursion -- cpu_us -- or one "
              "thread sequentially -- cpu1_us; a 16-thread team costs more than a 256-leaf tree), EVP contexts built once outside the repetitions",
}
TOP = {
    "roofline": "headline kernel against the LDS roof: achieved = 2^B x 32 AES x 640 B / kernel_ms; clk / frac_at_clock as the sub-records' "
                "clk / fc; traffic = PMC HBM bytes per launch "
                "(replayed from `replay`, valid when replay_ok); lds_busy / valu_busy = SQ_LDS_IDX_ACTIVE and SQ_ACTIVE_INST_VALU x 4 "
                "per SIMD over the kernel's cycles in that PMC pass",
    "roofline_hbm": "algorithmic 20 B per evaluation / kernel_ms against 8 TB/s",
    "roofline_valu": "wave64 VALU instructions per launch (replayed SQ_INSTS_VALU) / kernel_ms against 256 CU x 4 SIMD x 2.4 GHz / 2 clk",
    "cpu_baseline": "the oracle with the OpenSSL EVP PRG on the first --cpu-sample points of the timed batch: value = all usable cores, "
                    "t1 = one thread, aesni = the AES-NI variant (the reference's Aes128MmoRaw), ok = GPU outputs equal",
    "gates": "every `ok` / `same` / `kat` of the line and'ed, and how many there were",
    "ms_median": "median of the K timed launches of `value`'s kernel on rank 0, ms, from an event between the steps (SURVEY 8(d); "
                 "src/bench_gpu.cu:48-68); ms_per_step stays the wall mean the rate is computed from",
    "config.ranks": "one entry per rank, all-gathered: rank, local device index, device name, UUID and PCI address as the runtime reports "
                    "them, visible-devices variable if set, pg = [process-group backend, its size as that rank sees it]; "
                    "config.distinct_devices = how many different GPUs the entries name",
}


def legend():
    return {"line": "bench.py's one JSON line; numbers rounded to 5 significant digits", "top_level": TOP, "sub_record_keys": KEYS,
            "records": RECORDS}


def r5(x):
    """5 significant digits (what the line prints; the detail file keeps full precision)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x == 0 or not math.isfinite(x):
            return x
        return float(f"{x:.5g}")
    if isinstance(x, dict):
        return {k: r5(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [r5(v) for v in x]
    return x


SHORT = (("value", "v"), ("ms_per_step", "ms"), ("kernel_ms", "kms"), ("ms_median", "med"), ("units_per_launch", "n"), ("aes_per_unit", "a"), ("k", "k"),
         ("frac", "f"), ("frac_executed", "fx"), ("clk", "clk"), ("frac_at_clock", "fc"), ("speedup", "x"), ("hbm_frac", "h"), ("traffic_ratio", "tr"), ("lds_busy_pmc", "lb"), ("replay_ok", "rp"), ("ok", "ok"),
         ("cpu", "cpu"), ("cpu1", "cpu1"), ("serial", "serial"), ("pipe", "pipe"), ("gbs", "gbs"), ("same", "same"), ("us", "us"),
         ("cpu_us", "cpu_us"), ("cpu1_us", "cpu1_us"), ("kat", "kat"))


def compact(rec):
    """A verbose record -> the short form of the line (KEYS)."""
    out = {}
    for long, short in SHORT:
        if rec.get(long) is not None:
            out[short] = r5(rec[long])
    if rec.get("unit") not in (None, "evals/s"):
        out["u"] = rec["unit"]
    return out


# ---------------------------------------------------------------------------------------------------------- work tables
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
    doubled as MI355X_MICROARCH.md prescribes for gfx950.  Replayed from profiles/ (see `roofline.replay`)."""
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


def _workload_pmc(name):
    """Counters of one workload's dominant kernel REPLAYED from profiles/r<N>/pmc_workloads.json (highest N first; written
    by tools/profile_workloads.sh from separate rocprofv3 --pmc passes of `bench.py --workload <name>`)."""
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


# ---------------------------------------------------------------------------------------------------------- inputs
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


def _rand_xs(n, batch, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    if n <= 32:
        xs = torch.randint(-2**31, 2**31, (batch,), dtype=torch.int32, device="cuda", generator=gen)
        return xs & ((1 << n) - 1) if n < 32 else xs
    return torch.randint(-2**63, 2**63 - 1, (batch,), dtype=torch.int64, device="cuda", generator=gen)


def _u64(a):
    """int32 / int64 bit patterns (numpy) -> uint64 values."""
    return (a.view(np.uint32) if a.dtype == np.int32 else a.view(np.uint64)).astype(np.uint64)
