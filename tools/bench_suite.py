#!/usr/bin/env python3
"""Every other kernel of the library on one GPU (kernel-only rates, HIP events): `python bench.py --suite` runs this after
its own line and writes gpurun_out/bench_suite.json.  Not part of the driver contract."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
from bench import (CLK_HZ, LDS_PEAK_GBS, LOOKUP_BYTES_PER_AES, N_CU, VALU_ISSUE_PEAK_GINSTR, aes_blocks_per_eval,  # noqa: E402
                   hbm_bytes_per_eval, make_key, time_steady)

CHACHA_VALU_PER_WAVE_BLOCK = 1003           # measured wave-instructions per 64 ChaCha20 blocks (960 in the rounds)
# Issue floor of the add-rotate-xor mix on gfx950: the rotates are slow-class instructions (4 clk) and drag the adds and
# xors around them along; the lowest cost per instruction this mix has been measured at, in the
# stand-alone permutation loop (tools/ubench/chacha_rate.hip) and in the best walk alike, is 3.75 clk at the nominal clock.
CHACHA_CLK_PER_INSTR = 3.75


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
                                   "clk_per_instruction_and_simd": N_CU * 4 * CLK_HZ / wave_instr,
                                   # against the issue floor of THIS instruction mix (one instruction per CHACHA_CLK_PER_INSTR clocks and SIMD)
                                   "frac_of_issue_floor": wave_instr / (N_CU * 4 * CLK_HZ / CHACHA_CLK_PER_INSTR)}
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
    for scheme, n in (("dpf", 24), ("dpf", 26), ("ht", 24), ("dcf", 24), ("dcf", 26)):
        sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, G, 7 + n)
        sd, cd = s0s[0].cuda(), cws.cuda()
        od = None if ocw is None else ocw.cuda()
        ys = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
        record(f"{scheme} EvalAll n={n}", 1 << n, lambda: eng.eval_all(sid, n, G, 0, sd, cd, ocw=od, out=ys),
               {"dpf": 2.0, "ht": 1.5, "dcf": 4.0}[scheme], 16, steps=5)  # AES blocks per leaf (DCF: seed and value block of both children)
        del ys
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_suite.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    import argparse
    run_suite(argparse.Namespace())
