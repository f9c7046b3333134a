"""Sizes past 2^31 on one MI355X: the index arithmetic of the walks is 64-bit end to end.

The reference's launchers index with int (include/fss/eval_all_gpu.cuh:257,287 overflow at in_bits >= 31;
point_eval_gpu.cuh takes `int n`); 288 GB of HBM hold batches the reference cannot address, so the sizes
where a 32-bit index would wrap are tested here: a points batch of 2^31 + 4099 evaluations, a whole EvalAll
domain of 2^32 leaves (64 GiB of outputs) and Grotto's one-byte EvalAll at 2^33 leaves.  Each is checked
against launches that stay below 2^31 (themselves oracle-checked in test_gpu_parity.py), against the oracle
on samples that straddle 2^31 and 2^32, and through the FSS property on every output.
"""
import numpy as np
import pytest

import groups as G
from conftest import SplitMix64

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

DPF, DCF, HT, GROTTO = 0, 1, 2, 3
GiB = 1 << 30


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _need_free(nbytes):
    free, _ = torch.cuda.mem_get_info()
    if free < nbytes:
        pytest.skip(f"needs {nbytes / GiB:.0f} GiB of free HBM, {free / GiB:.0f} GiB available")


def _u64(t):
    return (t[:, 0].to(torch.int64) & 0xFFFFFFFF) | (t[:, 1].to(torch.int64) << 32)


def _release():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("scheme", [DPF, DCF])
def test_points_batch_past_2e31(engine, engine_plain, O, scheme):
    n, g = 32, G.U64
    npts = (1 << 31) + 4099
    _need_free(npts * 20 + 4 * GiB)
    rng = SplitMix64(0x5EED3100 + scheme)
    p = O.Prg(2 if scheme == DPF else 4)
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), rng.i4()
    cws = O.dpf_gen(p, n, g, s, alpha, beta) if scheme == DPF else O.dcf_gen(p, n, g, 0, s, alpha, beta)
    cd = T(cws).cuda()
    gen = torch.Generator(device="cuda").manual_seed(0x3100 + scheme)
    xs = torch.empty((npts,), dtype=torch.int32, device="cuda")
    for lo in range(0, npts, 1 << 28):
        hi = min(npts, lo + (1 << 28))
        xs[lo:hi] = torch.randint(-2**31, 2**31, (hi - lo,), dtype=torch.int32, device="cuda", generator=gen)
    a32 = alpha - (1 << 32) if alpha >= (1 << 31) else alpha
    marks = [0, (1 << 31) - 1, 1 << 31, (1 << 31) + 1, npts - 1]
    for m in marks:
        xs[m] = a32
    ys = torch.full((npts, 4), -1, dtype=torch.int32, device="cuda")
    party = 1
    engine_plain.eval_points(scheme, n, g, party, T(s[party]).cuda(), cd, xs, out=ys)   # every point from the root
    assert engine_plain.last_kernel()[0].endswith("_eval_points")
    # slices of the batch through launches of their own, below 2^31 each (FSS_AMD_POINTS_AUTO: the shared-prefix plan)
    for lo, cnt in ((0, 1 << 20), ((1 << 31) - (1 << 19), 1 << 20), (npts - (1 << 20) - 3, (1 << 20) + 3)):
        part = engine.eval_points(scheme, n, g, party, T(s[party]).cuda(), cd, xs[lo:lo + cnt].clone())
        assert torch.equal(part, ys[lo:lo + cnt]), (lo, cnt)
    # no output was skipped: the upper words of a Uint<uint64_t> share are zero, the fill was -1
    for lo in range(0, npts, 1 << 28):
        assert bool((ys[lo:lo + (1 << 28), 2:] == 0).all()), lo
    # the oracle on a strided sample + the marks
    idx = np.unique(np.concatenate([np.arange(0, npts, npts // 4096 + 1), np.array(marks)])).astype(np.int64)
    it = torch.from_numpy(idx).cuda()
    xs_s = xs[it].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    want = O.eval_points(p, scheme, n, g, party, s[party], cws, xs_s.astype(np.uint64), nthreads=4)
    assert (ys[it].cpu().numpy() == want).all()
    del ys, xs
    _release()


def test_dpf_eval_all_2e32_leaves(engine, O):
    from fss_amd import shard
    n, g = 32, G.U64
    N = 1 << n
    _need_free(16 * N + 16 * (N >> 4) + 8 * GiB)
    rng = SplitMix64(0x5EED3200)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    alpha, beta = (1 << 31) + 12345, rng.i4()  # the hit lies past 2^31
    cws = O.dpf_gen(p, n, g, s, alpha, beta)
    cd = T(cws).cuda()
    whole = torch.full((N, 4), -1, dtype=torch.int32, device="cuda")
    engine.eval_all(DPF, n, g, 0, T(s[0]).cuda(), cd, out=whole)
    # party 1 comes in the sixteen 2^28-leaf blocks of a 16-way split; every leaf reconstructs
    world = 16
    blk = torch.empty((N // world, 4), dtype=torch.int32, device="cuda")
    bv = (int(beta[0]) & 0xFFFFFFFF) | ((int(beta[1]) & 0xFFFFFFFF) << 32)
    nonzero = 0
    for r in range(world):
        first, cnt = shard.domain_block(n, r, world)
        blk.fill_(-1)
        engine.eval_all(DPF, n, g, 1, T(s[1]).cuda(), cd, first_leaf=first, nleaves=cnt, out=blk)
        for lo in range(0, cnt, 1 << 26):
            tot = _u64(whole[first + lo:first + lo + (1 << 26)]) + _u64(blk[lo:lo + (1 << 26)])
            nz = torch.nonzero(tot).flatten()
            nonzero += nz.numel()
            if first + lo <= alpha < first + lo + (1 << 26):
                assert nz.numel() == 1 and int(nz[0]) + first + lo == alpha
                assert (int(tot[alpha - first - lo]) & 0xFFFFFFFFFFFFFFFF) == bv
            assert bool((whole[first + lo:first + lo + (1 << 26), 2:] == 0).all())
            del tot, nz
    assert nonzero == 1
    # party 0's blocks past 2^31 from launches of their own = the same rows of the whole domain
    for r in (7, 8, 15):
        first, cnt = shard.domain_block(n, r, world)
        blk.fill_(-1)
        engine.eval_all(DPF, n, g, 0, T(s[0]).cuda(), cd, first_leaf=first, nleaves=cnt, out=blk)
        assert torch.equal(blk, whole[first:first + cnt]), r
    # the oracle's Dpf::Eval on leaves either side of 2^31, at both ends and around alpha
    idx = np.unique(np.concatenate([
        np.arange(0, N, N // 2048 + 1), np.arange((1 << 31) - 64, (1 << 31) + 64),
        np.array([0, N - 1, alpha - 1, alpha, alpha + 1])])).astype(np.int64)
    it = torch.from_numpy(idx).cuda()
    want = O.eval_points(p, DPF, n, g, 0, s[0], cws, idx.astype(np.uint64), nthreads=4)
    assert (whole[it].cpu().numpy() == want).all()
    del whole, blk
    _release()


def test_grotto_eval_all_2e33_leaves(engine, O):
    n = 33
    N = 1 << n
    _need_free(2 * N + 6 * GiB)
    rng = SplitMix64(0x5EED3300)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    alpha = (1 << 32) + 98765
    cws = O.grotto_gen(p, n, s, alpha)
    cd = T(cws).cuda()
    y = [engine.eval_all(GROTTO, n, G.BYTES, b, T(s[b]).cuda(), cd) for b in (0, 1)]
    assert y[0].numel() == N
    # every leaf reconstructs: y0 ^ y1 = 1{x >= alpha} (grotto_dcf.cuh:137-170)
    for lo in range(0, N, 1 << 28):
        hi = lo + (1 << 28)
        rec = y[0][lo:hi] ^ y[1][lo:hi]
        if hi <= alpha:
            assert not bool(rec.any()), lo
        elif lo >= alpha:
            assert bool((rec == 1).all()), lo
        else:
            assert not bool(rec[:alpha - lo].any()) and bool((rec[alpha - lo:] == 1).all())
        del rec
    # blocks past 2^32 from launches of their own, carry applied as fss_amd.shard does it
    from fss_amd import shard
    world = 8
    lasts = []
    blocks = {}
    for r in range(world):
        first, cnt = shard.domain_block(n, r, world)
        b = engine.eval_all(GROTTO, n, G.BYTES, 0, T(s[0]).cuda(), cd, first_leaf=first, nleaves=cnt)
        lasts.append(int(b[-1]))
        if r in (3, 4, 7):
            blocks[r] = b
        else:
            del b
    carries = shard.grotto_carries(lasts)
    for r, b in blocks.items():
        first, cnt = shard.domain_block(n, r, world)
        assert torch.equal(b ^ carries[r], y[0][first:first + cnt]), r
    del y, blocks
    _release()


@pytest.mark.parametrize("scheme", [DPF, HT])
def test_eval_all_batch_past_2e32_leaves(engine, O, scheme):
    """fss::gpu::*EvalAllGpuBatch with nkeys * 2^in_bits > 2^32: ys[k * 2^n + x], 64-bit in k * 2^n."""
    n, g = 16, G.U64
    nkeys = (1 << 16) + 3
    _need_free(16 * (nkeys << n) + 8 * GiB)
    gen = torch.Generator(device="cuda").manual_seed(0x3400 + scheme)
    s0s = torch.randint(-2**31, 2**31, (nkeys, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    alphas = torch.randint(0, 1 << n, (nkeys,), dtype=torch.int32, device="cuda", generator=gen)
    betas = torch.randint(-2**31, 2**31, (nkeys, 4), dtype=torch.int32, device="cuda", generator=gen)
    betas[:, 2:] = 0
    cws, ocws = engine.gen(scheme, n, g, s0s, alphas, betas)
    party = 0
    seeds = s0s[:, party].contiguous()
    ys = torch.full((nkeys, 1 << n, 4), -1, dtype=torch.int32, device="cuda")
    engine.eval_all_batch(scheme, n, g, party, seeds, cws, ocws=ocws, out=ys)
    # keys either side of leaf 2^31 and 2^32 and at both ends: the single-key EvalAll of the same key
    for k in (0, 1, (1 << 15) - 1, 1 << 15, (1 << 16) - 1, 1 << 16, nkeys - 1):
        one = engine.eval_all(scheme, n, g, party, seeds[k], cws[k], ocw=None if ocws is None else ocws[k])
        assert torch.equal(one, ys[k]), k
    # ... and the oracle's EvalAll for three of them
    p = O.Prg(1 if scheme == HT else 2)
    for k in (0, 1 << 15, nkeys - 1):
        c = cws[k].cpu().numpy()
        if scheme == HT:
            want = O.ht_eval_all(p, n, g, party, seeds[k].cpu().numpy(), c, ocws[k].cpu().numpy())
        else:
            want = O.dpf_eval_all(p, n, g, party, seeds[k].cpu().numpy(), c)
        assert (ys[k].cpu().numpy() == want).all(), k
    # every output written (upper words of a Uint<uint64_t> share are zero, the fill was -1)
    for lo in range(0, nkeys, 1 << 12):
        assert bool((ys[lo:lo + (1 << 12), :, 2:] == 0).all()), lo
    # party 1 on the last 2^12 keys: every leaf reconstructs
    lo = nkeys - (1 << 12)
    y1 = engine.eval_all_batch(scheme, n, g, 1, s0s[lo:, 1].contiguous(), cws[lo:].contiguous(),
                               ocws=None if ocws is None else ocws[lo:].contiguous())
    tot = _u64(ys[lo:].reshape(-1, 4)) + _u64(y1.reshape(-1, 4))
    want = torch.zeros((1 << 12, 1 << n), dtype=torch.int64, device="cuda")
    bl = (betas[lo:, 0].to(torch.int64) & 0xFFFFFFFF) | (betas[lo:, 1].to(torch.int64) << 32)
    want[torch.arange(1 << 12, device="cuda"), alphas[lo:].to(torch.int64)] = bl
    assert torch.equal(tot.view(1 << 12, 1 << n), want)
    del ys, y1, tot, want
    _release()
