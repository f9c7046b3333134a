"""BASELINE.json configs[3] and configs[4] at their stated sizes on one MI355X, through the C ABI.

configs[3]: DPF full-domain EvalAll kInBits=28 (2^28 outputs, 4 GiB per party), whole and as the eight
2^25-leaf blocks an 8-GPU run hands out (fss_amd.shard.domain_block).  In the spirit of the reference's
check_evalall_gpu.cu:64-91 -- every output, both parties -- the blocks are compared with the whole domain
leaf for leaf on the device, the FSS property is checked over all 2^28 leaves (:94-114), and a strided
sample plus every block edge is compared with the oracle's point evaluation.

configs[4]: Half-Tree DPF and Grotto DCF kInBits=64, 2^24 independent keys x 1 point each: every key
reconstructs, a strided sample equals the oracle, and the per-rank key slices of an 8-way split
(fss_amd.shard.batch_slice, 2^21 keys each) give the rows of the whole batch.
"""
import numpy as np
import pytest

import groups as G
from conftest import SplitMix64

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

DPF, DCF, HT, GROTTO = 0, 1, 2, 3


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def u64_of(t):
    """int32[n,4] shares of Uint<uint64_t> -> int64[n] bit patterns (wrapping arithmetic = mod 2^64)."""
    return (t[:, 0].to(torch.int64) & 0xFFFFFFFF) | (t[:, 1].to(torch.int64) << 32)


def test_config3_dpf_eval_all_n28_whole_and_blocks(engine, O):
    from fss_amd import shard
    n, g, world = 28, G.U64, 8
    N = 1 << n
    rng = SplitMix64(0x5EED0004)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), rng.i4()
    cws = O.dpf_gen(p, n, g, s, alpha, beta)
    cd = T(cws).cuda()
    whole = [engine.eval_all(DPF, n, g, b, T(s[b]).cuda(), cd) for b in (0, 1)]
    assert whole[0].shape == (N, 4)

    # (1) the eight per-GPU blocks are the whole domain, leaf for leaf, both parties
    nl = N // world
    blk = torch.empty((nl, 4), dtype=torch.int32, device="cuda")
    edges = []
    for b in (0, 1):
        for r in range(world):
            first, cnt = shard.domain_block(n, r, world)
            assert (first, cnt) == (r * nl, nl)
            blk.fill_(-1)
            engine.eval_all(DPF, n, g, b, T(s[b]).cuda(), cd, first_leaf=first, nleaves=cnt, out=blk)
            assert torch.equal(blk, whole[b][first:first + cnt]), (b, r)
            edges += [first, first + cnt - 1]

    # (2) FSS property over all 2^28 leaves: exactly one non-zero reconstructed leaf, beta at alpha
    bv = (int(beta[0]) & 0xFFFFFFFF) | ((int(beta[1]) & 0xFFFFFFFF) << 32)
    nonzero = 0
    for lo in range(0, N, 1 << 25):  # chunked: bounded temporaries
        tot = u64_of(whole[0][lo:lo + (1 << 25)]) + u64_of(whole[1][lo:lo + (1 << 25)])
        nz = torch.nonzero(tot).flatten()
        nonzero += nz.numel()
        if lo <= alpha < lo + (1 << 25):
            assert nz.numel() == 1 and int(nz[0]) + lo == alpha
            assert (int(tot[alpha - lo]) & 0xFFFFFFFFFFFFFFFF) == bv
        assert bool((whole[0][lo:lo + (1 << 25), 2:] == 0).all())
    assert nonzero == 1

    # (3) 8192 strided leaves + every block edge + alpha and its neighbours equal Dpf::Eval of the oracle
    idx = np.unique(np.concatenate([np.arange(0, N, N // 8192 + 1), np.array(edges), np.array(
        [alpha, max(alpha - 1, 0), min(alpha + 1, N - 1)])])).astype(np.int64)
    assert len(idx) >= 4096
    it = torch.from_numpy(idx).cuda()
    for b in (0, 1):
        got = whole[b][it].cpu().numpy()
        want = O.eval_points(p, DPF, n, g, b, s[b], cws, idx.astype(np.uint64), nthreads=4)
        assert (got == want).all(), b


def _sub_keys(keys, idx_t):
    """The level-major arrays of a subset of keys (columns idx of every level)."""
    from fss_amd.engine import LevelMajorKeys
    pick = lambda a, per_key: None if a is None else (a[:, idx_t].contiguous() if per_key else a[idx_t].contiguous())
    tr = keys.tr_bits
    if tr is not None:
        planes = tr.numel() // keys.nkeys
        tr = tr.view(planes, keys.nkeys)[:, idx_t].contiguous().view(-1)
    return LevelMajorKeys(keys.scheme, keys.in_bits, int(idx_t.numel()), pick(keys.cw_s, True), cw_v=pick(keys.cw_v, True),
                          tr_bits=tr, out_cw=pick(keys.out_cw, False), extra=pick(keys.extra, False))


def _c5_inputs(nkeys, seed):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    s0s = torch.randint(-2**31, 2**31, (nkeys, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    alphas = torch.randint(-2**63, 2**63 - 1, (nkeys,), dtype=torch.int64, device="cuda", generator=gen)
    betas = torch.randint(-2**31, 2**31, (nkeys, 4), dtype=torch.int32, device="cuda", generator=gen)
    xs = torch.randint(-2**63, 2**63 - 1, (nkeys,), dtype=torch.int64, device="cuda", generator=gen)
    return s0s, alphas, betas, xs


def test_config4_half_tree_n64_2e24_keys(engine, O):
    from fss_amd import shard
    n, g, nkeys, world = 64, G.U64, 1 << 24, 8
    s0s, alphas, betas, xs = _c5_inputs(nkeys, 0x5EED0005)
    xs[0::2] = alphas[0::2]  # half of the points hit alpha
    keys, ocws = engine.gen_level_major(HT, n, g, s0s, alphas, betas)
    seeds = [s0s[:, b].contiguous() for b in (0, 1)]
    y = [engine.eval_keys(keys, g, b, seeds[b], xs, ocws=ocws) for b in (0, 1)]
    total = u64_of(y[0]) + u64_of(y[1])
    bl = (betas[:, 0].to(torch.int64) & 0xFFFFFFFF) | (betas[:, 1].to(torch.int64) << 32)
    want = torch.where(xs == alphas, bl, torch.zeros_like(bl))
    assert bool((total == want).all())
    assert int((xs == alphas).sum()) >= nkeys // 2
    assert bool((y[0][:, 2:] == 0).all())

    # strided sample (4099 keys) of party 0 and party 1 against the oracle, on the same keys in key-major form
    idx = torch.arange(0, nkeys, 4093, device="cuda")
    sub = _sub_keys(keys, idx)
    cws_km = engine.key_major(sub).cpu().numpy()
    ii = idx.cpu().numpy()
    p = O.Prg(1)
    for b in (0, 1):
        want_b = O.eval_keys(p, HT, n, g, b, seeds[b][idx].cpu().numpy(), cws_km, xs[idx].cpu().numpy().view(np.uint64),
                             ocws=ocws[idx].cpu().numpy(), nthreads=4)
        assert (y[b][idx].cpu().numpy() == want_b).all(), b
    assert len(ii) >= 4096

    # the 8-way key-batch split: rank r's 2^21-key slice evaluated by itself gives its rows of the whole batch
    for r in (0, 3, 7):
        lo, hi = shard.batch_slice(nkeys, r, world)
        assert hi - lo == 1 << 21
        sl = _sub_keys(keys, torch.arange(lo, hi, device="cuda"))
        got = engine.eval_keys(sl, g, 1, seeds[1][lo:hi].contiguous(), xs[lo:hi].contiguous(), ocws=ocws[lo:hi].contiguous())
        assert torch.equal(got, y[1][lo:hi]), r


def test_config4_grotto_n64_2e24_keys(engine, O):
    from fss_amd import shard
    n, nkeys, world = 64, 1 << 24, 8
    s0s, alphas, _, xs = _c5_inputs(nkeys, 0x5EED0006)
    xs[0::5] = alphas[0::5]
    xs[1::5] = alphas[1::5] - 1  # wraps at alpha = 0, like the unsigned domain
    xs[2] = -1                   # x = 2^64 - 1: the whole-domain case
    xs[7] = 0
    zeros = torch.zeros((nkeys, 4), dtype=torch.int32, device="cuda")
    keys, _ = engine.gen_level_major(GROTTO, n, G.BYTES, s0s, alphas, zeros)
    del zeros
    seeds = [s0s[:, b].contiguous() for b in (0, 1)]
    sh = [engine.eval_keys(keys, G.BYTES, b, seeds[b], xs) for b in (0, 1)]
    # unsigned alpha <= x on int64 bit patterns: flip the sign bit and compare signed
    flip = torch.tensor(-2**63, dtype=torch.int64, device="cuda")
    le = (alphas ^ flip) <= (xs ^ flip)
    assert bool(((sh[0] ^ sh[1]) == le.to(torch.uint8)).all())
    assert bool((sh[0] <= 1).all()) and bool((sh[1] <= 1).all())

    idx = torch.arange(0, nkeys, 4093, device="cuda")
    cws_km = engine.key_major(_sub_keys(keys, idx)).cpu().numpy()
    p = O.Prg(2)
    for b in (0, 1):
        want_b = O.eval_keys(p, GROTTO, n, G.BYTES, b, seeds[b][idx].cpu().numpy(), cws_km,
                             xs[idx].cpu().numpy().view(np.uint64), nthreads=4)[:, 0].astype(np.uint8)
        assert (sh[b][idx].cpu().numpy() == want_b).all(), b

    for r in (0, 5, 7):
        lo, hi = shard.batch_slice(nkeys, r, world)
        sl = _sub_keys(keys, torch.arange(lo, hi, device="cuda"))
        got = engine.eval_keys(sl, G.BYTES, 0, seeds[0][lo:hi].contiguous(), xs[lo:hi].contiguous())
        assert torch.equal(got, sh[0][lo:hi]), r
