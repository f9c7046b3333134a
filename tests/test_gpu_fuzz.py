"""Randomised parity sweep: scheme x in_bits x group x party x batch size drawn from a seeded stream, HIP path
against the oracle bit for bit (shared key, shared prefix, independent keys through device Gen + level-major
relayout), and the same for EvalAll (whole domains, aligned sub-blocks, key batches).  Complements the fixed
matrices of test_gpu_parity.py with the sizes in between.  FSS_AMD_FUZZ_CASES=N widens every sweep to N seeded
cases (48 / 40 by default; the 20000-case run is recorded in DESIGN.md section 2) and FSS_AMD_FUZZ_SEED=S draws a
different stream of them."""
import os

import numpy as np
import pytest

import groups as G
from conftest import SplitMix64
from test_gpu_parity import DCF, DPF, GROTTO, HT, T, rand_below, xs_tensor

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GROUPS = [G.BYTES, G.U8, G.U16, G.U32, G.U64, G.U128]
CASES = int(os.environ.get("FSS_AMD_FUZZ_CASES", "0"))
SEED = int(os.environ.get("FSS_AMD_FUZZ_SEED", "0"), 0) << 32  # a different stream of cases per soak run


def draw(rng):
    scheme = [DPF, DCF, HT][rng.next() % 3]
    n = [1, 2, 3, 5, 8, 13, 21, 31, 32, 34, 47, 63, 64, 66, 97, 127, 128][rng.next() % 17]
    g = GROUPS[rng.next() % len(GROUPS)]
    count = [1, 2, 63, 64, 65, 127, 129, 1000, 2049][rng.next() % 9]
    return scheme, n, g, count


@pytest.mark.parametrize("case", range(CASES or 48))
def test_random_configuration_vs_oracle(engine, O, case):
    rng = SplitMix64(SEED + 0xF5500000 + case)
    scheme, n, g, count = draw(rng)
    mul = {DPF: 2, DCF: 4, HT: 1}[scheme]
    p = O.Prg(mul)
    pred = rng.next() & 1
    s = [rng.i4(clear_lsb=bool(rng.next() & 1)), rng.i4(clear_lsb=bool(rng.next() & 1))]
    alpha, beta = rand_below(rng, n), rng.i4(clear_lsb=False)
    xs = [alpha if (i % 5 == 0) else rand_below(rng, n) for i in range(count)]
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, pred, s, alpha, beta)
    else:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
    xs_np = xs if n > 64 else np.array(xs, dtype=np.uint64)
    for b in (0, 1):
        want = O.eval_points(p, scheme, n, g, b, s[b], cws, xs_np, ocw=ocw, nthreads=4)
        got = engine.eval_points(scheme, n, g, b, T(s[b]), T(cws), xs_tensor(xs, n),
                                 ocw=None if ocw is None else T(ocw)).cpu().numpy()
        assert (got == want).all(), (scheme, n, g, b, count)
        if n <= 64 and n - (scheme == HT) >= 1:
            k = 1 + rng.next() % min(n - (scheme == HT), 12)
            got2 = engine.eval_points_shared(scheme, n, g, b, T(s[b]), T(cws), xs_tensor(xs, n),
                                             ocw=None if ocw is None else T(ocw), prefix_bits=k).cpu().numpy()
            assert (got2 == want).all(), (scheme, n, g, b, count, k)
    # the same key replicated as `count` independent keys through the device Gen, level-major
    m = min(count, 257)
    s0s = np.broadcast_to(np.stack(s), (m, 2, 4)).copy()
    alphas, betas = [alpha] * m, np.broadcast_to(beta, (m, 4)).copy()
    keys, ocws = engine.gen_level_major(scheme, n, g, T(s0s), xs_tensor(alphas, n), T(betas), pred=pred)
    for b in (0, 1):
        want = O.eval_points(p, scheme, n, g, b, s[b], cws, xs_np[:m] if n <= 64 else xs[:m], ocw=ocw, nthreads=4)
        got = engine.eval_keys(keys, g, b, T(np.ascontiguousarray(s0s[:, b])), xs_tensor(xs[:m], n), ocws=ocws).cpu().numpy()
        assert (got == want).all(), (scheme, n, g, b, m)


def oracle_tree(O, p, scheme, n, g, b, s0, cws, ocw):
    if scheme == DPF:
        return O.dpf_eval_all(p, n, g, b, s0, cws)
    if scheme == DCF:
        return O.dcf_eval_all(p, n, g, b, s0, cws)
    if scheme == HT:
        return O.ht_eval_all(p, n, g, b, s0, cws, ocw)
    return O.grotto_eval_all(p, n, b, s0, cws)


@pytest.mark.parametrize("case", range(CASES or 40))
def test_random_eval_all_vs_oracle(engine, O, case):
    """Whole domain and one random aligned power-of-two block of it, both parties; every third case a key batch."""
    rng = SplitMix64(SEED + 0xEA110000 + case)
    scheme = [DPF, DCF, HT, GROTTO][rng.next() % 4]
    n = 1 + rng.next() % 17
    g = G.BYTES if scheme == GROTTO else GROUPS[rng.next() % len(GROUPS)]
    p = O.Prg({DPF: 2, DCF: 4, HT: 1, GROTTO: 2}[scheme])
    pred = rng.next() & 1
    s = [rng.i4(clear_lsb=bool(rng.next() & 1)), rng.i4(clear_lsb=bool(rng.next() & 1))]
    alpha, beta = rng.below(n), rng.i4(clear_lsb=False)
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, pred, s, alpha, beta)
    elif scheme == HT:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
    else:
        cws = O.grotto_gen(p, n, s, alpha)
    lo = 1 if scheme == HT else 0  # a Half-Tree block holds whole sibling pairs
    r = lo + rng.next() % (n - lo + 1)
    first = (rng.next() % (1 << (n - r))) << r
    for b in (0, 1):
        want = oracle_tree(O, p, scheme, n, g, b, s[b], cws, ocw)
        dev = lambda **kw: engine.eval_all(scheme, n, g, b, T(s[b]), T(cws), ocw=None if ocw is None else T(ocw),
                                           **kw).cpu().numpy()
        assert (dev() == want).all(), (scheme, n, g, b)
        if scheme != GROTTO:  # Grotto blocks need the carry of the blocks before them (test_grotto_eval_all_sharded_*)
            assert (dev(first_leaf=first, nleaves=1 << r) == want[first:first + (1 << r)]).all(), (scheme, n, g, b, first, r)
    if case % 3 == 0 and scheme != GROTTO and n <= 12:
        nkeys = 1 + rng.next() % 70
        s0s = np.stack([np.stack([rng.i4(), rng.i4()]) for _ in range(nkeys)])
        alphas = np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64)
        betas = np.stack([rng.i4() for _ in range(nkeys)])
        kc, ko = O.gen_keys(p, scheme, n, g, s0s, alphas, betas, pred=pred, nthreads=4)
        b = rng.next() & 1
        got = engine.eval_all_batch(scheme, n, g, b, T(np.ascontiguousarray(s0s[:, b])), T(kc),
                                    ocws=None if ko is None else T(ko)).cpu().numpy()
        for k in {0, nkeys // 2, nkeys - 1}:
            want = oracle_tree(O, p, scheme, n, g, b, s0s[k, b], kc[k], None if ko is None else ko[k])
            assert (got[k] == want).all(), (scheme, n, g, b, nkeys, k)


@pytest.mark.parametrize("case", range(CASES or 24))
def test_random_grotto_walk_vs_oracle(engine, O, case):
    """Grotto path walk (fss_amd_grotto_eval_keys: taken child per level, siblings batched through the per-wave
    queue) against the oracle's walk and the predicate, random domain sizes and ragged key counts."""
    rng = SplitMix64(SEED + 0x6A077000 + case)
    n = 1 + rng.next() % 64
    nkeys = [1, 2, 63, 64, 65, 127, 128, 129, 200, 1000][rng.next() % 10]
    p = O.Prg(2)
    s0s = np.stack([np.stack([rng.i4(clear_lsb=bool(rng.next() & 1)), rng.i4()]) for _ in range(nkeys)])
    alphas = np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64)
    cws, _ = O.gen_keys(p, GROTTO, n, G.BYTES, s0s, alphas, np.zeros((nkeys, 4), np.int32), nthreads=4)
    top = (1 << n) - 1
    xs = np.array([[alphas[i], top, 0, rng.below(n), (int(alphas[i]) - 1) & top][i % 5] for i in range(nkeys)], dtype=np.uint64)
    keys = engine.relayout(GROTTO, n, T(cws))
    sh = []
    for b in (0, 1):
        want = O.eval_keys(p, GROTTO, n, G.BYTES, b, s0s[:, b], cws, xs, nthreads=4)[:, 0].astype(np.uint8)
        got = engine.eval_keys(keys, G.BYTES, b, T(s0s[:, b].copy()), xs_tensor(xs, n)).cpu().numpy()
        assert (got == want).all(), (n, nkeys, b)
        sh.append(got)
    assert ((sh[0] ^ sh[1]) == (alphas <= xs)).all(), (n, nkeys)
