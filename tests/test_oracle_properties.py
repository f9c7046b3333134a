"""The reference's own correctness tests restated against the oracle: they pin the
FSS property (share reconstruction), with the constants of src/dpf_test.cu:30-32
(alpha = 107, beta = 604) and per-run random seeds made deterministic here.
Follows src/dpf_test.cu:58-139,189-266, src/dcf_test.cu:58-135,187-252,
src/half_tree_dpf_test.cu:64-140,205-289,327-451, src/grotto_dcf_test.cu:39-137."""
import numpy as np
import pytest

import groups as G

ALPHA, BETA_X = 107, 604
GROUPS = [G.BYTES, G.U8, G.U16, G.U32, G.U64, G.U128]


def beta_for(g):
    if g in (G.BYTES, G.U128):
        return np.array([BETA_X, 5, 6, 8], dtype=np.int32)
    return np.array([BETA_X, 0, 0, 0], dtype=np.int32)


@pytest.mark.parametrize("g", GROUPS)
@pytest.mark.parametrize("n", [1, 8, 16, 33, 64])
def test_dpf_eval_reconstructs(O, rng, g, n):
    p = O.Prg(2)
    alpha = ALPHA & ((1 << n) - 1)
    s = [rng.i4(), rng.i4()]
    beta = beta_for(g)
    cws = O.dpf_gen(p, n, g, s, alpha, beta)
    xs = {0, alpha, (1 << n) - 1, max(alpha - 1, 0), min(alpha + 1, (1 << n) - 1)} | {rng.below(n) for _ in range(8)}
    for x in xs:
        y = G.add(g, O.dpf_eval(p, n, g, 0, s[0], cws, x), O.dpf_eval(p, n, g, 1, s[1], cws, x))
        want = G.canon(g, beta) if x == alpha else np.zeros(4, np.int32)
        assert (y == want).all(), (g, n, x)


@pytest.mark.parametrize("g", [G.BYTES, G.U32, G.U64])
def test_dpf_eval_all_matches_eval_and_reconstructs(O, rng, g):
    n, p = 10, O.Prg(2)
    s = [rng.i4(), rng.i4()]
    beta = beta_for(g)
    cws = O.dpf_gen(p, n, g, s, ALPHA, beta)
    y0, y1 = O.dpf_eval_all(p, n, g, 0, s[0], cws), O.dpf_eval_all(p, n, g, 1, s[1], cws)
    for x in range(0, 1 << n, 37):
        assert (y0[x] == O.dpf_eval(p, n, g, 0, s[0], cws, x)).all()
    r = G.add_rows(g, y0, y1)
    want = np.zeros_like(r)
    want[ALPHA] = G.canon(g, beta)
    assert (r == want).all()


@pytest.mark.parametrize("threads", [1, 2, 3, 8])
@pytest.mark.parametrize("n", [1, 2, 9, 13])
def test_dpf_eval_all_in_the_reference_task_shape(O, rng, n, threads):
    """The depth-first recursion with OpenMP tasks above ceil(log2(threads)) levels (dpf.cuh:232-303) writes the
    leaves the flattened form writes, for thread counts that are and are not powers of two and for domains
    shallower than the task depth; with the EVP PRG (one context set per thread) as bench.py's CPU leg runs it."""
    for use_evp in (False, True):
        p = O.Prg(2, use_evp=use_evp)
        s = [rng.i4(), rng.i4()]
        cws = O.dpf_gen(p, n, G.U64, s, ALPHA & ((1 << n) - 1), beta_for(G.U64))
        for b in (0, 1):
            want = O.dpf_eval_all(p, n, G.U64, b, s[b], cws)
            assert (O.dpf_eval_all(p, n, G.U64, b, s[b], cws, task_threads=threads) == want).all()
            assert (O.dpf_eval_all_repeat(p, n, G.U64, b, s[b], cws, threads, 3) == want).all()
            assert (O.dpf_eval_all_repeat(p, n, G.U64, b, s[b], cws, 0, 2) == want).all()


@pytest.mark.parametrize("pred", [0, 1])
@pytest.mark.parametrize("g", GROUPS)
@pytest.mark.parametrize("n", [1, 8, 16, 64])
def test_dcf_eval_reconstructs(O, rng, g, n, pred):
    p = O.Prg(4)
    alpha = ALPHA & ((1 << n) - 1)
    s = [rng.i4(), rng.i4()]
    beta = beta_for(g)
    cws = O.dcf_gen(p, n, g, pred, s, alpha, beta)
    xs = {0, alpha, (1 << n) - 1, max(alpha - 1, 0), min(alpha + 1, (1 << n) - 1)} | {rng.below(n) for _ in range(8)}
    for x in xs:
        y = G.add(g, O.dcf_eval(p, n, g, 0, s[0], cws, x), O.dcf_eval(p, n, g, 1, s[1], cws, x))
        hit = x < alpha if pred == 0 else x > alpha
        want = G.canon(g, beta) if hit else np.zeros(4, np.int32)
        assert (y == want).all(), (g, n, x, pred)


@pytest.mark.parametrize("g", [G.BYTES, G.U64])
def test_dcf_eval_all_matches_eval(O, rng, g):
    n, p = 9, O.Prg(4)
    s = [rng.i4(), rng.i4()]
    cws = O.dcf_gen(p, n, g, 0, s, ALPHA, beta_for(g))
    for b in (0, 1):
        ys = O.dcf_eval_all(p, n, g, b, s[b], cws)
        for x in range(0, 1 << n, 13):
            assert (ys[x] == O.dcf_eval(p, n, g, b, s[b], cws, x)).all()


@pytest.mark.parametrize("g", GROUPS)
@pytest.mark.parametrize("n", [1, 2, 8, 16, 64])
def test_half_tree_reconstructs(O, rng, g, n):
    p = O.Prg(1)
    alpha = ALPHA & ((1 << n) - 1)
    s = [rng.i4(), rng.i4()]
    beta = beta_for(g)
    cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
    xs = {0, alpha, (1 << n) - 1} | {rng.below(n) for _ in range(8)}
    for x in xs:
        y = G.add(g, O.ht_eval(p, n, g, 0, s[0], cws, ocw, x), O.ht_eval(p, n, g, 1, s[1], cws, ocw, x))
        want = G.canon(g, beta) if x == alpha else np.zeros(4, np.int32)
        assert (y == want).all(), (g, n, x)


@pytest.mark.parametrize("n", [1, 2, 9])
def test_half_tree_eval_all_matches_eval(O, rng, n):
    p, g = O.Prg(1), G.U64
    s = [rng.i4(), rng.i4()]
    cws, ocw = O.ht_gen(p, n, g, s, ALPHA & ((1 << n) - 1), beta_for(g))
    for b in (0, 1):
        ys = O.ht_eval_all(p, n, g, b, s[b], cws, ocw)
        for x in range(1 << n):
            assert (ys[x] == O.ht_eval(p, n, g, b, s[b], cws, ocw, x)).all()


@pytest.mark.parametrize("alpha", [0, 107, (1 << 10) - 1])
def test_grotto(O, rng, alpha):
    n, p = 10, O.Prg(2)
    s = [rng.i4(), rng.i4()]
    cws = O.grotto_gen(p, n, s, alpha)
    ea = [O.grotto_eval_all(p, n, b, s[b], cws) for b in (0, 1)]
    want = (np.arange(1 << n) >= alpha).astype(np.uint8)
    assert ((ea[0] ^ ea[1]) == want).all()  # src/grotto_dcf_test.cu:39-77
    for b in (0, 1):  # Preprocess + Eval == EvalAll, :79-96
        pt = O.grotto_preprocess(p, n, b, s[b], cws)
        got = np.array([O.grotto_eval(n, pt, x) for x in range(1 << n)], dtype=np.uint8)
        assert (got == ea[b]).all()
    walk = [np.array([O.grotto_eval_walk(p, n, b, s[b], cws, x) for x in range(1 << n)], dtype=np.uint8)
            for b in (0, 1)]
    assert ((walk[0] ^ walk[1]) == want).all()  # the path walk reconstructs the same predicate


def test_grotto_walk_64(O, rng):
    n, p = 64, O.Prg(2)
    s = [rng.i4(), rng.i4()]
    alpha = 0x0123456789ABCDEF
    cws = O.grotto_gen(p, n, s, alpha)
    for x in [0, alpha - 1, alpha, alpha + 1, (1 << 64) - 1, rng.below(64), rng.below(64)]:
        r = O.grotto_eval_walk(p, n, 0, s[0], cws, x) ^ O.grotto_eval_walk(p, n, 1, s[1], cws, x)
        assert r == int(alpha <= x)


# ---------------------------------------------------------------- the same properties with the ChaCha PRG
# (the reference's own test PRG: src/dpf_test.cu, src/dcf_test.cu, src/half_tree_dpf_test.cu use ChaCha too)
@pytest.mark.parametrize("g", [G.BYTES, G.U64, G.U128])
@pytest.mark.parametrize("n", [1, 16, 64, 128])
def test_chacha_schemes_reconstruct(O, rng, g, n):
    alpha = ((rng.next() << 64) | rng.next()) & ((1 << n) - 1)
    s = [rng.i4(), rng.i4()]
    beta = beta_for(g)
    top = (1 << n) - 1
    xs = {0, top, alpha, max(alpha - 1, 0), min(alpha + 1, top)}
    p2, p4, p1 = O.Prg(2, chacha=True), O.Prg(4, chacha=True), O.Prg(1, chacha=True)
    cws = O.dpf_gen(p2, n, g, s, alpha, beta)
    dc = O.dcf_gen(p4, n, g, O.LT, s, alpha, beta)
    hc, ocw = O.ht_gen(p1, n, g, s, alpha, beta)
    for x in xs:
        y = G.add(g, O.dpf_eval(p2, n, g, 0, s[0], cws, x), O.dpf_eval(p2, n, g, 1, s[1], cws, x))
        assert (y == (G.canon(g, beta) if x == alpha else np.zeros(4, np.int32))).all()
        y = G.add(g, O.dcf_eval(p4, n, g, 0, s[0], dc, x), O.dcf_eval(p4, n, g, 1, s[1], dc, x))
        assert (y == (G.canon(g, beta) if x < alpha else np.zeros(4, np.int32))).all()
        y = G.add(g, O.ht_eval(p1, n, g, 0, s[0], hc, ocw, x), O.ht_eval(p1, n, g, 1, s[1], hc, ocw, x))
        assert (y == (G.canon(g, beta) if x == alpha else np.zeros(4, np.int32))).all()


# ---------------------------------------------------------------- VDPF (src/vdpf_test.cu)
@pytest.mark.parametrize("chacha", [True, False])
@pytest.mark.parametrize("g", [G.BYTES, G.U64, G.U128])
def test_vdpf_reconstructs_and_verifies(O, rng, g, chacha):
    """src/vdpf_test.cu: shares reconstruct the point function, both parties compute the same pi~ at every
    point, Prove gives equal proofs, and a tampered key is caught by Verify."""
    n = 16
    p = O.Prg(2, chacha=chacha)
    s = [rng.i4(), rng.i4()]
    beta = beta_for(g)
    cws, cs, ocw, rc = O.vdpf_gen(p, n, g, s, ALPHA, beta)
    assert rc == 0
    xs = [0, ALPHA, ALPHA - 1, ALPHA + 1, (1 << n) - 1] + [rng.below(n) for _ in range(20)]
    pts = [[], []]
    for x in xs:
        ys = []
        for b in (0, 1):
            y, pt = O.vdpf_eval(p, n, g, b, s[b], cws, cs, ocw, x)
            ys.append(y)
            pts[b].append(pt)
        want = G.canon(g, beta) if x == ALPHA else np.zeros(4, np.int32)
        assert (G.add(g, ys[0], ys[1]) == want).all(), x
        assert (pts[0][-1] == pts[1][-1]).all()
    pi0, pi1 = O.vdpf_prove(np.array(pts[0]), cs), O.vdpf_prove(np.array(pts[1]), cs)
    assert (pi0 == pi1).all()
    bad = cws.copy()
    bad[3, 0] ^= 0x10  # a malicious dealer hands party 1 a different correction word
    pts1 = [O.vdpf_eval(p, n, g, 1, s[1], bad, cs, ocw, x)[1] for x in xs]
    assert not (O.vdpf_prove(np.array(pts1), cs) == pi0).all()


@pytest.mark.parametrize("g,mod", [(G.U32, 1000003), (G.U64, (1 << 61) - 1), (G.U8, 251)])
def test_modular_uint_group_reconstructs(O, rng, g, mod):
    """fss::group::Uint<T, mod>, mod != 0 (uint.cuh:11-31; src/group_test.cu): DPF and DCF shares add up to
    beta mod `mod` at / below alpha and to zero elsewhere."""
    O.set_uint_modulus(mod)
    try:
        n = 12
        s = [rng.i4(), rng.i4()]
        beta = rng.i4()
        bits = {G.U8: 8, G.U32: 32, G.U64: 64}[g]
        val = lambda y: int(np.uint32(y[0])) | (int(np.uint32(y[1])) << 32)
        bval = (val(beta) & ((1 << bits) - 1)) % mod
        p2, p4 = O.Prg(2), O.Prg(4)
        cws, dcw = O.dpf_gen(p2, n, g, s, ALPHA, beta), O.dcf_gen(p4, n, g, O.LT, s, ALPHA, beta)
        for x in [0, ALPHA - 1, ALPHA, ALPHA + 1, (1 << n) - 1] + [rng.below(n) for _ in range(20)]:
            y = [val(O.dpf_eval(p2, n, g, b, s[b], cws, x)) for b in (0, 1)]
            assert all(v < mod for v in y) and (y[0] + y[1]) % mod == (bval if x == ALPHA else 0)
            y = [val(O.dcf_eval(p4, n, g, b, s[b], dcw, x)) for b in (0, 1)]
            assert (y[0] + y[1]) % mod == (bval if x < ALPHA else 0)
    finally:
        O.set_uint_modulus(0)
