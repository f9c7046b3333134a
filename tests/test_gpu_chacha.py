"""ChaCha PRG on the device (fss::prg::ChaCha<mul>, include/fss/prg/chacha.cuh): the same kernels
instantiated without tables, bit-compared with the oracle's restatement (itself pinned to RFC 7539 for
the permutation; the reference holds no ChaCha vectors, and its Python binding draws a random nonce per
process, so there are no reference outputs to pin the construction to beyond the source)."""
import numpy as np
import torch
import pytest

import groups as G
from conftest import SplitMix64
from test_gpu_parity import DCF, DPF, GROTTO, HT, T, beta_for, edge_xs, rand_below, xs_tensor

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

NONCE = (0x0BADC0DE, 0x5EED1234)


@pytest.fixture(scope="module")
def ceng():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this container")
    import fss_amd
    return fss_amd.Engine(0, prg="chacha", chacha_nonce=NONCE)


def cprg(O, mul):
    return O.Prg(mul, chacha=True, nonce=NONCE)


@pytest.mark.parametrize("mul", [1, 2, 4])
def test_chacha_prg_vs_oracle(ceng, O, mul):
    rng = SplitMix64(3)
    seeds = np.stack([rng.i4(clear_lsb=False) for _ in range(500)])
    got = ceng.prg_gen(mul, T(seeds)).cpu().numpy()
    p = cprg(O, mul)
    for i in range(0, 500, 3):
        assert (got[i] == p.gen(seeds[i])).all()


@pytest.mark.parametrize("mul", [1, 2, 4])
@pytest.mark.parametrize("n", [1, 64, 65, 128, 129, 4097, (1 << 20) + 131])
def test_chacha_prg_ragged_sizes(ceng, O, mul, n):
    gen = torch.Generator(device="cuda").manual_seed(77 * mul + n % 997)
    seeds = torch.randint(-2**31, 2**31, (n, 4), dtype=torch.int32, device="cuda", generator=gen)
    got = ceng.prg_gen(mul, seeds).cpu().numpy()
    sh = seeds.cpu().numpy()
    p = cprg(O, mul)
    idx = sorted(set(range(0, n, max(1, n // 64))) | set(range(max(0, n - 140), n)) | set(range(min(n, 140))))
    for i in idx:
        assert (got[i] == p.gen(sh[i])).all(), i


@pytest.mark.parametrize("g", [G.BYTES, G.U32, G.U64, G.U128])
@pytest.mark.parametrize("n", [1, 16, 32, 64, 128])
@pytest.mark.parametrize("scheme", [DPF, DCF, HT])
def test_chacha_eval_points_vs_oracle(ceng, O, scheme, n, g):
    rng = SplitMix64(2000 * scheme + 10 * n + g)
    p = cprg(O, {DPF: 2, DCF: 4, HT: 1}[scheme])
    s = [rng.i4(), rng.i4()]
    alpha, beta = rand_below(rng, n), beta_for(rng)
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, rng.next() & 1, s, alpha, beta)
    else:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
    xs = edge_xs(rng, n, alpha, 200)
    for b in (0, 1):
        want = O.eval_points(p, scheme, n, g, b, s[b], cws, xs if n > 64 else np.array(xs, dtype=np.uint64), ocw=ocw)
        got = ceng.eval_points(scheme, n, g, b, T(s[b]), T(cws), xs_tensor(xs, n),
                               ocw=None if ocw is None else T(ocw)).cpu().numpy()
        assert (got == want).all(), (scheme, n, g, b)
        if n <= 64:
            got2 = ceng.eval_points_shared(scheme, n, g, b, T(s[b]), T(cws), xs_tensor(xs, n),
                                           ocw=None if ocw is None else T(ocw), prefix_bits=min(5, n - (scheme == HT)) or None)
            assert (got2.cpu().numpy() == want).all()


@pytest.mark.parametrize("scheme,n,g,pred", [(DPF, 32, G.U64, 0), (DCF, 32, G.U64, 1), (HT, 64, G.U64, 0),
                                             (DPF, 100, G.BYTES, 0), (DCF, 20, G.U128, 0), (HT, 9, G.BYTES, 0)])
def test_chacha_gen_relayout_eval_keys_vs_oracle(ceng, O, scheme, n, g, pred):
    rng = SplitMix64(99 + 1000 * scheme + n)
    nkeys = 150
    p = cprg(O, {DPF: 2, DCF: 4, HT: 1}[scheme])
    s0s = np.stack([np.stack([rng.i4(), rng.i4()]) for _ in range(nkeys)])
    alphas = [rand_below(rng, n) for _ in range(nkeys)]
    xs = [alphas[i] if i % 2 else rand_below(rng, n) for i in range(nkeys)]
    betas = np.stack([rng.i4() for _ in range(nkeys)])
    want_cws, want_ocw = O.gen_keys(p, scheme, n, g, s0s, alphas if n > 64 else np.array(alphas, dtype=np.uint64),
                                    betas, pred=pred, nthreads=4)
    cws, ocws = ceng.gen(scheme, n, g, T(s0s), xs_tensor(alphas, n), T(betas), pred=pred)
    assert (cws.cpu().numpy() == want_cws).all()
    if scheme == HT:
        assert (ocws.cpu().numpy() == want_ocw).all()
    keys = ceng.relayout(scheme, n, cws)
    lm, lm_ocws = ceng.gen_level_major(scheme, n, g, T(s0s), xs_tensor(alphas, n), T(betas), pred=pred)
    for name in ("cw_s", "cw_v", "tr_bits", "out_cw", "extra"):  # Gen straight into the level-major layout
        a, b = getattr(lm, name), getattr(keys, name)
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), name
    if scheme == HT:
        assert torch.equal(lm_ocws, ocws)
    for b in (0, 1):
        want = O.eval_keys(p, scheme, n, g, b, s0s[:, b], want_cws, xs if n > 64 else np.array(xs, dtype=np.uint64),
                           ocws=want_ocw, nthreads=4)
        got = ceng.eval_keys(keys, g, b, T(s0s[:, b].copy()), xs_tensor(xs, n), ocws=ocws).cpu().numpy()
        assert (got == want).all()


@pytest.mark.parametrize("n", [1, 6, 13, 17])
@pytest.mark.parametrize("scheme", [DPF, DCF, HT])
def test_chacha_eval_all_vs_oracle(ceng, O, scheme, n):
    g = G.U64
    rng = SplitMix64(8 + 100 * scheme + n)
    p = cprg(O, {DPF: 2, DCF: 4, HT: 1}[scheme])
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), beta_for(rng)
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
        ref = O.dpf_eval_all(p, n, g, 1, s[1], cws)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, 0, s, alpha, beta)
        ref = O.dcf_eval_all(p, n, g, 1, s[1], cws)
    else:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
        ref = O.ht_eval_all(p, n, g, 1, s[1], cws, ocw)
    got = ceng.eval_all(scheme, n, g, 1, T(s[1]), T(cws), ocw=None if ocw is None else T(ocw)).cpu().numpy()
    assert (got == ref).all()


def test_chacha_grotto_and_python_surface(ceng, O):
    import fss_amd
    n = 11
    rng = SplitMix64(12)
    p = cprg(O, 2)
    s = [rng.i4(), rng.i4()]
    cws = O.grotto_gen(p, n, s, 777)
    for b in (0, 1):
        assert (ceng.eval_all(GROTTO, n, G.BYTES, b, T(s[b]), T(cws)).cpu().numpy()
                == O.grotto_eval_all(p, n, b, s[b], cws)).all()
    # fss_crypto-style use with the binding's default PRG name
    d = fss_amd.Dpf(16, "bytes", prg="chacha", engine=ceng)
    s0s = T(np.stack(s))
    beta = torch.tensor([0, 0, 0, 604], dtype=torch.int32)
    c = d.gen(s0s, 107, beta)
    assert c.shape == (17, 8) and c.device.type == "cpu"
    y0, y1 = d.eval(0, s0s[0], c, 107), d.eval(1, s0s[1], c, 107)
    assert (torch.bitwise_xor(y0, y1) == beta).all()
    assert (torch.bitwise_xor(d.eval(0, s0s[0], c, 50), d.eval(1, s0s[1], c, 50)) == 0).all()
    assert d.eval_all(0, s0s[0], c).shape == (1 << 16, 4)


@pytest.mark.parametrize("scheme,n,nkeys", [(DPF, 9, 21), (DCF, 7, 5), (HT, 10, 64)])
def test_chacha_eval_all_batch_and_key_major(ceng, O, scheme, n, nkeys):
    """Batched full-domain evaluation and the level-major -> key-major conversion with the ChaCha PRG."""
    g = G.U64
    rng = SplitMix64(6100 + 10 * scheme + n)
    p = cprg(O, {DPF: 2, DCF: 4, HT: 1}[scheme])
    s0s = np.stack([np.stack([rng.i4(), rng.i4()]) for _ in range(nkeys)])
    alphas = np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64)
    betas = np.stack([rng.i4() for _ in range(nkeys)])
    cws, ocws = O.gen_keys(p, scheme, n, g, s0s, alphas, betas, pred=O.LT, nthreads=4)
    keys, docws = ceng.gen_level_major(scheme, n, g, T(s0s), xs_tensor(list(map(int, alphas)), n), T(betas))
    km = ceng.key_major(keys)
    assert (km.cpu().numpy() == cws).all()
    ref = {DPF: lambda b, k: O.dpf_eval_all(p, n, g, b, s0s[k, b], cws[k]),
           DCF: lambda b, k: O.dcf_eval_all(p, n, g, b, s0s[k, b], cws[k]),
           HT: lambda b, k: O.ht_eval_all(p, n, g, b, s0s[k, b], cws[k], ocws[k])}[scheme]
    for b in (0, 1):
        got = ceng.eval_all_batch(scheme, n, g, b, T(np.ascontiguousarray(s0s[:, b])), km, ocws=docws).cpu().numpy()
        for k in range(0, nkeys, max(1, nkeys // 7)):
            assert (got[k] == ref(b, k)).all(), (scheme, n, b, k)
