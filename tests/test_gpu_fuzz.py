"""Randomised parity sweep: scheme x in_bits x group x party x batch size drawn from a seeded stream, HIP path
against the oracle bit for bit (shared key, shared prefix, independent keys through device Gen + level-major
relayout).  Complements the fixed matrices of test_gpu_parity.py with the sizes in between."""
import numpy as np
import pytest

import groups as G
from conftest import SplitMix64
from test_gpu_parity import DCF, DPF, HT, T, rand_below, xs_tensor

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GROUPS = [G.BYTES, G.U8, G.U16, G.U32, G.U64, G.U128]


def draw(rng):
    scheme = [DPF, DCF, HT][rng.next() % 3]
    n = [1, 2, 3, 5, 8, 13, 21, 31, 32, 34, 47, 63, 64, 66, 97, 127, 128][rng.next() % 17]
    g = GROUPS[rng.next() % len(GROUPS)]
    count = [1, 2, 63, 64, 65, 127, 129, 1000, 2049][rng.next() % 9]
    return scheme, n, g, count


@pytest.mark.parametrize("case", range(48))
def test_random_configuration_vs_oracle(engine, O, case):
    rng = SplitMix64(0xF5500000 + case)
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
