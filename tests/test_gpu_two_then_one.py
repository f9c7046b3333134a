"""The point walk's "two, then one" split (eval_points_kernel: evaluations [0, n_body) two per lane, the rest one per
lane) only engages from four 64-evaluation units per resident wave on (2^20 evaluations on an MI355X), so
the small oracle-checked cases never cross its region boundary.  These do, at ragged sizes: batches that are not a
multiple of 64, of 128, of the grid, just above and below the thresholds, in the fixed-share and in the dynamic regime --
every output of both parties through the FSS property (y0 + y1 = beta at alpha, 0 elsewhere), and a sample that covers
both ends of both regions, the boundary itself and a stride in between against the oracle."""
import numpy as np
import pytest

import groups as G
from conftest import SplitMix64

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

DPF, DCF, HT = 0, 1, 2


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def u64_of(t):
    return (t[:, 0].to(torch.int64) & 0xFFFFFFFF) | (t[:, 1].to(torch.int64) << 32)


def _sample(n):
    """Indices at both ends, around every multiple of 2^19 (region boundaries are multiples of 128 near fractions of
    n), and a stride."""
    idx = set(range(0, min(n, 300))) | set(range(max(0, n - 300), n)) | set(range(0, n, 8191))
    for c in range(1 << 19, n, 1 << 19):
        idx |= set(range(c - 130, min(n, c + 130)))
    for frac in (0.5, 0.75, 0.86, 0.92):
        c = int(n * frac) // 128 * 128
        idx |= set(range(max(0, c - 130), min(n, c + 130)))
    return np.array(sorted(idx), dtype=np.int64)


# sizes: just above the fixed-share threshold (256 evaluations per resident wave = 2^20), ragged in every way;
# 2^23 + ragged: the dynamic regime for a 32-level walk (65 536 blocks per resident wave)
SIZES = [(1 << 20) + 1, (1 << 20) + 12345, 3 * (1 << 19) + 77, (1 << 21) - 1, (1 << 23) + 999]


@pytest.mark.parametrize("plan", ["plain", "auto"])
@pytest.mark.parametrize("n_points", SIZES)
@pytest.mark.parametrize("scheme,in_bits", [(DPF, 32), (HT, 40), (DCF, 32)])
def test_shared_key_ragged_large_batches(engine, engine_plain, O, scheme, in_bits, n_points, plan):
    """plain: the walk from the root (FSS_AMD_POINTS_PLAIN), whose "two, then one" regions these sizes cross.  auto: what
    the same call does by default for batches this size -- the shared-prefix plan with the library's own prefix_bits."""
    engine = engine if plan == "auto" else engine_plain
    rng = SplitMix64(0xBEEF00 + scheme * 1000 + n_points % 997)
    g = G.U64
    mul = {DPF: 2, DCF: 4, HT: 1}[scheme]
    p = O.Prg(mul)
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(in_bits), rng.i4()
    beta[2:] = 0
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, in_bits, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, in_bits, g, 0, s, alpha, beta)
    else:
        cws, ocw = O.ht_gen(p, in_bits, g, s, alpha, beta)
    gen = torch.Generator(device="cuda").manual_seed(n_points)
    if in_bits <= 32:
        xs = torch.randint(-2**31, 2**31, (n_points,), dtype=torch.int32, device="cuda", generator=gen)
        a_t = alpha - (1 << 32) if alpha >= (1 << 31) else alpha
        xs_u = lambda t: t.to(torch.int64) & 0xFFFFFFFF
    else:
        xs = torch.randint(0, 1 << in_bits, (n_points,), dtype=torch.int64, device="cuda", generator=gen)
        a_t = alpha
        xs_u = lambda t: t
    xs[::3] = a_t                      # a third of the points hit alpha
    xs[-1] = a_t
    cd = T(cws).cuda()
    od = None if ocw is None else T(ocw).cuda()
    ys = [engine.eval_points(scheme, in_bits, g, b, T(s[b]).cuda(), cd, xs, ocw=od) for b in (0, 1)]
    assert engine.last_kernel()[0].endswith("_eval_points_shared") == (plan == "auto")
    assert ys[0].shape == (n_points, 4)
    total = u64_of(ys[0]) + u64_of(ys[1])
    bl = (int(np.uint32(beta[0])) | (int(np.uint32(beta[1])) << 32))
    bl = bl - (1 << 64) if bl >= (1 << 63) else bl
    xu = xs_u(xs)
    hit = (xu == alpha) if scheme != DCF else (xu < alpha)
    want = torch.where(hit, torch.full_like(total, bl), torch.zeros_like(total))
    assert bool((total == want).all()), "reconstruction"
    idx = _sample(n_points)
    it = torch.from_numpy(idx).cuda()
    xh = xu[it].cpu().numpy().astype(np.uint64)
    for b in (0, 1):
        ref = O.eval_points(p, scheme, in_bits, g, b, s[b], cws, xh, ocw=ocw, nthreads=4)
        assert (ys[b][it].cpu().numpy() == ref).all(), b


@pytest.mark.parametrize("nkeys", [(1 << 20) + 17, 3 * (1 << 19) + 640 + 5])
@pytest.mark.parametrize("scheme,in_bits", [(DPF, 32), (HT, 64), (DCF, 32), (DPF, 16), (DPF, 64)])
def test_independent_keys_ragged_large_batches(engine, O, scheme, in_bits, nkeys):
    """The cells of the metric's independent-keys grid (bench.py: dpf_keys, dcf_keys, dpf_keys_n16, dpf_keys_n64, ht_c5's
    shape) at ragged sizes across the two-then-one region boundary: every output of both parties through the FSS
    property, a sample across both regions against the oracle on the same keys brought back key-major."""
    g = G.U64
    gen = torch.Generator(device="cuda").manual_seed(nkeys + scheme + in_bits)
    s0s = torch.randint(-2**31, 2**31, (nkeys, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    betas = torch.randint(-2**31, 2**31, (nkeys, 4), dtype=torch.int32, device="cuda", generator=gen)
    if in_bits <= 32:
        alphas = torch.randint(-2**31, 2**31, (nkeys,), dtype=torch.int32, device="cuda", generator=gen)
        xs = torch.randint(-2**31, 2**31, (nkeys,), dtype=torch.int32, device="cuda", generator=gen)
        if in_bits < 32:
            alphas &= (1 << in_bits) - 1
            xs &= (1 << in_bits) - 1
    else:
        alphas = torch.randint(-2**63, 2**63 - 1, (nkeys,), dtype=torch.int64, device="cuda", generator=gen)
        xs = torch.randint(-2**63, 2**63 - 1, (nkeys,), dtype=torch.int64, device="cuda", generator=gen)
    xs[::2] = alphas[::2]
    xs[-1] = alphas[-1]
    if scheme == DCF:   # a third of the points just below their alpha, where a comparison share is beta
        xs[1::3] = torch.where(alphas[1::3] != 0, alphas[1::3] - 1, alphas[1::3])
    keys, ocws = engine.gen_level_major(scheme, in_bits, g, s0s, alphas, betas)
    seeds = [s0s[:, b].contiguous() for b in (0, 1)]
    y = [engine.eval_keys(keys, g, b, seeds[b], xs, ocws=ocws) for b in (0, 1)]
    total = u64_of(y[0]) + u64_of(y[1])
    bl = (betas[:, 0].to(torch.int64) & 0xFFFFFFFF) | (betas[:, 1].to(torch.int64) << 32)
    if scheme == DCF:   # x < alpha as unsigned in_bits-bit values (in_bits = 32: int32 bit patterns)
        ux, ua = xs.to(torch.int64) & 0xFFFFFFFF, alphas.to(torch.int64) & 0xFFFFFFFF
        hit = ux < ua
    else:
        hit = xs == alphas
    assert bool((total == torch.where(hit, bl, torch.zeros_like(bl))).all())
    # a sample across both regions against the oracle, on the same keys brought back key-major
    from test_gpu_configs import _sub_keys
    idx = torch.from_numpy(_sample(nkeys)).cuda()
    km = engine.key_major(_sub_keys(keys, idx)).cpu().numpy()
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    xh = xs[idx].cpu().numpy()
    xh = (xh.view(np.uint32) if xh.dtype == np.int32 else xh.view(np.uint64)).astype(np.uint64)
    for b in (0, 1):
        ref = O.eval_keys(p, scheme, in_bits, g, b, seeds[b][idx].cpu().numpy(), km, xh,
                          ocws=None if ocws is None else ocws[idx].cpu().numpy(), nthreads=4)
        assert (y[b][idx].cpu().numpy() == ref).all(), b
