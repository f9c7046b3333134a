"""Parity of the HIP path (through the C ABI, libfss_amd.so) with the CPU oracle and
with the reference's known answers.  Bit-exact: everything on this path is integer /
XOR arithmetic."""
import hashlib
import os

import numpy as np
import pytest

import groups as G
from conftest import SplitMix64
from kat import load, words

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GROUPS = [G.BYTES, G.U8, G.U16, G.U32, G.U64, G.U128]
DPF, DCF, HT, GROTTO = 0, 1, 2, 3


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def xs_tensor(xs, n):
    if n > 64:  # __uint128_t as (lo, hi) pairs
        m = (1 << 64) - 1
        a = np.array([[int(v) & m, int(v) >> 64] for v in xs], dtype=np.uint64)
        return T(a.view(np.int64))
    a = np.array(xs, dtype=np.uint64)
    return T(a.astype(np.uint32).view(np.int32)) if n <= 32 else T(a.view(np.int64))


def rand_below(rng, n):
    return rng.below(n) if n <= 64 else ((rng.next() << 64) | rng.next()) & ((1 << n) - 1)


def edge_xs(rng, n, alpha, extra):
    top = (1 << n) - 1
    xs = [0, top, alpha, max(alpha - 1, 0), min(alpha + 1, top)] + [rand_below(rng, n) for _ in range(extra)]
    return xs


def beta_for(rng):
    return rng.i4()


# ---------------------------------------------------------------- PRG
def test_prg_matches_reference_kat(engine):
    K = load()
    out = engine.prg_gen(4, T(words(K["s0"])).view(1, 4)).cpu().numpy()[0]
    for i in range(4):
        assert (out[i] == words(K["prg4"][i])).all()


@pytest.mark.parametrize("mul", [1, 2, 4])
def test_prg_matches_oracle(engine, O, mul):
    rng = SplitMix64(11)
    seeds = np.stack([rng.i4(clear_lsb=False) for _ in range(1000)])
    got = engine.prg_gen(mul, T(seeds)).cpu().numpy()
    p = O.Prg(mul)
    for i in range(0, 1000, 7):
        assert (got[i] == p.gen(seeds[i])).all()


@pytest.mark.parametrize("mul", [1, 2, 4])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 127, 128, 129, 191, 1000, 4097, (1 << 16) + 77, (1 << 21) + 131])
def test_prg_ragged_sizes(engine, O, mul, n):
    """Every batch shape of the raw PRG launch: one seed, either side of a wave and of the two-seeds-per-lane chunk
    (mul = 1), fixed shares and the dynamic hand-out; the ragged end is checked seed by seed."""
    gen = torch.Generator(device="cuda").manual_seed(1000 * mul + n % 997)
    seeds = torch.randint(-2**31, 2**31, (n, 4), dtype=torch.int32, device="cuda", generator=gen)
    got = engine.prg_gen(mul, seeds).cpu().numpy()
    assert got.shape == (n, mul, 4)
    sh = seeds.cpu().numpy()
    p = O.Prg(mul)
    idx = sorted(set(range(0, n, max(1, n // 128))) | set(range(max(0, n - 200), n)) | set(range(min(n, 200))))
    for i in idx:
        assert (got[i] == p.gen(sh[i])).all(), i
    if n > 4097:  # the same seeds through launches with fixed shares: every output
        for lo in range(0, n, 1 << 14):
            part = engine.prg_gen(mul, seeds[lo:lo + (1 << 14)].contiguous()).cpu().numpy()
            assert (part == got[lo:lo + (1 << 14)]).all(), lo


# ---------------------------------------------------------------- reference KATs on the GPU
def test_reference_te0_on_gpu(engine):
    """The round table the kernels fill LDS from is the reference's Te0 (te0[0] = c66363a5, te0[1] =
    f87c7c84 as the reference stores them, big-endian words; here little-endian, so byte-swapped)."""
    import ctypes as C
    K = load()
    buf = (C.c_uint32 * 256)()
    from fss_amd import _lib
    _lib.check(engine._L.fss_amd_ctx_read_te0(engine._h, buf), "fss_amd_ctx_read_te0")
    for i, want in enumerate(K["te0_be"]):
        assert int.from_bytes(int(buf[i]).to_bytes(4, "little"), "big") == int(want, 16)
    # every entry has the (2S, S, S, 3S) shape over GF(2^8)
    for x in range(256):
        b = int(buf[x]).to_bytes(4, "little")
        s = b[1]
        assert b[2] == s and b[0] == ((s << 1) ^ (0x1B if s & 0x80 else 0)) & 0xFF and b[3] == b[0] ^ s


def test_reference_kats_on_gpu(engine):
    import fss_amd
    K = load()
    s0, s1, beta = T(words(K["s0"])), T(words(K["s1"])), T(words(K["beta"]))
    s0s = torch.stack([s0, s1])

    k = K["dpf8_bytes"]
    d = fss_amd.Dpf(8, "bytes")
    cws = d.gen(s0s, k["alpha"], beta)
    assert (cws[0, :4].numpy() == words(k["cw0_s"])).all() and int(cws[0, 4]) == k["cw0_tr"]
    assert (cws[8, :4].numpy() == words(k["cw8_s"])).all()
    assert (d.eval(0, s0, cws, 42).numpy() == words(k["y0_42"])).all()
    assert (d.eval(1, s1, cws, 42).numpy() == words(k["y1_42"])).all()
    assert (d.eval(0, s0, cws, 100).numpy() == words(k["y0_100"])).all()
    assert hashlib.sha256(d.eval_all(0, s0, cws).numpy().tobytes()).hexdigest() == k["evalall_sha256_p0"]
    assert hashlib.sha256(d.eval_all(1, s1, cws).numpy().tobytes()).hexdigest() == k["evalall_sha256_p1"]

    k = K["dcf8_bytes_lt"]
    d = fss_amd.Dcf(8, "bytes", pred="lt")
    cws = d.gen(s0s, k["alpha"], beta)
    assert (cws[0, :4].numpy() == words(k["cw0_s"])).all() and (cws[0, 4:].numpy() == words(k["cw0_v"])).all()
    assert hashlib.sha256(cws.numpy().tobytes()).hexdigest() == k["cws_sha256"]
    assert (d.eval(0, s0, cws, 10).numpy() == words(k["y0_10"])).all()
    assert (d.eval(1, s1, cws, 10).numpy() == words(k["y1_10"])).all()
    assert hashlib.sha256(d.eval_all(0, s0, cws).numpy().tobytes()).hexdigest() == k["evalall_sha256_p0"]

    k = K["dpf32_u64"]
    d = fss_amd.Dpf(32, "uint")  # uint -> Uint<uint32_t> at in_bits 32 in the binding; the KAT is Uint<uint64_t>
    d64 = fss_amd.Dpf(32, "u64")
    cws = d64.gen(s0s, k["alpha"], beta)
    assert (cws[32, :4].numpy() == words(k["cw32_s"])).all()
    assert (d64.eval(0, s0, cws, k["alpha"]).numpy() == words(k["y0_alpha"])).all()
    assert (d64.eval(1, s1, cws, k["alpha"]).numpy() == words(k["y1_alpha"])).all()
    assert (d64.eval(0, s0, cws, k["x_other"]).numpy() == words(k["y0_other"])).all()
    assert d._g == G.U32

    k = K["ht64_u64"]
    h = fss_amd.HalfTreeDpf(64, "u64")
    cws, ocw = h.gen(s0s, k["alpha"], beta)
    assert (cws[0, :4].numpy() == words(k["cw0_s"])).all() and int(cws[63, 4]) == k["cw63_extra"]
    assert (ocw.numpy() == words(k["ocw"])).all()
    assert (h.eval(0, s0, cws, ocw, k["alpha"]).numpy() == words(k["y0_alpha"])).all()
    assert (h.eval(1, s1, cws, ocw, k["alpha"]).numpy() == words(k["y1_alpha"])).all()
    assert (h.eval(0, s0, cws, ocw, k["x_other"]).numpy() == words(k["y0_other"])).all()


# ---------------------------------------------------------------- one key, many points
@pytest.mark.parametrize("g", GROUPS)
@pytest.mark.parametrize("n", [1, 7, 16, 32, 33, 64, 65, 100, 128])
@pytest.mark.parametrize("scheme", [DPF, DCF, HT])
def test_eval_points_vs_oracle(engine, O, scheme, n, g):
    rng = SplitMix64(1000 * scheme + 10 * n + g)
    mul = {DPF: 2, DCF: 4, HT: 1}[scheme]
    p = O.Prg(mul)
    s = [rng.i4(), rng.i4()]
    alpha, beta = rand_below(rng, n), beta_for(rng)
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, rng.next() & 1, s, alpha, beta)
    else:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
    xs = edge_xs(rng, n, alpha, 300)  # ragged: not a multiple of the wave or block size
    for b in (0, 1):
        want = O.eval_points(p, scheme, n, g, b, s[b], cws, xs if n > 64 else np.array(xs, dtype=np.uint64), ocw=ocw)
        got = engine.eval_points(scheme, n, g, b, T(s[b]), T(cws), xs_tensor(xs, n),
                                 ocw=None if ocw is None else T(ocw)).cpu().numpy()
        assert (got == want).all(), (scheme, n, g, b, np.argwhere(got != want)[:4])


@pytest.mark.parametrize("k", [1, 4, 11, 12, 13, 16])
@pytest.mark.parametrize("scheme,n,g", [(DPF, 32, G.U64), (DPF, 17, G.BYTES), (DCF, 32, G.U64), (DCF, 20, G.U128),
                                        (HT, 64, G.U64), (HT, 18, G.BYTES)])
def test_eval_points_shared_prefix_vs_oracle(engine, O, scheme, n, g, k):
    """Expanding the shared top k levels once must not change a single output bit."""
    rng = SplitMix64(555 + 1000 * scheme + 10 * n + k)
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), beta_for(rng)
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, 1, s, alpha, beta)
    else:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
    xs = edge_xs(rng, n, alpha, 500)
    for b in (0, 1):
        want = O.eval_points(p, scheme, n, g, b, s[b], cws, np.array(xs, dtype=np.uint64), ocw=ocw, nthreads=4)
        got = engine.eval_points_shared(scheme, n, g, b, T(s[b]), T(cws), xs_tensor(xs, n),
                                        ocw=None if ocw is None else T(ocw), prefix_bits=k).cpu().numpy()
        assert (got == want).all(), (scheme, n, g, b, k)


def test_eval_points_shared_prefix_full_size(engine, engine_plain):
    """2^22 points, automatic prefix: identical to the plain walk."""
    n, g = 32, G.U64
    import fss_amd
    d = fss_amd.Dpf(n, "u64", engine=engine)
    gen = torch.Generator().manual_seed(5)
    s0s = torch.randint(-2**31, 2**31, (2, 4), dtype=torch.int32, generator=gen)
    cws = d.gen(s0s, 0x1234567, torch.tensor([9, 8, 7, 6], dtype=torch.int32))
    xs = torch.randint(-2**31, 2**31, (1 << 22,), dtype=torch.int32, generator=gen).cuda()
    a = engine_plain.eval_points(DPF, n, g, 1, s0s[1], cws, xs)
    assert engine_plain.last_kernel()[0] == "dpf_eval_points"
    b = engine.eval_points_shared(DPF, n, g, 1, s0s[1], cws, xs)
    assert engine.last_kernel()[0] == "dpf_eval_points_shared"
    assert bool((a == b).all())
    # FSS_AMD_POINTS_AUTO (the default): the plain entry point picks the same plan for a batch this size ...
    c = engine.eval_points(DPF, n, g, 1, s0s[1], cws, xs)
    assert engine.last_kernel()[0] == "dpf_eval_points_shared"
    assert engine._L.fss_amd_points_prefix_bits(DPF, n, 1 << 22) == 21
    assert bool((a == c).all())
    # ... not for a small one, and not inside a stream capture (its buffer may have to be allocated)
    engine.eval_points(DPF, n, g, 1, s0s[1], cws, xs[: (1 << 18) - 1])
    assert engine.last_kernel()[0] == "dpf_eval_points"
    assert engine._L.fss_amd_points_prefix_bits(DPF, n, (1 << 18) - 1) == 0
    assert engine._L.fss_amd_points_prefix_bits(DPF, n, 1 << 18) == 16 and engine._L.fss_amd_points_prefix_bits(DPF, 12, 1 << 24) == 12
    assert engine._L.fss_amd_points_prefix_bits(HT, 12, 1 << 24) == 11 and engine._L.fss_amd_points_prefix_bits(DCF, 64, 1 << 40) == 26
    side = torch.cuda.Stream()
    out = torch.empty((1 << 22, 4), dtype=torch.int32, device="cuda")
    s1, cw1 = s0s[1].cuda(), cws.cuda()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            engine.eval_points(DPF, n, g, 1, s1, cw1, xs, out=out)
        assert engine.last_kernel()[0] == "dpf_eval_points"
        graph.replay()
    side.synchronize()
    assert bool((a == out).all())


@pytest.mark.parametrize("scheme,n,k", [(DPF, 20, 20), (DPF, 32, 22), (DPF, 64, 13), (DCF, 32, 18), (DCF, 18, 18), (DCF, 18, 17),
                                        (HT, 21, 20), (HT, 21, 19), (HT, 64, 16)])
def test_eval_points_shared_prefix_pipeline(engine, engine_plain, O, scheme, n, k):
    """A batch large enough for the shared-prefix walk proper (past the quad-walk threshold, several chunks per wave):
    every chunk's start nodes are requested half way down the chunk before it -- with zero, one, two and many walked
    levels left (k = n: no level at all; Half-Tree k = n - 1: only the last level).  Ragged size, both parties,
    the whole batch against the plain walk and a strided sample of it against the oracle."""
    g = G.U64
    rng = SplitMix64(9090 + 1000 * scheme + 10 * n + k)
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), beta_for(rng)
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, 1, s, alpha, beta)
    else:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
    N = (1 << 20) + 77
    gen = torch.Generator().manual_seed(n * 100 + k)
    if n <= 32:
        xs = torch.randint(-2**31, 2**31, (N,), dtype=torch.int32, generator=gen)
        if n < 32:
            xs &= (1 << n) - 1
    else:
        xs = torch.randint(-2**63, 2**63 - 1, (N,), dtype=torch.int64, generator=gen)
    bits = 32 if n <= 32 else 64
    xs[5] = alpha - (1 << bits) if alpha >= (1 << (bits - 1)) else alpha  # the key's alpha, as a bit pattern
    xs = xs.cuda()
    oc = None if ocw is None else T(ocw)
    pick = torch.arange(0, N, 511)
    xh = xs[pick.cuda()].cpu().numpy()
    xu = (xh.view(np.uint32) if n <= 32 else xh.view(np.uint64)).astype(np.uint64)
    for b in (0, 1):
        plain = engine_plain.eval_points(scheme, n, g, b, T(s[b]), T(cws), xs, ocw=oc)
        got = engine.eval_points_shared(scheme, n, g, b, T(s[b]), T(cws), xs, ocw=oc, prefix_bits=k)
        assert bool((got == plain).all()), (scheme, n, k, b)
        auto = engine.eval_points(scheme, n, g, b, T(s[b]), T(cws), xs, ocw=oc)  # FSS_AMD_POINTS_AUTO: its own prefix_bits
        assert engine.last_kernel()[0].endswith("_eval_points_shared")
        assert bool((auto == plain).all()), (scheme, n, k, b)
        want = O.eval_points(p, scheme, n, g, b, s[b], cws, xu, ocw=ocw, nthreads=4)
        assert (got[pick.cuda()].cpu().numpy() == want).all(), (scheme, n, k, b)


def test_eval_points_empty(engine):
    out = engine.eval_points(DPF, 16, G.BYTES, 0, torch.zeros(4, dtype=torch.int32),
                             torch.zeros((17, 8), dtype=torch.int32), torch.zeros(0, dtype=torch.int32))
    assert out.shape == (0, 4)


# ---------------------------------------------------------------- independent keys (relayout + eval_keys)
@pytest.mark.parametrize("scheme,n,g", [(DPF, 32, G.U64), (DPF, 64, G.BYTES), (DPF, 5, G.U8), (DCF, 32, G.U64),
                                        (DCF, 64, G.U128), (DCF, 9, G.BYTES), (HT, 64, G.U64), (HT, 32, G.BYTES),
                                        (HT, 1, G.U16), (HT, 2, G.U64), (DPF, 128, G.U128), (DPF, 77, G.BYTES),
                                        (DCF, 128, G.U128), (HT, 100, G.U64)])
def test_eval_keys_vs_oracle(engine, O, scheme, n, g):
    rng = SplitMix64(77 + 1000 * scheme + n)
    nkeys = 333
    mul = {DPF: 2, DCF: 4, HT: 1}[scheme]
    p = O.Prg(mul)
    s0s = np.stack([np.stack([rng.i4(), rng.i4()]) for _ in range(nkeys)])
    if n > 64:
        alphas = [rand_below(rng, n) for _ in range(nkeys)]
        xs = [alphas[i] if i % 3 == 0 else rand_below(rng, n) for i in range(nkeys)]
    else:
        alphas = np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64)
    betas = np.stack([rng.i4() for _ in range(nkeys)])
    cws, ocws = O.gen_keys(p, scheme, n, g, s0s, alphas, betas, pred=O.GT if n == 9 else O.LT, nthreads=4)
    if n <= 64:
        xs = np.array([alphas[i] if i % 3 == 0 else rng.below(n) for i in range(nkeys)], dtype=np.uint64)
    keys = engine.relayout(scheme, n, T(cws))
    # the relayout itself, against a numpy transposition (point_eval_gpu.cuh:53-91)
    assert (keys.cw_s.cpu().numpy() == np.transpose(cws[:, :n, :4], (1, 0, 2))).all()
    if scheme == DPF:
        tr = np.zeros(((n + 63) // 64, nkeys), dtype=np.uint64)
        for i in range(n):
            tr[i // 64] |= (cws[:, i, 4].astype(np.uint64) & np.uint64(1)) << np.uint64(i % 64)
        assert (keys.tr_bits.cpu().numpy().view(np.uint64) == tr.reshape(-1)).all()
        assert (keys.out_cw.cpu().numpy() == cws[:, n, :4]).all()
    elif scheme == DCF:
        assert (keys.cw_v.cpu().numpy() == np.transpose(cws[:, :n, 4:], (1, 0, 2))).all()
        assert (keys.out_cw.cpu().numpy() == cws[:, n, 4:]).all()
    else:
        assert (keys.extra.cpu().numpy() == (cws[:, n - 1, 4] & 1)).all()
    for b in (0, 1):
        want = O.eval_keys(p, scheme, n, g, b, s0s[:, b], cws, xs, ocws=ocws, nthreads=4)
        got = engine.eval_keys(keys, g, b, T(s0s[:, b].copy()), xs_tensor(xs, n),
                               ocws=None if ocws is None else T(ocws)).cpu().numpy()
        assert (got == want).all(), (scheme, n, g, b)


# ---------------------------------------------------------------- key generation on the device
@pytest.mark.parametrize("scheme,n,g,pred", [(DPF, 32, G.U64, 0), (DPF, 64, G.U128, 0), (DPF, 1, G.BYTES, 0),
                                             (DCF, 32, G.U64, 0), (DCF, 16, G.BYTES, 1), (DCF, 64, G.U128, 1),
                                             (DCF, 8, G.U8, 0), (HT, 64, G.U64, 0), (HT, 1, G.BYTES, 0),
                                             (HT, 17, G.U128, 0), (HT, 8, G.U16, 0), (DPF, 128, G.U128, 0),
                                             (DCF, 100, G.BYTES, 1), (HT, 128, G.U128, 0)])
def test_gen_vs_oracle(engine, O, scheme, n, g, pred):
    rng = SplitMix64(5 + 100 * scheme + n)
    nkeys = 130
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    s0s = np.stack([np.stack([rng.i4(clear_lsb=False), rng.i4(clear_lsb=False)]) for _ in range(nkeys)])
    alphas = ([rand_below(rng, n) for _ in range(nkeys)] if n > 64
              else np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64))
    betas = np.stack([rng.i4(clear_lsb=False) for _ in range(nkeys)])
    want, want_ocw = O.gen_keys(p, scheme, n, g, s0s, alphas, betas, pred=pred, nthreads=4)
    got, got_ocw = engine.gen(scheme, n, g, T(s0s), xs_tensor(alphas, n), T(betas), pred=pred)
    assert (got.cpu().numpy() == want).all()
    if scheme == HT:
        assert (got_ocw.cpu().numpy() == want_ocw).all()


@pytest.mark.parametrize("scheme,n,g,pred", [(DPF, 32, G.U64, 0), (DPF, 1, G.BYTES, 0), (DPF, 64, G.U128, 0),
                                             (DPF, 100, G.BYTES, 0), (DCF, 32, G.U64, 1), (DCF, 7, G.U8, 0),
                                             (DCF, 128, G.U128, 0), (HT, 64, G.U64, 0), (HT, 1, G.BYTES, 0),
                                             (HT, 33, G.U32, 0)])
def test_gen_level_major_vs_oracle(engine, O, scheme, n, g, pred):
    """Gen writing level-major directly = the oracle's key-major Gen transposed (numpy, not the
    relayout kernel) = relayout(gen()); and the keys evaluate to the oracle's shares."""
    rng = SplitMix64(77 + 100 * scheme + n)
    nkeys = 193
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    s0s = np.stack([np.stack([rng.i4(clear_lsb=False), rng.i4(clear_lsb=False)]) for _ in range(nkeys)])
    alphas = ([rand_below(rng, n) for _ in range(nkeys)] if n > 64
              else np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64))
    betas = np.stack([rng.i4(clear_lsb=False) for _ in range(nkeys)])
    want, want_ocw = O.gen_keys(p, scheme, n, g, s0s, alphas, betas, pred=pred, nthreads=4)
    keys, ocws = engine.gen_level_major(scheme, n, g, T(s0s), xs_tensor(alphas, n), T(betas), pred=pred)
    assert (keys.cw_s.cpu().numpy() == want[:, :n, 0:4].transpose(1, 0, 2)).all()
    if scheme == DPF:
        tr = keys.tr_bits.cpu().numpy().view(np.uint64).reshape(-1, nkeys)
        for i in range(n):
            assert (((tr[i // 64] >> np.uint64(i % 64)) & np.uint64(1)) == (want[:, i, 4] & 1)).all(), i
        assert (keys.out_cw.cpu().numpy() == want[:, n, 0:4]).all()
    elif scheme == DCF:
        assert (keys.cw_v.cpu().numpy() == want[:, :n, 4:8].transpose(1, 0, 2)).all()
        assert (keys.out_cw.cpu().numpy() == want[:, n, 4:8]).all()
    else:
        assert (keys.extra.cpu().numpy() == (want[:, n - 1, 4] & 1)).all()
        assert (ocws.cpu().numpy() == want_ocw).all()
    assert (engine.key_major(keys).cpu().numpy() == want).all()  # and back to the reference's slot format
    ref = engine.relayout(scheme, n, T(want))
    for name in ("cw_s", "cw_v", "tr_bits", "out_cw", "extra"):
        a, b = getattr(keys, name), getattr(ref, name)
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b), name
    xs = ([rand_below(rng, n) for _ in range(nkeys)] if n > 64
          else np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64))
    for b in (0, 1):
        exp = O.eval_keys(p, scheme, n, g, b, s0s[:, b], want, xs, ocws=want_ocw, nthreads=4)
        got = engine.eval_keys(keys, g, b, T(np.ascontiguousarray(s0s[:, b])), xs_tensor(xs, n), ocws=ocws).cpu().numpy()
        assert (got == exp).all(), (scheme, n, g, b)


# ---------------------------------------------------------------- full domain
@pytest.mark.parametrize("g", [G.BYTES, G.U32, G.U64, G.U128])
@pytest.mark.parametrize("n", [1, 2, 5, 11, 12, 13, 14, 16])
@pytest.mark.parametrize("scheme", [DPF, DCF, HT])
def test_eval_all_vs_oracle(engine, O, scheme, n, g):
    if g in (G.U32, G.U128) and n not in (5, 14):
        pytest.skip("group coverage on two sizes is enough")
    rng = SplitMix64(31337 + 100 * scheme + n)
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), beta_for(rng)
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
        ref = lambda b: O.dpf_eval_all(p, n, g, b, s[b], cws)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, 0, s, alpha, beta)
        ref = lambda b: O.dcf_eval_all(p, n, g, b, s[b], cws)
    else:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
        ref = lambda b: O.ht_eval_all(p, n, g, b, s[b], cws, ocw)
    for b in (0, 1):
        got = engine.eval_all(scheme, n, g, b, T(s[b]), T(cws), ocw=None if ocw is None else T(ocw)).cpu().numpy()
        assert (got == ref(b)).all(), (scheme, n, g, b)


@pytest.mark.parametrize("scheme,n", [(DPF, 18), (DPF, 19), (DPF, 20), (DPF, 21), (DPF, 22), (GROTTO, 19), (GROTTO, 21),
                                      (DCF, 17), (DCF, 19), (DCF, 20), (DCF, 21), (HT, 17), (HT, 18), (HT, 19), (HT, 21)])
def test_eval_all_planner_shapes_vs_oracle(engine, O, scheme, n):
    """Domains around the planner's thresholds (frontier depth 15 / 18, leaf-stage depth sized by the domain:
    fss_abi_evalall.hip top_max / leaf_extra), whole and as the second quarter, bit for bit."""
    rng = SplitMix64(5150 + 100 * scheme + n)
    g = G.BYTES if scheme == GROTTO else G.U64
    p = O.Prg({DPF: 2, DCF: 4, HT: 1, GROTTO: 2}[scheme])
    s = [rng.i4(), rng.i4()]
    alpha, beta, ocw = rng.below(n), beta_for(rng), None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
        want = O.dpf_eval_all(p, n, g, 1, s[1], cws)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, 0, s, alpha, beta)
        want = O.dcf_eval_all(p, n, g, 1, s[1], cws)
    elif scheme == HT:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
        want = O.ht_eval_all(p, n, g, 1, s[1], cws, ocw)
    else:
        cws = O.grotto_gen(p, n, s, alpha)
        want = O.grotto_eval_all(p, n, 1, s[1], cws)
    dev = lambda **kw: engine.eval_all(scheme, n, g, 1, T(s[1]), T(cws), ocw=None if ocw is None else T(ocw), **kw).cpu().numpy()
    assert (dev() == want).all()
    if scheme != GROTTO:
        q = 1 << (n - 2)
        assert (dev(first_leaf=q, nleaves=q) == want[q:2 * q]).all()


@pytest.mark.parametrize("scheme", [DPF, DCF, HT])
def test_eval_all_subblocks(engine, O, scheme):
    """Domain sharding: aligned power-of-two blocks reproduce the slices of the whole."""
    n, g = 15, G.U64
    rng = SplitMix64(4242 + scheme)
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    s = [rng.i4(), rng.i4()]
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, rng.below(n), beta_for(rng))
        whole = O.dpf_eval_all(p, n, g, 1, s[1], cws)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, 1, s, rng.below(n), beta_for(rng))
        whole = O.dcf_eval_all(p, n, g, 1, s[1], cws)
    else:
        cws, ocw = O.ht_gen(p, n, g, s, rng.below(n), beta_for(rng))
        whole = O.ht_eval_all(p, n, g, 1, s[1], cws, ocw)
    for nl, first in [(1 << 12, 3 << 12), (1 << 14, 1 << 14), (1 << 5, 17 << 5), (2, 2 * 777),
                      (1 if scheme != HT else 2, 4 * 100)]:
        got = engine.eval_all(scheme, n, g, 1, T(s[1]), T(cws), ocw=None if ocw is None else T(ocw),
                              first_leaf=first, nleaves=nl).cpu().numpy()
        assert (got == whole[first:first + nl]).all(), (scheme, nl, first)


def test_eval_all_deep_prefix_block(engine, O):
    """A small block far into a 40-bit domain: 28 prefix levels walked, 64-bit leaf indices."""
    n, g = 40, G.U64
    rng = SplitMix64(77777)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    nl = 1 << 12
    first = (0xABCDE12 << 12)
    alpha = first + 1234  # inside the block
    beta = beta_for(rng)
    cws = O.dpf_gen(p, n, g, s, alpha, beta)
    xs = np.arange(first, first + nl, dtype=np.uint64)
    for b in (0, 1):
        want = O.eval_points(p, DPF, n, g, b, s[b], cws, xs, nthreads=4)
        got = engine.eval_all(DPF, n, g, b, T(s[b]), T(cws), first_leaf=first, nleaves=nl).cpu().numpy()
        assert (got == want).all()


def test_eval_points_graph_capture(engine, O):
    """The eval entry points only enqueue work: they can be captured in a HIP graph and replayed."""
    n, g = 32, G.U64
    rng = SplitMix64(31)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    cws = O.dpf_gen(p, n, g, s, 12345, beta_for(rng))
    xs = xs_tensor([rng.below(n) for _ in range(5000)], n).cuda()
    seed, cw = T(s[0]).cuda(), T(cws).cuda()
    ys = torch.zeros((5000, 4), dtype=torch.int32, device="cuda")
    eager = engine.eval_points(DPF, n, g, 0, seed, cw, xs).clone()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        engine.eval_points(DPF, n, g, 0, seed, cw, xs, out=ys)  # warm-up on the side stream
        st.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            engine.eval_points(DPF, n, g, 0, seed, cw, xs, out=ys)
    ys.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert bool((ys == eager).all())


def test_large_batch_graph_capture_and_streams(engine, O):
    """Large walks use the launch-wide tail queue (a counter pair per stream).  A captured launch must not depend on a
    stream's pair (it may be replayed anywhere), a first launch may sit inside a capture (nothing is allocated), and
    launches on several streams at once stay independent: all of them bit-identical with the eager result."""
    n, g, N = 32, G.U64, 1 << 22
    rng = SplitMix64(77)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    cws = O.dpf_gen(p, n, g, s, 0x0BADF00D, beta_for(rng))
    gen = torch.Generator(device="cuda").manual_seed(5)
    xs = torch.randint(-2**31, 2**31, (N,), dtype=torch.int32, device="cuda", generator=gen)
    seed, cw = T(s[0]).cuda(), T(cws).cuda()
    eager = engine.eval_points(DPF, n, g, 0, seed, cw, xs).clone()
    idx = np.arange(0, N, 8191)
    want = O.eval_points(p, DPF, n, g, 0, s[0], cws, xs.cpu().numpy().view(np.uint32)[idx].astype(np.uint64), nthreads=4)
    assert (eager.cpu().numpy()[idx] == want).all()
    # capture on a fresh stream with no warm-up launch on it, replay twice
    ys = torch.zeros((N, 4), dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st):
            engine.eval_points(DPF, n, g, 0, seed, cw, xs, out=ys)
    for _ in range(2):
        ys.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(ys, eager)
    # four streams at once, several launches each
    streams = [torch.cuda.Stream() for _ in range(4)]
    outs = [torch.zeros((N, 4), dtype=torch.int32, device="cuda") for _ in streams]
    torch.cuda.synchronize()
    for _ in range(3):
        for st_i, o in zip(streams, outs):
            with torch.cuda.stream(st_i):
                engine.eval_points(DPF, n, g, 0, seed, cw, xs, out=o)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, eager)


def test_grotto_eval_all_sharded_with_carry(engine, O):
    """Domain-sharded Grotto EvalAll: per-block running XOR plus the parity carry of fss_amd.shard."""
    from fss_amd import shard
    n, world = 12, 4
    rng = SplitMix64(4711)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    cws = O.grotto_gen(p, n, s, 1500)
    whole = O.grotto_eval_all(p, n, 1, s[1], cws)
    blocks = []
    for r in range(world):
        first, nl = shard.domain_block(n, r, world)
        blocks.append(engine.eval_all(GROTTO, n, G.BYTES, 1, T(s[1]), T(cws), first_leaf=first, nleaves=nl).cpu().numpy())
    carries = shard.grotto_carries([int(b[-1]) for b in blocks])
    got = np.concatenate([b ^ c for b, c in zip(blocks, carries)])
    assert (got == whole).all()


# ---------------------------------------------------------------- Grotto
@pytest.mark.parametrize("n", [1, 3, 10, 13, 16])
def test_grotto_eval_all_vs_oracle(engine, O, n):
    rng = SplitMix64(900 + n)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    for alpha in {0, (1 << n) - 1, rng.below(n)}:
        cws = O.grotto_gen(p, n, s, alpha)
        got = []
        for b in (0, 1):
            y = engine.eval_all(GROTTO, n, G.BYTES, b, T(s[b]), T(cws)).cpu().numpy()
            assert (y == O.grotto_eval_all(p, n, b, s[b], cws)).all(), (n, alpha, b)
            got.append(y)
        assert ((got[0] ^ got[1]) == (np.arange(1 << n) >= alpha)).all()


@pytest.mark.parametrize("n", [1, 2, 9, 14])
def test_grotto_preprocess_and_table_eval_vs_oracle(engine, O, n):
    """GrottoDcf::Preprocess + Eval (grotto_dcf.cuh:94-135): the table and every table lookup, per party."""
    rng = SplitMix64(902 + n)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    for alpha in {0, (1 << n) - 1, rng.below(n)}:
        cws = O.grotto_gen(p, n, s, alpha)
        xs = np.arange(1 << n, dtype=np.uint64) if n <= 9 else np.array(
            [0, (1 << n) - 1, alpha] + [rng.below(n) for _ in range(500)], dtype=np.uint64)
        sh = []
        for b in (0, 1):
            want_pt = O.grotto_preprocess(p, n, b, s[b], cws)
            pt = engine.grotto_preprocess(n, b, T(s[b]), T(cws))
            assert (pt.cpu().numpy() == want_pt).all(), (n, alpha, b)
            got = engine.grotto_eval_table(n, pt, xs_tensor(xs, n)).cpu().numpy()
            want = np.array([O.grotto_eval(n, want_pt, int(x)) for x in xs], dtype=np.uint8)
            assert (got == want).all()
            sh.append(got)
        assert ((sh[0] ^ sh[1]) == (xs >= alpha)).all()


@pytest.mark.parametrize("n,nkeys", [(1, 200), (16, 200), (32, 200), (64, 200), (64, 1), (64, 63), (64, 129), (37, 4099)])
def test_grotto_eval_keys_vs_oracle_walk(engine, O, n, nkeys):
    rng = SplitMix64(901 + n + 7 * nkeys)
    p = O.Prg(2)
    s0s = np.stack([np.stack([rng.i4(), rng.i4()]) for _ in range(nkeys)])
    alphas = np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64)
    cws, _ = O.gen_keys(p, GROTTO, n, G.BYTES, s0s, alphas, np.zeros((nkeys, 4), np.int32), nthreads=4)
    top = (1 << n) - 1
    xs = np.array([[alphas[i], top, 0, rng.below(n)][i % 4] for i in range(nkeys)], dtype=np.uint64)
    keys = engine.relayout(GROTTO, n, T(cws))
    sh = []
    for b in (0, 1):
        want = O.eval_keys(p, GROTTO, n, G.BYTES, b, s0s[:, b], cws, xs, nthreads=4)[:, 0].astype(np.uint8)
        got = engine.eval_keys(keys, G.BYTES, b, T(s0s[:, b].copy()), xs_tensor(xs, n)).cpu().numpy()
        assert (got == want).all()
        sh.append(got)
    assert ((sh[0] ^ sh[1]) == (alphas <= xs)).all()


# ---------------------------------------------------------------- BASELINE sizes: size-independent properties
@pytest.mark.parametrize("scheme,n", [(DPF, 32), (DCF, 32)])
def test_full_size_points_reconstruct(engine, O, scheme, n):
    """configs[1]/[2]: one key, 2^20 random points, both parties on the GPU; the sum of
    shares is beta exactly where the predicate holds; a strided sample equals the oracle."""
    g = G.U64
    rng = SplitMix64(0x5EED0000 + 2 + scheme)
    p = O.Prg(2 if scheme == DPF else 4)
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), rng.i4()
    cws = O.dpf_gen(p, n, g, s, alpha, beta) if scheme == DPF else O.dcf_gen(p, n, g, 0, s, alpha, beta)
    N = 1 << 20
    gen = torch.Generator().manual_seed(1234)
    xs = torch.randint(-2**31, 2**31, (N,), dtype=torch.int32, generator=gen)
    xs[:3] = torch.tensor([alpha - (1 << 32) if alpha >= 2**31 else alpha, 0, -1], dtype=torch.int32)
    xd = xs.cuda()
    y = [engine.eval_points(scheme, n, g, b, T(s[b]), T(cws), xd) for b in (0, 1)]
    v = [(t[:, 0].to(torch.int64) & 0xFFFFFFFF) | (t[:, 1].to(torch.int64) << 32) for t in y]
    total = v[0] + v[1]  # wraps mod 2^64
    xu = xd.to(torch.int64) & 0xFFFFFFFF
    hit = (xu == alpha) if scheme == DPF else (xu < alpha)
    bv = (int(beta[0]) & 0xFFFFFFFF) | ((int(beta[1]) & 0xFFFFFFFF) << 32)
    bv = bv - (1 << 64) if bv >= (1 << 63) else bv
    want = torch.where(hit, torch.tensor(bv, dtype=torch.int64, device=xd.device), torch.zeros_like(total))
    assert bool((total == want).all())
    assert bool((y[0][:, 2:] == 0).all())
    idx = np.arange(0, N, 4099)
    xs_np = xs.numpy().view(np.uint32).astype(np.uint64)[idx]
    want0 = O.eval_points(p, scheme, n, g, 0, s[0], cws, xs_np, nthreads=4)
    assert (y[0].cpu().numpy()[idx] == want0).all()


def test_full_size_eval_all_properties(engine, O):
    """configs[3] shape at a size that fits the test budget (2^24 leaves): exactly one
    non-zero reconstructed leaf (= beta at alpha), and random leaves equal Eval."""
    n, g = 24, G.U64
    rng = SplitMix64(0x5EED0004)
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), rng.i4()
    cws = O.dpf_gen(p, n, g, s, alpha, beta)
    y = [engine.eval_all(DPF, n, g, b, T(s[b]), T(cws)) for b in (0, 1)]
    v = [(t[:, 0].to(torch.int64) & 0xFFFFFFFF) | (t[:, 1].to(torch.int64) << 32) for t in y]
    total = v[0] + v[1]
    nz = torch.nonzero(total).flatten()
    assert nz.numel() == 1 and int(nz[0]) == alpha
    bv = (int(beta[0]) & 0xFFFFFFFF) | ((int(beta[1]) & 0xFFFFFFFF) << 32)
    assert (int(total[alpha]) & 0xFFFFFFFFFFFFFFFF) == bv
    y0 = y[0].cpu().numpy()
    for x in [0, (1 << n) - 1, alpha] + [rng.below(n) for _ in range(50)]:
        assert (y0[x] == O.dpf_eval(p, n, g, 0, s[0], cws, x)).all()


def test_big_dcf_eval_all_properties(engine, O):
    """DCF full domains from 2^25 leaves on take two levels per intermediate stage (fss_abi_evalall.hip: mid_extra) and, like
    every DCF tree on the AES tables, expand a node's children one at a time with 64-byte leaf stores (fss_evalall.hpp:
    dcf_child): n = 26 whole and its second 2^25-leaf half as a block -- the two parties' leaves add up to beta below alpha
    and to zero from alpha on, over the WHOLE domain; leaves at and around alpha, at the ends and at random places equal
    the oracle's Eval; the block equals the same leaves of the whole."""
    n, g = 26, G.U64
    rng = SplitMix64(0x5EED0D26)
    p = O.Prg(4)
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), rng.i4()
    cws = O.dcf_gen(p, n, g, 0, s, alpha, beta)
    y = [engine.eval_all(DCF, n, g, b, T(s[b]), T(cws)) for b in (0, 1)]
    v = [(t[:, 0].to(torch.int64) & 0xFFFFFFFF) | (t[:, 1].to(torch.int64) << 32) for t in y]
    total = v[0] + v[1]
    bv = (int(beta[0]) & 0xFFFFFFFF) | ((int(beta[1]) & 0xFFFFFFFF) << 32)
    bv = bv - (1 << 64) if bv >= (1 << 63) else bv
    assert bool((total[:alpha] == bv).all()) and bool((total[alpha:] == 0).all())
    y0 = y[0].cpu().numpy()
    xs = sorted({0, 1, (1 << n) - 1, max(alpha - 1, 0), alpha, min(alpha + 1, (1 << n) - 1)} | {rng.below(n) for _ in range(40)})
    want = O.eval_points(p, DCF, n, g, 0, s[0], cws, np.array(xs, dtype=np.uint64), nthreads=4)
    assert (y0[xs] == want).all()
    half = 1 << (n - 1)
    blk = engine.eval_all(DCF, n, g, 0, T(s[0]), T(cws), first_leaf=half, nleaves=half)
    assert torch.equal(blk, y[0][half:])


def test_full_size_half_tree_keys_reconstruct(engine, O):
    """configs[4] shape (Half-Tree n=64, independent keys) at 2^18 keys generated on the
    device: shares reconstruct beta at alpha and 0 elsewhere; a sample equals the oracle."""
    import fss_amd
    n, g, nkeys = 64, G.U64, 1 << 18
    gen = torch.Generator().manual_seed(99)
    s0s = torch.randint(-2**31, 2**31, (nkeys, 2, 4), dtype=torch.int32, generator=gen)
    alphas = torch.randint(-2**63, 2**63 - 1, (nkeys,), dtype=torch.int64, generator=gen)
    betas = torch.randint(-2**31, 2**31, (nkeys, 4), dtype=torch.int32, generator=gen)
    h = fss_amd.HalfTreeDpf(n, "u64", engine=engine)
    cws, ocws = h.gen_batch(s0s, alphas, betas)
    keys = h.relayout(cws)
    xs = alphas.clone()
    xs[1::2] = torch.randint(-2**63, 2**63 - 1, (nkeys // 2,), dtype=torch.int64, generator=gen)
    y = [h.eval_keys(b, s0s[:, b].contiguous(), keys, ocws, xs) for b in (0, 1)]
    v = [(t[:, 0].to(torch.int64) & 0xFFFFFFFF) | (t[:, 1].to(torch.int64) << 32) for t in y]
    total = (v[0] + v[1]).cpu()
    bl = (betas[:, 0].to(torch.int64) & 0xFFFFFFFF) | (betas[:, 1].to(torch.int64) << 32)
    want = torch.where(xs == alphas, bl, torch.zeros_like(bl))
    assert bool((total == want).all())
    p = O.Prg(1)
    idx = np.arange(0, nkeys, 1031)
    want0 = O.eval_keys(p, HT, n, g, 0, s0s[:, 0].numpy()[idx], cws.cpu().numpy()[idx],
                        xs.numpy().view(np.uint64)[idx], ocws=ocws.cpu().numpy()[idx], nthreads=4)
    assert (y[0].cpu().numpy()[idx] == want0).all()


def test_full_size_grotto_keys_reconstruct(engine, O):
    """configs[4] shape, Grotto half (n=64, independent keys, path walk) at 2^20 keys generated on the device: the
    two shares XOR to 1[alpha <= x] (unsigned) for every key; a sample equals the oracle's walk bit for bit."""
    n, nkeys = 64, 1 << 20
    gen = torch.Generator().manual_seed(77)
    s0s = torch.randint(-2**31, 2**31, (nkeys, 2, 4), dtype=torch.int32, generator=gen)
    alphas = torch.randint(-2**63, 2**63 - 1, (nkeys,), dtype=torch.int64, generator=gen)
    xs = torch.randint(-2**63, 2**63 - 1, (nkeys,), dtype=torch.int64, generator=gen)
    xs[0::5] = alphas[0::5]
    xs[1::5] = alphas[1::5] - 1  # wraps at alpha = 0, like the unsigned domain
    xs[2] = -1                   # x = 2^64 - 1: the whole-domain case
    keys, _ = engine.gen_level_major(GROTTO, n, G.BYTES, s0s.cuda(), alphas.cuda(),
                                      torch.zeros((nkeys, 4), dtype=torch.int32, device="cuda"))
    sh = [engine.eval_keys(keys, G.BYTES, b, s0s[:, b].contiguous().cuda(), xs.cuda()).cpu().numpy() for b in (0, 1)]
    au, xu = alphas.numpy().view(np.uint64), xs.numpy().view(np.uint64)
    assert ((sh[0] ^ sh[1]) == (au <= xu)).all()
    idx = np.arange(0, nkeys, 4093)
    cws = engine.key_major(keys).cpu().numpy()[idx]
    want0 = O.eval_keys(O.Prg(2), GROTTO, n, G.BYTES, 0, s0s[:, 0].numpy()[idx], cws, xu[idx], nthreads=4)[:, 0].astype(np.uint8)
    assert (sh[0][idx] == want0).all()


def test_cxx_sample_self_check():
    """samples/dpf_dcf_gpu.cpp: device Gen -> relayout -> EvalPointGpu for both parties through
    include/fss_amd.hpp, reconstructing beta (the reference's samples/dpf_dcf_gpu.cu:155-162,227-234)."""
    import subprocess
    from fss_amd import _build
    exe = _build.build_sample()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout and "OK" in r.stdout


@pytest.mark.parametrize("scheme,n,dtype", [(DPF, 8, "uint8"), (DCF, 5, "uint8"), (HT, 8, "uint8"), (DPF, 16, "int16"),
                                            (DCF, 16, "int16"), (HT, 11, "int16")])
def test_narrow_input_types_vs_oracle(engine, O, scheme, n, dtype):
    """In = uint8_t / uint16_t (the reference's C1 sample is Dpf<8, Bytes, Aes128Mmo<2>, uint8_t>): x arrays of
    1- and 2-byte elements through the shared-key, shared-prefix, independent-keys and Gen entry points."""
    rng = SplitMix64(555 + 10 * scheme + n)
    g = G.U64
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(n), beta_for(rng)
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, 0, s, alpha, beta)
    else:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
    xs = np.array([alpha] + [rng.below(n) for _ in range(300)], dtype=np.uint64)
    tdt = torch.uint8 if dtype == "uint8" else torch.int16
    xt = torch.from_numpy(xs.astype(np.uint8) if dtype == "uint8" else xs.astype(np.uint16).view(np.int16)).cuda()
    assert xt.dtype == tdt
    m = len(xs)
    s0s = np.broadcast_to(np.stack(s), (m, 2, 4)).copy()
    at = torch.full((m,), int(alpha), dtype=torch.int64).to(tdt).cuda() if dtype == "uint8" else \
        torch.from_numpy(np.full(m, alpha, dtype=np.uint16).view(np.int16)).cuda()
    keys, ocws = engine.gen_level_major(scheme, n, g, T(s0s), at, T(np.broadcast_to(beta, (m, 4)).copy()))
    for b in (0, 1):
        want = O.eval_points(p, scheme, n, g, b, s[b], cws, xs, ocw=ocw)
        got = engine.eval_points(scheme, n, g, b, T(s[b]), T(cws), xt, ocw=None if ocw is None else T(ocw)).cpu().numpy()
        assert (got == want).all()
        k = min(4, n - (scheme == HT))
        if k >= 1:
            got = engine.eval_points_shared(scheme, n, g, b, T(s[b]), T(cws), xt, ocw=None if ocw is None else T(ocw),
                                            prefix_bits=k).cpu().numpy()
            assert (got == want).all()
        got = engine.eval_keys(keys, g, b, T(np.ascontiguousarray(s0s[:, b])), xt, ocws=ocws).cpu().numpy()
        assert (got == want).all()


@pytest.mark.parametrize("scheme,n,g,mod", [(DPF, 16, G.U32, 1000003), (DCF, 16, G.U32, 1000003), (HT, 16, G.U32, 97),
                                            (DPF, 33, G.U64, (1 << 61) - 1), (DCF, 20, G.U64, (1 << 63) + 29),
                                            (HT, 9, G.U8, 251), (DCF, 12, G.U16, 65521), (DPF, 8, G.U16, 1 << 16)])
def test_modular_uint_groups_vs_oracle(O, scheme, n, g, mod):
    """fss::group::Uint<T, mod> with mod != 0 (uint.cuh:11-31): shared key, independent keys through device Gen,
    EvalAll; and the shares reconstruct beta mod `mod`."""
    import fss_amd
    eng = fss_amd.Engine(0)
    eng.set_uint_modulus(mod)
    O.set_uint_modulus(mod)
    try:
        rng = SplitMix64(9000 + 7 * n + scheme)
        p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
        s = [rng.i4(), rng.i4()]
        alpha, beta = rng.below(n), rng.i4()
        ocw = None
        if scheme == DPF:
            cws = O.dpf_gen(p, n, g, s, alpha, beta)
        elif scheme == DCF:
            cws = O.dcf_gen(p, n, g, 0, s, alpha, beta)
        else:
            cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
        xs = np.array([alpha, max(alpha - 1, 0)] + [rng.below(n) for _ in range(200)], dtype=np.uint64)
        m = len(xs)
        s0s = np.broadcast_to(np.stack(s), (m, 2, 4)).copy()
        got_cws, got_ocws = eng.gen(scheme, n, g, T(s0s[:3]), xs_tensor([alpha] * 3, n), T(np.broadcast_to(beta, (3, 4)).copy()))
        assert (got_cws.cpu().numpy()[1] == cws).all()
        if scheme == HT:
            assert (got_ocws.cpu().numpy()[2] == ocw).all()
        keys, ocws = eng.gen_level_major(scheme, n, g, T(s0s), xs_tensor([alpha] * m, n), T(np.broadcast_to(beta, (m, 4)).copy()))
        ys = []
        for b in (0, 1):
            want = O.eval_points(p, scheme, n, g, b, s[b], cws, xs, ocw=ocw)
            got = eng.eval_points(scheme, n, g, b, T(s[b]), T(cws), xs_tensor(xs, n), ocw=None if ocw is None else T(ocw)).cpu().numpy()
            assert (got == want).all()
            got = eng.eval_keys(keys, g, b, T(np.ascontiguousarray(s0s[:, b])), xs_tensor(xs, n), ocws=ocws).cpu().numpy()
            assert (got == want).all()
            if n <= 16:
                ref = {DPF: O.dpf_eval_all, DCF: O.dcf_eval_all}.get(scheme)
                full = ref(p, n, g, b, s[b], cws) if ref else O.ht_eval_all(p, n, g, b, s[b], cws, ocw)
                got_all = eng.eval_all(scheme, n, g, b, T(s[b]), T(cws), ocw=None if ocw is None else T(ocw)).cpu().numpy()
                assert (got_all == full).all()
            ys.append(want)
        val = lambda y: (int(np.uint32(y[0])) | (int(np.uint32(y[1])) << 32))
        bits = {G.U8: 8, G.U16: 16, G.U32: 32, G.U64: 64}[g]
        bval = (val(beta) & ((1 << bits) - 1)) % mod
        for i, x in enumerate(xs):
            tot = (val(ys[0][i]) + val(ys[1][i])) % mod
            hit = (int(x) == alpha) if scheme != DCF else (int(x) < alpha)
            assert tot == (bval if hit else 0), (i, int(x), alpha)
        # a modulus that does not fit the group's T is refused
        if g == G.U8:
            eng.set_uint_modulus(300)
            with pytest.raises(fss_amd._lib.FssAmdError):
                eng.eval_points(scheme, n, g, 0, T(s[0]), T(cws), xs_tensor(xs, n), ocw=None if ocw is None else T(ocw))
    finally:
        O.set_uint_modulus(0)


@pytest.mark.parametrize("scheme,n,nkeys,g", [(DPF, 1, 5, G.U64), (DPF, 5, 37, G.BYTES), (DPF, 12, 9, G.U64), (DPF, 14, 3, G.U128),
                                              (DPF, 9, 300, G.U32), (DCF, 2, 7, G.U64), (DCF, 10, 20, G.BYTES), (DCF, 13, 3, G.U64),
                                              (HT, 1, 4, G.U64), (HT, 6, 33, G.U64), (HT, 12, 5, G.BYTES), (DPF, 16, 1, G.U64),
                                              (HT, 3, 1000, G.U64)])
def test_eval_all_batch_vs_oracle(engine, O, scheme, n, nkeys, g):
    """fss::gpu::{Dpf,HalfTreeDpf}EvalAllGpuBatch (eval_all_gpu.cuh:511-535; DCF added): the full domain of every
    key of a key-major batch in one call, ys[k, x] = Eval(party, seeds[k], key k, x)."""
    rng = SplitMix64(31000 + 100 * scheme + n)
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    s0s = np.stack([np.stack([rng.i4(), rng.i4()]) for _ in range(nkeys)])
    alphas = np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64)
    betas = np.stack([rng.i4() for _ in range(nkeys)])
    cws, ocws = O.gen_keys(p, scheme, n, g, s0s, alphas, betas, pred=O.LT, nthreads=4)
    ref = {DPF: lambda b, k: O.dpf_eval_all(p, n, g, b, s0s[k, b], cws[k]),
           DCF: lambda b, k: O.dcf_eval_all(p, n, g, b, s0s[k, b], cws[k]),
           HT: lambda b, k: O.ht_eval_all(p, n, g, b, s0s[k, b], cws[k], ocws[k])}[scheme]
    check = range(nkeys) if nkeys <= 40 else [0, 1, nkeys // 2, nkeys - 2, nkeys - 1]
    for b in (0, 1):
        got = engine.eval_all_batch(scheme, n, g, b, T(np.ascontiguousarray(s0s[:, b])), T(cws),
                                    ocws=None if ocws is None else T(ocws)).cpu().numpy()
        assert got.shape == (nkeys, 1 << n, 4)
        for k in check:
            assert (got[k] == ref(b, k)).all(), (scheme, n, g, b, k)
    # one key per call gives the same rows
    one = engine.eval_all(scheme, n, g, 1, T(s0s[nkeys - 1, 1].copy()), T(cws[nkeys - 1]),
                          ocw=None if ocws is None else T(ocws[nkeys - 1])).cpu().numpy()
    assert (got[nkeys - 1] == one).all()


@pytest.mark.parametrize("scheme,n", [(DPF, 1), (DPF, 8), (DPF, 17), (DPF, 20), (HT, 1), (HT, 12), (HT, 21), (DCF, 16),
                                      (GROTTO, 10), (GROTTO, 21)])
def test_eval_all_is_one_launch(engine, O, scheme, n):
    """With the AES tables a single-key EvalAll of every scheme and size is ONE kernel launch (evalall_fused_kernel:
    walks, stages and leaves of a workgroup's share in one go), and so are the batched DPF / Half-Tree forms -- the
    parity tests above therefore exercise that kernel, not the chain of launches it replaced (FSS_AMD_FUSED=0)."""
    if os.environ.get("FSS_AMD_TUNE") == "1" and (os.environ.get("FSS_AMD_FUSED") == "0" or os.environ.get("FSS_AMD_FUSED_GRID")):
        pytest.skip("the chain of launches was asked for (or a workgroup count the workspace was not sized for)")
    rng = SplitMix64(777 + 100 * scheme + n)
    g = G.BYTES if scheme == GROTTO else G.U64
    p = O.Prg({DPF: 2, DCF: 4, HT: 1, GROTTO: 2}[scheme])
    s = [rng.i4(), rng.i4()]
    alpha, beta, ocw = rng.below(n), beta_for(rng), None
    if scheme == DPF:
        cws = O.dpf_gen(p, n, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, n, g, 0, s, alpha, beta)
    elif scheme == HT:
        cws, ocw = O.ht_gen(p, n, g, s, alpha, beta)
    else:
        cws = O.grotto_gen(p, n, s, alpha)
    # (Grotto: the tree is the one launch; the running-XOR scan kernels behind it do not record themselves)
    engine.eval_all(scheme, n, g, 0, T(s[0]), T(cws), ocw=None if ocw is None else T(ocw))
    assert engine.last_kernel()[0] == "evalall_fused", engine.last_kernel()
    if scheme in (DPF, HT):
        nk = 5
        s0s = torch.stack([T(s[0])] * nk)
        cwb = torch.stack([T(cws)] * nk)
        ocb = None if ocw is None else torch.stack([T(ocw)] * nk)
        engine.eval_all_batch(scheme, n, g, 0, s0s, cwb, ocws=ocb)
        assert engine.last_kernel()[0] == "evalall_fused", engine.last_kernel()


@pytest.mark.parametrize("grid", [77, 304])
def test_eval_all_one_launch_with_uneven_workgroup_shares(grid):
    """evalall_fused_kernel splits the frontier walks over the workgroups as [w * b / G, w * (b + 1) / G): with a
    workgroup count that does not divide 2^16 (FSS_AMD_FUSED_GRID, read once per process, hence the child process) the
    shares are ragged and the private workspace slices are sized by the largest one.  Same oracle comparisons as
    test_eval_all_planner_shapes_vs_oracle (DPF n = 20 .. 22, Grotto n = 21, Half-Tree n = 21 take the one-launch form)."""
    import subprocess
    import sys
    env = dict(os.environ, FSS_AMD_TUNE="1", FSS_AMD_FUSED_GRID=str(grid), FSS_AMD_FUSED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_eval_all_planner_shapes_vs_oracle or test_grotto_eval_all_vs_oracle"],
                       capture_output=True, text=True, timeout=900, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_cxx_dropin_boundary():
    """tests/cxx/dropin_test.cpp: caller code in the reference's own call syntax (samples/dpf_dcf_cpu.cu flow with
    CreateCtxs / Gen / Eval / EvalAll on host arrays; samples/dpf_dcf_gpu.cu:58; src/bench_gpu.cu:578,618,625) compiled
    against include/fss_amd.hpp, checked against the reference's known answers and by reconstruction."""
    import subprocess
    from fss_amd import _build
    exe = _build.build_dropin_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK") and "FAILED" not in r.stdout


def test_cxx_dropin_device_side():
    """tests/cxx/dropin_device_test.cpp: Gen / Eval called per thread from inside the caller's own kernels (the flow of
    samples/dpf_dcf_gpu.cu:39-80) with ChaCha and with AES-128-MMO, against the reference's known answers and bit for
    bit against the engine's batched forms; caller-defined Prgable / Groupable types in a kernel and through the host
    signatures."""
    import subprocess
    from fss_amd import _build
    exe = _build.build_dropin_device_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK") and "FAILED" not in r.stdout


def test_cxx_reference_test_list():
    """tests/cxx/reference_cases_test.cpp: the reference's own gtest list (src/{group,dpf,dcf,half_tree_dpf,grotto_dcf,vdpf}_test.cu:
    91 cases) under its suite and case names, written against the reference's include lines and host signatures; the schemes'
    host forms stage onto the GPU.  Seeds are random per run as in the reference (the seed is printed); a second run pins one."""
    import subprocess
    from fss_amd import _build
    exe = _build.build_reference_cases_test()
    names = subprocess.run([exe, "--list"], capture_output=True, text=True, timeout=60).stdout.split()
    assert len(names) == 91 and "GrottoDcfChaChaTest.EdgeAlphaMax" in names and "GroupAxiomTest/10.Inverses" in names
    for extra in ([], ["--seed", "20261003"]):
        r = subprocess.run([exe] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
        assert r.stdout.strip().endswith("OK") and "FAILED" not in r.stdout and "91 cases, 0 failed" in r.stdout


def test_cxx_multi_gpu_device_set():
    """tests/cxx/multi_gpu_test.cpp: fss::gpu::DeviceSet (include/fss_amd_multi.hpp) with eight members sharing device 0
    -- the headline batch, configs[3] in eight blocks, configs[4]'s schemes on independent keys, Grotto's carries -- gives
    the bytes of one launch; then samples/multi_gpu.cpp over every visible device."""
    import subprocess
    from fss_amd import _build
    exe = _build.build_multi_gpu_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK") and "FAILED" not in r.stdout
    r = subprocess.run([_build.build_multi_gpu_sample(), "22"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("n,nkeys", [(1, 70), (20, 1000), (32, 5000), (31, 129)])
def test_u32_extra_entry_points_vs_u64(engine, O, n, nkeys):
    """fss_amd_dpf_relayout_u32 / fss_amd_dpf_eval_keys_u32: the reference's uint32_t `extra` packing of the tr bits
    (point_eval_gpu.cuh:57-62) gives the same level-major words, the low 32 tr bits, and the same shares as the
    64-bit layout; in_bits > 32 is refused."""
    import ctypes as C
    from fss_amd import _lib
    L, h = engine._L, engine._h
    rng = SplitMix64(4400 + n)
    p = O.Prg(2)
    s0s = np.stack([np.stack([rng.i4(), rng.i4()]) for _ in range(nkeys)])
    alphas = np.array([rng.below(n) for _ in range(nkeys)], dtype=np.uint64)
    betas = np.stack([rng.i4() for _ in range(nkeys)])
    cws, _ = O.gen_keys(p, DPF, n, G.U64, s0s, alphas, betas, nthreads=4)
    xs = np.array([alphas[i] if i % 2 else rng.below(n) for i in range(nkeys)], dtype=np.uint64)
    cd = T(cws).cuda()
    keys = engine.relayout(DPF, n, cd)
    cw_s = torch.empty((n, nkeys, 4), dtype=torch.int32, device="cuda")
    extra = torch.empty((nkeys,), dtype=torch.int32, device="cuda")
    out_cw = torch.empty((nkeys, 4), dtype=torch.int32, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(L.fss_amd_dpf_relayout_u32(h, n, P(cd), nkeys, P(cw_s), P(extra), P(out_cw), None), "relayout_u32")
    assert torch.equal(cw_s, keys.cw_s) and torch.equal(out_cw, keys.out_cw)
    assert torch.equal(extra.to(torch.int64) & 0xFFFFFFFF, keys.tr_bits & 0xFFFFFFFF)
    xt = xs_tensor(xs, n).cuda()
    for b in (0, 1):
        seeds = T(np.ascontiguousarray(s0s[:, b])).cuda()
        ys = torch.empty((nkeys, 4), dtype=torch.int32, device="cuda")
        _lib.check(L.fss_amd_dpf_eval_keys_u32(h, n, G.U64, b, P(seeds), P(cw_s), P(extra), P(out_cw), P(xt), 4, nkeys, P(ys),
                                               None), "eval_keys_u32")
        want = O.eval_keys(p, DPF, n, G.U64, b, s0s[:, b], cws, xs, nthreads=4)
        assert (ys.cpu().numpy() == want).all()
    assert L.fss_amd_dpf_relayout_u32(h, 33, P(cd), 1, P(cw_s), P(extra), P(out_cw), None) == -1
    assert L.fss_amd_dpf_eval_keys_u32(h, 33, G.U64, 0, P(cw_s), P(cw_s), P(extra), P(out_cw), P(xt), 8, 1, P(out_cw), None) == -1


def test_ctx_scratch_and_upload(engine):
    """fss_amd_ctx_scratch: grow-only per-stream device scratch; fss_amd_upload: host bytes to device memory through
    kernel arguments, stream-ordered, any size up to 1 MiB and any alignment."""
    import ctypes as C
    from fss_amd import _lib
    L, h = engine._L, engine._h
    p1, p2, p3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(L.fss_amd_ctx_scratch(h, None, 1000, C.byref(p1)), "scratch")
    _lib.check(L.fss_amd_ctx_scratch(h, None, 500, C.byref(p2)), "scratch")
    assert p1.value == p2.value and p1.value % 256 == 0           # no shrink, no realloc for a smaller request
    st = torch.cuda.Stream()
    _lib.check(L.fss_amd_ctx_scratch(h, C.c_void_p(st.cuda_stream), 1 << 20, C.byref(p3)), "scratch")
    assert p3.value != p1.value                                    # another stream, another buffer
    assert L.fss_amd_ctx_scratch(h, None, 16, None) == -1
    dst = torch.zeros(5003, dtype=torch.uint8, device="cuda")
    src = np.random.RandomState(3).randint(0, 256, size=5000, dtype=np.uint8)
    for off, nbytes in ((0, 5000), (1, 4999), (3, 17), (2, 2049), (0, 1)):
        dst.zero_()
        _lib.check(L.fss_amd_upload(h, C.c_void_p(dst.data_ptr() + off), src.ctypes.data_as(C.c_void_p), nbytes, None), "upload")
        got = dst.cpu().numpy()
        assert (got[off:off + nbytes] == src[:nbytes]).all() and not got[:off].any() and not got[off + nbytes:].any()
    assert L.fss_amd_upload(h, C.c_void_p(dst.data_ptr()), src.ctypes.data_as(C.c_void_p), (1 << 20) + 1, None) == -1
    assert L.fss_amd_upload(h, None, src.ctypes.data_as(C.c_void_p), 16, None) == -1
