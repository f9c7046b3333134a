"""Verifiable DPF on the GPU (VDPF point evaluation) against the oracle's restatement
of include/fss/vdpf.cuh + include/fss/hash/blake3.cuh; through the C ABI (fss_amd_vdpf_*)."""
import numpy as np
import pytest

import groups as G
from conftest import SplitMix64
from test_gpu_parity import T, rand_below, xs_tensor

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[("aes128_mmo", "blake3"), ("chacha", "blake3"), ("aes128_mmo", "sha256")],
                ids=lambda p: "-".join(p))
def veng(request, O):
    """An engine per (PRG, hash pair); the oracle's process-wide hash switch follows it for the module's tests."""
    import fss_amd
    prg, hname = request.param
    e = fss_amd.Engine(0, prg=prg)
    if hname == "sha256":  # fss::hash::Sha256(key): the pair the reference's CPU benchmarks use with the AES PRG
        e.set_hash_sha256(HASH_IV_WORDS[:4])
        O.set_hash_kind(1)
    else:
        e.set_hash_iv([int(v) & 0xFFFFFFFF for v in HASH_IV_WORDS])
        O.set_hash_kind(0)
    yield e
    O.set_hash_kind(0)


HASH_IV_WORDS = [0x11111111, 0x22222222, 0x33333333, 0x44444444, 0x55555555, 0x66666666, 0x77777777, 0x88888888]


def oprg(O, eng):
    return O.Prg(2, chacha=(eng.prg == "chacha"))


@pytest.mark.parametrize("n,g", [(1, G.BYTES), (16, G.U64), (20, G.U32), (32, G.U64), (33, G.BYTES), (64, G.U128),
                                 (100, G.U64), (128, G.BYTES)])
def test_vdpf_gen_eval_keys_vs_oracle(veng, O, n, g):
    rng = SplitMix64(4000 + n)
    nkeys = 137
    p = oprg(O, veng)
    s0s = np.stack([np.stack([rng.i4(clear_lsb=False), rng.i4(clear_lsb=False)]) for _ in range(nkeys)])
    alphas = [rand_below(rng, n) for _ in range(nkeys)]
    xs = [alphas[i] if i % 2 else rand_below(rng, n) for i in range(nkeys)]
    betas = np.stack([rng.i4(clear_lsb=False) for _ in range(nkeys)])
    want = [O.vdpf_gen(p, n, g, s0s[k], alphas[k], betas[k]) for k in range(nkeys)]
    cws, cs, ocws, retry = veng.vdpf_gen(n, g, T(s0s), xs_tensor(alphas, n), T(betas))
    assert (cws.cpu().numpy() == np.stack([w[0] for w in want])).all()
    assert (cs.cpu().numpy() == np.stack([w[1] for w in want])).all()
    assert (retry.cpu().numpy() == np.array([w[3] for w in want])).all()
    ok = retry.cpu().numpy() == 0
    assert (ocws.cpu().numpy()[ok] == np.stack([w[2] for w in want])[ok]).all()
    # level-major: relayout of the key-major keys == Gen writing level-major directly
    keys = veng.vdpf_relayout(n, cws)
    lm, cs2, ocws2, retry2 = veng.vdpf_gen(n, g, T(s0s), xs_tensor(alphas, n), T(betas), level_major=True)
    assert torch.equal(lm.cw_s, keys.cw_s) and torch.equal(lm.tr_bits, keys.tr_bits)
    assert torch.equal(cs2, cs) and torch.equal(ocws2, ocws) and torch.equal(retry2, retry)
    assert (keys.cw_s.cpu().numpy() == np.stack([w[0] for w in want])[:, :, :4].transpose(1, 0, 2)).all()
    both = []
    for b in (0, 1):
        exp = [O.vdpf_eval(p, n, g, b, s0s[k, b], want[k][0], want[k][1], want[k][2], xs[k]) for k in range(nkeys)]
        ys, pis = veng.vdpf_eval_keys(keys, g, b, T(np.ascontiguousarray(s0s[:, b])), cs, ocws, xs_tensor(xs, n))
        assert (ys.cpu().numpy() == np.stack([e[0] for e in exp])).all(), (n, g, b)
        assert (pis.cpu().numpy() == np.stack([e[1] for e in exp])).all(), (n, g, b)
        proofs = veng.vdpf_prove_keys(pis, cs).cpu().numpy()
        for k in range(0, nkeys, 17):
            assert (proofs[k] == O.vdpf_prove(exp[k][1][None], want[k][1])).all()
        both.append((ys.cpu().numpy(), proofs))
    assert (both[0][1] == both[1][1]).all()  # Verify: the two parties' proofs agree
    for k in range(nkeys):
        tot = G.add(g, both[0][0][k], both[1][0][k])
        bk = betas[k].copy()
        bk[3] &= ~1  # Gen clamps beta (vdpf.cuh:111)
        assert (tot == (G.canon(g, bk) if xs[k] == alphas[k] else np.zeros(4, np.int32))).all()


@pytest.mark.parametrize("n,g", [(1, G.U64), (7, G.BYTES), (32, G.U64), (64, G.U64), (65, G.U128)])
def test_vdpf_eval_points_and_prove_vs_oracle(veng, O, n, g):
    rng = SplitMix64(4100 + n)
    p = oprg(O, veng)
    s = [rng.i4(), rng.i4()]
    alpha = rand_below(rng, n)
    beta = rng.i4()
    cws, cs, ocw, rc = O.vdpf_gen(p, n, g, s, alpha, beta)
    assert rc == 0
    xs = [0, (1 << n) - 1, alpha] + [rand_below(rng, n) for _ in range(300)]
    proofs = []
    for b in (0, 1):
        exp = [O.vdpf_eval(p, n, g, b, s[b], cws, cs, ocw, x) for x in xs]
        ys, pis = veng.vdpf_eval_points(n, g, b, T(s[b]), T(cws), T(cs), T(ocw), xs_tensor(xs, n))
        assert (ys.cpu().numpy() == np.stack([e[0] for e in exp])).all()
        assert (pis.cpu().numpy() == np.stack([e[1] for e in exp])).all()
        pi = veng.vdpf_prove(pis, T(cs)).cpu().numpy()
        assert (pi == O.vdpf_prove(np.stack([e[1] for e in exp]), cs)).all()
        proofs.append(pi)
    assert (proofs[0] == proofs[1]).all()
    empty = veng.vdpf_prove(torch.empty((0, 4, 4), dtype=torch.int32), T(cs)).cpu().numpy()
    assert (empty == cs).all()  # Prove of no evaluations is cs (vdpf.cuh:270)


def test_vdpf_python_surface(veng, O):
    import fss_amd
    n = 16
    v = fss_amd.Vdpf(in_bits=n, group="uint", prg=veng.prg, engine=veng)
    gen = torch.Generator().manual_seed(3)
    s0s = torch.randint(-2**31, 2**31, (2, 4), dtype=torch.int32, generator=gen)
    beta = torch.tensor([604, 0, 0, 0], dtype=torch.int32)
    cws, cs, ocw = v.gen(s0s, 107, beta)
    assert cws.shape == (n, 8) and cs.shape == (4, 4) and ocw.shape == (4,)
    xs = torch.arange(0, 1 << n, 257, dtype=torch.int32)
    out = [v.eval_points(b, s0s[b], cws, cs, ocw, xs) for b in (0, 1)]
    tot = (out[0][0][:, 0].to(torch.int64) + out[1][0][:, 0]) & 0xFFFFFFFF
    assert int(tot.sum()) == 0  # alpha = 107 is not among the points
    y0, p0 = v.eval(0, s0s[0], cws, cs, ocw, 107)
    y1, p1 = v.eval(1, s0s[1], cws, cs, ocw, 107)
    assert ((int(y0[0]) + int(y1[0])) & 0xFFFFFFFF) == 604 and torch.equal(p0, p1)
    assert v.verify(v.prove(out[0][1], cs), v.prove(out[1][1], cs))
    bad = cws.clone()
    bad[5, 1] ^= 4
    pb = v.eval_points(1, s0s[1], bad, cs, ocw, xs)[1]
    assert not v.verify(v.prove(out[0][1], cs), v.prove(pb, cs))


@pytest.mark.parametrize("n", [1, 20, 32])
def test_vdpf_u32_extra_entry_points_vs_u64(veng, n):
    """fss_amd_vdpf_relayout_u32 / fss_amd_vdpf_eval_keys_u32: the reference's own `uint32_t* extra` packing of the tr bits
    (VdpfRelayoutGpu / VdpfEvalPointGpu, point_eval_gpu.cuh:389-396, :514-526; in_bits <= 32) gives the outputs and hashes
    of the 64-bit form; in_bits = 33 is refused."""
    import ctypes as C
    from fss_amd import _lib
    L, h = _lib.lib(), veng._h
    rng = SplitMix64(4400 + n)
    nkeys, g = 333, G.U64
    s0s = np.stack([np.stack([rng.i4(clear_lsb=False), rng.i4(clear_lsb=False)]) for _ in range(nkeys)])
    alphas = [rand_below(rng, n) for _ in range(nkeys)]
    xs = [alphas[i] if i % 2 else rand_below(rng, n) for i in range(nkeys)]
    betas = np.stack([rng.i4(clear_lsb=False) for _ in range(nkeys)])
    cws, cs, ocws, _ = veng.vdpf_gen(n, g, T(s0s), xs_tensor(alphas, n), T(betas))
    keys = veng.vdpf_relayout(n, cws)
    P = lambda t: C.c_void_p(t.data_ptr())
    cw_s = torch.empty_like(keys.cw_s)
    extra = torch.empty((nkeys,), dtype=torch.int32, device="cuda")
    _lib.check(L.fss_amd_vdpf_relayout_u32(h, n, P(cws), nkeys, P(cw_s), P(extra), None), "vdpf_relayout_u32")
    torch.cuda.synchronize()
    assert torch.equal(cw_s, keys.cw_s)
    assert (extra.cpu().numpy().view(np.uint32).astype(np.uint64) == keys.tr_bits.cpu().numpy().view(np.uint64)).all()
    xt = xs_tensor(xs, n).cuda()
    for b in (0, 1):
        seeds = T(np.ascontiguousarray(s0s[:, b])).cuda()
        ys, pis = veng.vdpf_eval_keys(keys, g, b, seeds, cs, ocws, xt)
        ys2, pis2 = torch.empty_like(ys), torch.empty_like(pis)
        _lib.check(L.fss_amd_vdpf_eval_keys_u32(h, n, g, b, P(seeds), P(cw_s), P(extra), P(cs), P(ocws), P(xt), 4, nkeys, P(ys2), P(pis2),
                                                None), "vdpf_eval_keys_u32")
        torch.cuda.synchronize()
        assert torch.equal(ys, ys2) and torch.equal(pis, pis2)
    assert L.fss_amd_vdpf_relayout_u32(h, 33, P(cws), 1, P(cw_s), P(extra), None) == -1
    assert L.fss_amd_vdpf_eval_keys_u32(h, 33, g, 0, P(cw_s), P(cw_s), P(extra), P(cs), P(ocws), P(xt), 8, 1, P(cw_s), P(cw_s), None) == -1
