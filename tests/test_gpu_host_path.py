"""The PCIe-inclusive path (fss_amd_eval_points_host, Engine.eval_points_host): a batch whose points and outputs live in
pinned host memory is walked in ONE launch that reads x and writes y across PCIe itself.  Outputs must be those of the
resident call -- which the other tests pin to the oracle -- for every scheme, at ragged sizes either side of the
quad-walk and two-then-one thresholds; pageable memory is refused."""
import ctypes as C

import numpy as np
import pytest

import groups as G
from conftest import SplitMix64

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

DPF, DCF, HT = 0, 1, 2


@pytest.mark.parametrize("n_points", [1, 63, 4097, (1 << 17) + 5, (1 << 20) + 77])
@pytest.mark.parametrize("scheme,in_bits", [(DPF, 32), (DCF, 32), (HT, 40), (DPF, 64)])
def test_host_batch_equals_resident_batch(engine, O, scheme, in_bits, n_points):
    rng = SplitMix64(0xC0FFEE + scheme * 131 + in_bits + n_points % 1009)
    g = G.U64
    p = O.Prg({DPF: 2, DCF: 4, HT: 1}[scheme])
    s = [rng.i4(), rng.i4()]
    alpha, beta = rng.below(in_bits), rng.i4()
    ocw = None
    if scheme == DPF:
        cws = O.dpf_gen(p, in_bits, g, s, alpha, beta)
    elif scheme == DCF:
        cws = O.dcf_gen(p, in_bits, g, 0, s, alpha, beta)
    else:
        cws, ocw = O.ht_gen(p, in_bits, g, s, alpha, beta)
    gen = torch.Generator().manual_seed(n_points + in_bits)
    if in_bits <= 32:
        xs = torch.randint(-2**31, 2**31, (n_points,), dtype=torch.int32, generator=gen)
        a_t = alpha - (1 << 32) if alpha >= (1 << 31) else alpha
    else:
        xs = torch.randint(-2**63, 2**63 - 1, (n_points,), dtype=torch.int64, generator=gen)
        if in_bits < 64:
            xs &= (1 << in_bits) - 1
        a_t = alpha - (1 << 64) if alpha >= (1 << 63) else alpha
    xs[::3] = a_t
    xs_h = xs.pin_memory()
    cd = torch.from_numpy(np.ascontiguousarray(cws)).cuda()
    od = None if ocw is None else torch.from_numpy(np.ascontiguousarray(ocw)).cuda()
    for b in (0, 1):
        sd = torch.from_numpy(s[b]).cuda()
        want = engine.eval_points(scheme, in_bits, g, b, sd, cd, xs_h.cuda(), ocw=od).cpu()
        out = torch.full((n_points + 3, 4), -1, dtype=torch.int32).pin_memory()
        got = engine.eval_points_host(scheme, in_bits, g, b, sd, cd, xs_h, ocw=od, out=out)
        torch.cuda.synchronize()
        assert got.data_ptr() == out.data_ptr()
        assert bool((out[:n_points] == want).all()), b
        assert bool((out[n_points:] == -1).all())      # nothing written past the batch
    # a small sample against the oracle directly (party 0)
    m = min(n_points, 257)
    xh = xs[:m].numpy()
    xh = (xh.view(np.uint32) if xh.dtype == np.int32 else xh.view(np.uint64)).astype(np.uint64)
    ref = O.eval_points(p, scheme, in_bits, g, 0, s[0], cws, xh, ocw=ocw)
    got0 = engine.eval_points_host(scheme, in_bits, g, 0, torch.from_numpy(s[0]).cuda(), cd, xs_h, ocw=od)
    torch.cuda.synchronize()
    assert (got0[:m].numpy() == ref).all()


@pytest.mark.parametrize("in_bits,kind", [(8, "u8"), (16, "i16"), (100, "u128")])
def test_host_batch_with_narrow_and_wide_points(engine, O, in_bits, kind):
    """x arrays of 1-, 2- and 16-byte elements (In = uint8_t / uint16_t / __uint128_t) read from pinned host memory."""
    rng = SplitMix64(7000 + in_bits)
    g = G.U64
    p = O.Prg(2)
    s = [rng.i4(), rng.i4()]
    alpha = rng.next() & ((1 << min(in_bits, 63)) - 1)
    beta = rng.i4()
    cws = O.dpf_gen(p, in_bits, g, s, alpha, beta)
    n = 3001
    vals = [alpha] + [rng.next() & ((1 << min(in_bits, 63)) - 1) for _ in range(n - 1)]
    if kind == "u8":
        xs = torch.tensor(vals, dtype=torch.int64).to(torch.uint8)
        xo = np.array(vals, dtype=np.uint64)
    elif kind == "i16":
        xs = torch.from_numpy(np.array(vals, dtype=np.uint64).astype(np.uint16).view(np.int16).copy())
        xo = np.array(vals, dtype=np.uint64)
    else:
        lo = np.array(vals, dtype=np.uint64)
        hi = np.array([rng.next() & ((1 << (in_bits - 64)) - 1) for _ in range(n)], dtype=np.uint64)
        hi[0] = 0
        xs = torch.from_numpy(np.stack([lo, hi], axis=1).view(np.int64).copy())
        xo = None
    xs_h = xs.pin_memory()
    cd = torch.from_numpy(np.ascontiguousarray(cws)).cuda()
    for b in (0, 1):
        sd = torch.from_numpy(s[b]).cuda()
        want = engine.eval_points(0, in_bits, g, b, sd, cd, xs_h.cuda()).cpu()
        got = engine.eval_points_host(0, in_bits, g, b, sd, cd, xs_h)
        torch.cuda.synchronize()
        assert bool((got == want).all()), b
        if xo is not None:
            assert (got.numpy() == O.eval_points(p, 0, in_bits, g, b, s[b], cws, xo)).all()


def test_host_batch_in_a_hip_graph(engine):
    """The host-resident call only enqueues a launch: it can be captured in a HIP graph and replayed on new points."""
    import fss_amd
    n = 1 << 18
    gen = torch.Generator().manual_seed(5)
    s0s = torch.randint(-2**31, 2**31, (1, 2, 4), dtype=torch.int32, generator=gen)
    cws, _ = engine.gen(fss_amd.DPF, 32, fss_amd.U64, s0s, torch.tensor([77], dtype=torch.int32),
                        torch.randint(-2**31, 2**31, (1, 4), dtype=torch.int32, generator=gen))
    sd, cd = s0s[0, 0].cuda(), cws[0].contiguous()
    xs_h = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, generator=gen).pin_memory()
    out = torch.zeros((n, 4), dtype=torch.int32).pin_memory()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        engine.eval_points_host(fss_amd.DPF, 32, fss_amd.U64, 0, sd, cd, xs_h, out=out)   # warm: nothing allocates inside the capture
        st.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            engine.eval_points_host(fss_amd.DPF, 32, fss_amd.U64, 0, sd, cd, xs_h, out=out)
    for rep in range(2):
        xs_h.copy_(torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, generator=gen))
        xs_h[0] = 77
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        want = engine.eval_points(fss_amd.DPF, 32, fss_amd.U64, 0, sd, cd, xs_h.cuda()).cpu()
        assert bool((out == want).all()), rep


def test_views_into_pinned_buffers(engine):
    """Interior pointers: a batch that starts in the middle of a pinned allocation (a slice of a larger pinned tensor, for
    x and for y) is walked in place like one that starts at its base."""
    import fss_amd
    n, off = 70001, 12345
    gen = torch.Generator().manual_seed(99)
    s0s = torch.randint(-2**31, 2**31, (1, 2, 4), dtype=torch.int32, generator=gen)
    cws, _ = engine.gen(fss_amd.DPF, 32, fss_amd.U64, s0s, torch.tensor([123456789], dtype=torch.int32),
                        torch.randint(-2**31, 2**31, (1, 4), dtype=torch.int32, generator=gen))
    big_x = torch.randint(-2**31, 2**31, (off + n + 77,), dtype=torch.int32, generator=gen).pin_memory()
    big_y = torch.full((off + n + 5, 4), -1, dtype=torch.int32).pin_memory()
    xs, out = big_x[off:off + n], big_y[off:off + n]
    assert xs.is_pinned() and out.is_pinned() and xs.data_ptr() != big_x.data_ptr()
    sd = s0s[0, 0].cuda()
    engine.eval_points_host(fss_amd.DPF, 32, fss_amd.U64, 0, sd, cws[0], xs, out=out)
    torch.cuda.synchronize()
    want = engine.eval_points(fss_amd.DPF, 32, fss_amd.U64, 0, sd, cws[0], xs.cuda()).cpu()
    assert bool((out == want).all())
    assert bool((big_y[:off] == -1).all()) and bool((big_y[off + n:] == -1).all())   # nothing outside the view


def test_pageable_memory_is_refused(engine):
    import fss_amd
    z = torch.zeros((64,), dtype=torch.int32)
    seed, cws = torch.zeros(4, dtype=torch.int32, device="cuda"), torch.zeros((33, 8), dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError, match="pinned"):
        engine.eval_points_host(fss_amd.DPF, 32, fss_amd.U64, 0, seed, cws, z)
    with pytest.raises(ValueError, match="pinned"):
        engine.eval_points_host(fss_amd.DPF, 32, fss_amd.U64, 0, seed, cws, z.pin_memory(), out=torch.zeros((64, 4), dtype=torch.int32))
    # the C ABI itself: a pageable pointer is FSS_AMD_EINVAL, never dereferenced by a kernel
    buf = np.zeros(64, dtype=np.int32)
    out = torch.zeros((64, 4), dtype=torch.int32).pin_memory()
    rc = engine._L.fss_amd_eval_points_host(engine._h, fss_amd.DPF, 32, fss_amd.U64, 0, C.c_void_p(seed.data_ptr()),
                                            C.c_void_p(cws.data_ptr()), None, C.c_void_p(buf.ctypes.data), 4, 64,
                                            C.c_void_p(out.data_ptr()), None)
    assert rc == -1
    # an empty batch is fine whatever the pointers
    assert engine.eval_points_host(fss_amd.DPF, 32, fss_amd.U64, 0, seed, cws, torch.zeros((0,), dtype=torch.int32).pin_memory()).shape == (0, 4)
