"""Contexts come and go without leaving device memory behind: fss_amd_ctx_create / fss_amd_ctx_destroy around the
launches that make a context allocate on demand (the tables, the per-stream counter pair of the launch-wide tail
queue, per-stream scratch)."""
import gc

import pytest

import groups as G

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

DPF, DCF, HT, GROTTO = 0, 1, 2, 3
MiB = 1 << 20


def _free():
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]


def test_contexts_do_not_leak_device_memory():
    from fss_amd.engine import Engine
    gen = torch.Generator(device="cuda").manual_seed(5)
    n = 1 << 23  # large enough for the dynamic hand-out with its launch-wide tail
    xs = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda", generator=gen)
    ys = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    s0s = torch.randint(-2**31, 2**31, (1, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    al = torch.randint(-2**31, 2**31, (1,), dtype=torch.int32, device="cuda", generator=gen)
    be = torch.randint(-2**31, 2**31, (1, 4), dtype=torch.int32, device="cuda", generator=gen)
    side = torch.cuda.Stream()
    first = None

    def once(prg):
        nonlocal first
        e = Engine(prg=prg).set_points_plan("plain")
        cws, _ = e.gen(DPF, 32, G.U64, s0s, al, be)
        e.eval_points(DPF, 32, G.U64, 0, s0s[0, 0].contiguous(), cws[0], xs, out=ys)
        assert e.last_kernel()[0] == "dpf_eval_points"
        e.set_points_plan("auto")  # the shared-prefix plan's per-stream node table (64 MiB here) goes with the context too
        e.eval_points(DPF, 32, G.U64, 0, s0s[0, 0].contiguous(), cws[0], xs, out=ys)
        assert e.last_kernel()[0] == "dpf_eval_points_shared"
        e.set_points_plan("plain")
        with torch.cuda.stream(side):  # a second stream: its own counter pair
            e.eval_points(DPF, 32, G.U64, 1, s0s[0, 1].contiguous(), cws[0], xs[:1 << 22], out=ys[:1 << 22])
        e.eval_all(DPF, 16, G.U64, 0, s0s[0, 0].contiguous(), e.gen(DPF, 16, G.U64, s0s, al & 0xFFFF, be)[0][0])
        torch.cuda.synchronize()
        if first is None:
            first = ys[:64].clone()
        del e

    once("aes128_mmo")
    once("chacha")
    free0 = _free()
    for i in range(60):
        once("aes128_mmo" if i % 3 else "chacha")
    free1 = _free()
    assert free0 - free1 < 16 * MiB, (free0 - free1) / MiB


def test_release_stream_frees_the_shared_prefix_table():
    """FSS_AMD_POINTS_AUTO keeps a node table per stream in the context (64 MiB for 2^23 points); Engine.release_stream
    gives it back without destroying the context, and a later call on that stream builds a new one."""
    from fss_amd.engine import Engine
    gen = torch.Generator(device="cuda").manual_seed(6)
    n = 1 << 23
    xs = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device="cuda", generator=gen)
    ys = torch.empty((n, 4), dtype=torch.int32, device="cuda")
    s0s = torch.randint(-2**31, 2**31, (1, 2, 4), dtype=torch.int32, device="cuda", generator=gen)
    al = torch.randint(-2**31, 2**31, (1,), dtype=torch.int32, device="cuda", generator=gen)
    be = torch.randint(-2**31, 2**31, (1, 4), dtype=torch.int32, device="cuda", generator=gen)
    e = Engine()
    cws, _ = e.gen(DPF, 32, G.U64, s0s, al, be)
    seed = s0s[0, 0].contiguous()
    free0 = _free()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        e.eval_points(DPF, 32, G.U64, 0, seed, cws[0], xs, out=ys)
        assert e.last_kernel()[0] == "dpf_eval_points_shared"
        first = ys[:64].clone()
    held = free0 - _free()
    assert held >= 60 * MiB, held / MiB           # the 2^22-node table (64 MiB) + stage buffers
    e.release_stream(side)
    assert free0 - _free() < 32 * MiB
    with torch.cuda.stream(side):
        ys.zero_()
        e.eval_points(DPF, 32, G.U64, 0, seed, cws[0], xs, out=ys)
        side.synchronize()
        assert torch.equal(ys[:64], first)
    del e
    assert free0 - _free() < 32 * MiB   # (the runtime keeps a few MiB of freed blocks to itself; the table alone is 64)
