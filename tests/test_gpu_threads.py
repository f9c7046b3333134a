"""Host threads sharing one context.  include/fss_amd.h allows eval calls on one ctx from several host threads at once;
hipStreamPerThread is ONE handle value that names a different stream in every thread, so per-stream state keyed by
the raw handle would be shared by launches that run concurrently (round-2 advice: WaveQ's launch-wide tail counters --
tickets interleave, chunks are skipped, outputs stay unwritten).  The context keys such state on (handle, thread)."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HIP_STREAM_PER_THREAD = 2  # hipStreamPerThread = (hipStream_t)2, hip_runtime_api.h


def test_two_threads_on_hip_stream_per_thread_large_batches():
    import fss_amd
    from bench import make_key
    eng = fss_amd.default_engine(0)
    L, h = eng._L, eng._h
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    n, N, G = 32, 1 << 23, fss_amd.U64   # 2^23 points: the dynamic walk with the launch-wide tail
    sid, s0s, cws, _, _, _ = make_key(eng, "dpf", n, G, 0x7EAD)
    sd, cd = s0s[0].cuda(), cws.cuda()
    gen = torch.Generator(device="cuda").manual_seed(99)
    xs = [torch.randint(-2**31, 2**31, (N,), dtype=torch.int32, device="cuda", generator=gen) for _ in range(2)]
    want = [eng.eval_points(sid, n, G, 0, sd, cd, x).clone() for x in xs]
    assert eng.last_kernel()[0] == "dpf_eval_points"
    ys = [torch.full((N, 4), -1, dtype=torch.int32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    errs, rounds = [], 6
    start = threading.Barrier(2)

    def worker(i):
        try:
            st = C.c_void_p(HIP_STREAM_PER_THREAD)
            for _ in range(rounds):
                start.wait()
                rc = L.fss_amd_dpf_eval_points(h, n, G, 0, C.c_void_p(sd.data_ptr()), C.c_void_p(cd.data_ptr()),
                                               C.c_void_p(xs[i].data_ptr()), 4, N, C.c_void_p(ys[i].data_ptr()), st)
                assert rc == 0, rc
                assert hip.hipStreamSynchronize(st) == 0
                if not bool((ys[i] == want[i]).all()):
                    errs.append(f"thread {i}: outputs differ from the single-stream run")
                    return
                ys[i].fill_(-1)
                torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
            start.abort()

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs


def test_scratch_is_per_thread_for_hip_stream_per_thread_and_can_be_released():
    import fss_amd
    eng = fss_amd.default_engine(0)
    L, h = eng._L, eng._h
    got = {}

    def ask(name):
        p = C.c_void_p()
        assert L.fss_amd_ctx_scratch(h, C.c_void_p(HIP_STREAM_PER_THREAD), 1 << 16, C.byref(p)) == 0
        got[name] = p.value

    ask("main")
    t = threading.Thread(target=ask, args=("other",))
    t.start(); t.join()
    ask("main_again")
    assert got["main"] == got["main_again"] and got["main"] != got["other"]
    assert L.fss_amd_ctx_scratch_release(h, C.c_void_p(HIP_STREAM_PER_THREAD)) == 0
    assert L.fss_amd_ctx_scratch_release(h, C.c_void_p(HIP_STREAM_PER_THREAD)) == 0  # nothing left: still fine


def test_many_short_lived_threads_release_their_per_thread_stream_state():
    """1 100 threads, one after the other, each launching a dynamic walk on its own hipStreamPerThread stream and
    releasing the stream's state when done: the context has 1 024 counter pairs, so the later threads run on pairs the
    earlier ones handed back (fss_amd_ctx_scratch_release) -- every launch must still write every output."""
    import fss_amd
    from bench import make_key
    eng = fss_amd.default_engine(0)
    L, h = eng._L, eng._h
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    n, N, G = 32, 1 << 23, fss_amd.U64
    sid, s0s, cws, _, _, _ = make_key(eng, "dpf", n, G, 0x7EAE)
    sd, cd = s0s[0].cuda(), cws.cuda()
    xs = torch.randint(-2**31, 2**31, (N,), dtype=torch.int32, device="cuda")
    want = eng.eval_points(sid, n, G, 0, sd, cd, xs).clone()
    ys = torch.empty((N, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    bad = []

    def once(i, check):
        st = C.c_void_p(HIP_STREAM_PER_THREAD)
        if check:
            ys.fill_(-1)
            torch.cuda.synchronize()
        rc = L.fss_amd_dpf_eval_points(h, n, G, 0, C.c_void_p(sd.data_ptr()), C.c_void_p(cd.data_ptr()),
                                       C.c_void_p(xs.data_ptr()), 4, N, C.c_void_p(ys.data_ptr()), st)
        ok = rc == 0 and hip.hipStreamSynchronize(st) == 0
        if ok and check:
            ok = bool((ys == want).all())
        ok = ok and L.fss_amd_ctx_scratch_release(h, st) == 0
        if not ok:
            bad.append(i)

    for i in range(1100):
        t = threading.Thread(target=once, args=(i, i % 100 == 0 or i >= 1020))
        t.start()
        t.join(120)
    assert not bad, bad[:10]
