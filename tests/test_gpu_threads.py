"""Host threads sharing one context.  include/fss_amd.h allows eval calls on one ctx from several host threads at once;
hipStreamPerThread is ONE handle value that names a different stream in every thread, so per-stream state keyed by
the raw handle would be shared by launches that run concurrently (round-2 advice: WaveQ's launch-wide tail counters --
tickets interleave, chunks are skipped, outputs stay unwritten).  The context keys such state on (handle, thread)."""
import ctypes as C
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu

HIP_STREAM_PER_THREAD = 2  # hipStreamPerThread = (hipStream_t)2, hip_runtime_api.h


@pytest.mark.parametrize("plan", ["plain", "auto"])
def test_two_threads_on_hip_stream_per_thread_large_batches(plan):
    """plain: the dynamic walk with the launch-wide tail (its counters are per-stream state).  auto: the shared-prefix
    plan FSS_AMD_POINTS_AUTO picks for a batch this size -- its node table is per-stream state too (one buffer per
    (handle, thread)), and "table, then walk" must be enqueued whole by each thread."""
    import fss_amd
    from bench import make_key
    eng = fss_amd.Engine(0).set_points_plan(plan)
    L, h = eng._L, eng._h
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    n, N, G = 32, 1 << 23, fss_amd.U64   # 2^23 points: the dynamic walk with the launch-wide tail
    sid, s0s, cws, _, _, _ = make_key(eng, "dpf", n, G, 0x7EAD)
    sd, cd = s0s[0].cuda(), cws.cuda()
    gen = torch.Generator(device="cuda").manual_seed(99)
    xs = [torch.randint(-2**31, 2**31, (N,), dtype=torch.int32, device="cuda", generator=gen) for _ in range(2)]
    want = [eng.eval_points(sid, n, G, 0, sd, cd, x).clone() for x in xs]
    assert eng.last_kernel()[0] == ("dpf_eval_points" if plan == "plain" else "dpf_eval_points_shared")
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


@pytest.mark.parametrize("plan,nthreads", [("plain", 1100), ("auto", 48)])
def test_many_short_lived_threads_release_their_per_thread_stream_state(plan, nthreads):
    """1 100 threads, one after the other, each launching a dynamic walk on its own hipStreamPerThread stream and
    releasing the stream's state when done: the context has 1 024 counter pairs, so the later threads run on pairs the
    earlier ones handed back (fss_amd_ctx_scratch_release) -- every launch must still write every output.
    auto: every thread's launch takes the shared-prefix plan, whose node table is a per-(stream, thread) buffer of the
    context that the same release call frees (48 threads x 70 MiB would otherwise pile up)."""
    import fss_amd
    from bench import make_key
    eng = fss_amd.Engine(0).set_points_plan(plan)
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

    for i in range(nthreads):
        t = threading.Thread(target=once, args=(i, i % 100 == 0 or i >= nthreads - 80 or plan == "auto"))
        t.start()
        t.join(120)
    assert not bad, bad[:10]


def test_four_streams_mixed_workloads_on_one_context():
    """One context, four streams, every kind of launch in flight at once (point walks with the launch-wide tail, keys,
    the Grotto walk, one-launch EvalAll, batch EvalAll, device Gen): each output equals the one-stream result."""
    import fss_amd
    from fss_amd.engine import DPF, DCF, HT, GROTTO
    eng = fss_amd.default_engine(0)
    engp = fss_amd.Engine(0).set_points_plan("plain")  # same keys; its large point batches walk from the root
    G64, GB = fss_amd.U64, fss_amd.BYTES
    gen = torch.Generator(device="cuda").manual_seed(0x57EA)
    ri = lambda *shape: torch.randint(-2**31, 2**31, shape, dtype=torch.int32, device="cuda", generator=gen)
    r64 = lambda n: torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=gen)
    s0s, al, be = ri(1, 2, 4), ri(1), ri(1, 4)
    seed = s0s[0, 0].contiguous()
    cw_dpf, _ = eng.gen(DPF, 32, G64, s0s, al, be)
    cw_dcf, _ = eng.gen(DCF, 32, G64, s0s, al, be)
    cw_20, _ = eng.gen(DPF, 20, G64, s0s, al & 0xFFFFF, be)
    xs23, xs21 = ri(1 << 23), ri(1 << 21)
    nk = 1 << 20
    ks0, kbe, kx = ri(nk, 2, 4), ri(nk, 4), r64(nk)
    kseeds = ks0[:, 0].contiguous()
    hkeys, hocw = eng.gen_level_major(HT, 64, G64, ks0, r64(nk), kbe)
    gkeys, _ = eng.gen_level_major(GROTTO, 64, GB, ks0, r64(nk), torch.zeros_like(kbe))
    bk = 256
    bcw, _ = eng.gen(DPF, 10, G64, ks0[:bk].contiguous(), ri(bk) & 0x3FF, kbe[:bk].contiguous())
    bseeds = ks0[:bk, 0].contiguous()
    ops = [
        lambda: engp.eval_points(DPF, 32, G64, 0, seed, cw_dpf[0], xs23),                    # dynamic + tail
        lambda: engp.eval_points(DCF, 32, G64, 0, seed, cw_dcf[0], xs21),
        lambda: eng.eval_points(DPF, 32, G64, 0, seed, cw_dpf[0], xs23),                     # FSS_AMD_POINTS_AUTO: shared prefix
        lambda: eng.eval_points(DCF, 32, G64, 0, seed, cw_dcf[0], xs21),
        lambda: eng.eval_keys(hkeys, G64, 0, kseeds, kx, ocws=hocw),
        lambda: eng.eval_keys(gkeys, GB, 0, kseeds, kx),                                    # Grotto walk (tail)
        lambda: eng.eval_all(DPF, 20, G64, 0, seed, cw_20[0]),
        lambda: eng.eval_all_batch(DPF, 10, G64, 0, bseeds, bcw),
        lambda: eng.gen(DPF, 32, G64, ks0[:1 << 16].contiguous(), ri(1 << 16), kbe[:1 << 16].contiguous())[0],
        lambda: eng.eval_points_shared(DPF, 32, G64, 0, seed, cw_dpf[0], xs21),
    ]
    # the Gen op draws its alphas anew on every call: fix them
    al16 = ri(1 << 16)
    ops[8] = lambda: eng.gen(DPF, 32, G64, ks0[:1 << 16].contiguous(), al16, kbe[:1 << 16].contiguous())[0]
    want = [op().clone() for op in ops]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    for rnd in range(3):
        got = []
        for i in range(len(ops) * 2):
            j = (i + rnd) % len(ops)
            with torch.cuda.stream(streams[i % 4]):
                got.append((j, ops[j]()))
        torch.cuda.synchronize()
        for j, y in got:
            assert torch.equal(y, want[j]), (rnd, j)
