import sys; sys.path.insert(0, ".")
import torch, fss_amd
from bench import make_key, time_steady
eng = fss_amd.default_engine(0)
for scheme, n in (("dpf", 20), ("dpf", 24), ("ht", 24)):
    sid, s0s, cws, ocw, _, _ = make_key(eng, scheme, n, fss_amd.U64, 7 + n)
    sd, cd = s0s[0].cuda(), cws.cuda()
    od = None if ocw is None else ocw.cuda()
    ys = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
    fn = lambda: eng.eval_all(sid, n, fss_amd.U64, 0, sd, cd, ocw=od, out=ys)
    fn(); torch.cuda.synchronize(); eager = ys.clone()
    _, k0 = time_steady(fn)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn()
    ys.zero_(); g.replay(); torch.cuda.synchronize()
    ok = bool((ys == eager).all())
    _, k1 = time_steady(lambda: g.replay())
    print(f"{scheme} EvalAll n={n}: eager {k0*1e3:.1f} us, graph replay {k1*1e3:.1f} us, equal={ok}")
