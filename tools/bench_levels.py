import sys, torch
sys.path.insert(0, "/root/repo")
import fss_amd
from bench import make_key, time_steps
eng = fss_amd.default_engine(0)
for lp in (20, 24):
    for n in (8, 16, 32, 48, 64):
        sid, s0s, cws, ocw, _, _ = make_key(eng, "dpf", n, fss_amd.U64, 3 + n)
        batch = 1 << lp
        xs = torch.randint(0, 1 << min(n, 31), (batch,), dtype=torch.int32 if n <= 32 else torch.int64, device="cuda")
        sd, cd = s0s[0].cuda(), cws.cuda()
        ys = torch.empty((batch, 4), dtype=torch.int32, device="cuda")
        w, k = time_steps(lambda: eng.eval_points(sid, n, fss_amd.U64, 0, sd, cd, xs, out=ys), 30, 3, None)
        print(f"2^{lp} n={n}: {k*1000:.1f} us  ({k*1000/n:.2f} us/level)")
