import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fss_amd
from bench import make_key
eng = fss_amd.default_engine(0)
G = fss_amd.U64
n = 28
sid, s0s, cws, ocw, _, _ = make_key(eng, sys.argv[1] if len(sys.argv) > 1 else "dcf", n, G, 7 + n)
sd, cd = s0s[0].cuda(), cws.cuda()
ys = torch.empty((1 << n, 4), dtype=torch.int32, device="cuda")
for _ in range(6): eng.eval_all(sid, n, G, 0, sd, cd, out=ys)
torch.cuda.synchronize()
