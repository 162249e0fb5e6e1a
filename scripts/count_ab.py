import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
eng = Engine()
for n, seed in ((1024, 1234), (32768, 5678)):
    cfg = workloads.config2(n, seed)
    eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    P = torch.from_numpy(cfg["walkers"]).cuda()
    o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
    for rep in range(2):
        for cnt in (False, True):
            eng.set_refinement_counting(cnt)
            eng.time_lnprob_torch(P, *o, reps=5)
            t = np.median([eng.time_lnprob_torch(P, *o, reps=1) for _ in range(30 if n == 1024 else 8)])
            print("N=%d counting %-5s: %.4f ms" % (n, cnt, t), flush=True)
    eng.set_refinement_counting(False)
