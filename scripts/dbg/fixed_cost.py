import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
e = Engine()
cfg = workloads.config2(1024, seed=1234)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
for n in (1, 4, 64, 256, 512, 1024):
    P = torch.from_numpy(cfg["walkers"][:n].copy()).cuda()
    o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
    r = []
    for mx in (1, 2, 6):
        e.set_iteration_limits(0, mx)
        e.time_lnprob_torch(P, *o, reps=3)
        r.append(np.median([e.time_lnprob_torch(P, *o, reps=1) for _ in range(15)]))
    e.set_iteration_limits(10, 200)
    print("N=%4d: maxiter 1 / 2 / 6: %.1f %.1f %.1f us" % (n, r[0] * 1e3, r[1] * 1e3, r[2] * 1e3), flush=True)
