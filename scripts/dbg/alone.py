"""Launch time of each never-converging walker of the 1024-walker batch ALONE on the chip (every other walker is
rejected by the prior): the launch time the batch would have without two of them on one CU = the maximum."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
n = 1024
cfg = workloads.config2(n, seed=1234)
e = Engine(); e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
lnp, st, nit = e.lnprob_batch(cfg["walkers"], return_info=True)
slow = np.where(st == 1)[0]
out = cfg["walkers"][0].copy(); out[0] = 99.0
o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
res = []
for w in slow:
    W = np.tile(out, (n, 1)); W[0] = cfg["walkers"][w]
    P = torch.from_numpy(W).cuda()
    e.time_lnprob_torch(P, *o, reps=3)
    res.append(e.time_lnprob_torch(P, *o, reps=10))
print("alone: " + " ".join("%d:%.4f" % (w, t) for w, t in zip(slow, res)))
print("max %.4f ms  min %.4f ms" % (max(res), min(res)))
