"""alone.py for two fixed walkers (one of each speed class) with more repetitions: a precise per-iteration instrument."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
n = 1024
cfg = workloads.config2(n, seed=1234)
e = Engine(); e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
out = cfg["walkers"][0].copy(); out[0] = 99.0
o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
for w in (741, 724):
    W = np.tile(out, (n, 1)); W[0] = cfg["walkers"][w]
    P = torch.from_numpy(W).cuda()
    e.time_lnprob_torch(P, *o, reps=5)
    ts = [e.time_lnprob_torch(P, *o, reps=10) for _ in range(3)]
    print("walker %d alone: %s ms (niter %d)" % (w, " ".join("%.4f" % t for t in ts), int(o[2][0])))
