"""20 launches of the first N walkers of the headline batch with the iteration limit at 1 (for rocprofv3 --kernel-trace --stats:
the kernel's own duration against the HIP-event time of scripts/dbg/fixed_cost.py).  usage: fixed_cost_one.py N"""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
n = int(sys.argv[1])
e = Engine()
cfg = workloads.config2(1024, seed=1234)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
P = torch.from_numpy(cfg["walkers"][:n].copy()).cuda()
o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
e.set_iteration_limits(0, 1)
ts = [e.time_lnprob_torch(P, *o, reps=1) for _ in range(20)]
print("N=%d maxiter 1: HIP events median %.1f us" % (n, np.median(ts) * 1e3))
