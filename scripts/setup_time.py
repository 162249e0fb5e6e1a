"""Diagnostic: per-iteration time and per-walker set-up time of the never-converging walkers of BASELINE
config 2 (the ones that set the launch time), from launches with different iteration limits."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
cfg = workloads.config2(1024); e = Engine()
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
lnp, st, nit = e.lnprob_batch(cfg["walkers"], return_info=True)
P = torch.from_numpy(np.ascontiguousarray(cfg["walkers"][st == 1])).cuda()
o = [torch.empty(len(P), dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
res = {}
for mx in (11, 21, 41, 101, 200):
    e.set_iteration_limits(10, mx)
    res[mx] = e.time_lnprob_torch(P, *o, reps=10)
    print("maxiter %3d: %.4f ms" % (mx, res[mx]))
per = (res[200] - res[101]) / 99
print("per iteration %.3f us ; extrapolated set-up + epilogue %.1f us" % (per * 1e3, (res[200] - 200 * per) * 1e3))
