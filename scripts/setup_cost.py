"""Per-walker fixed cost (set-up + epilogue) from launches with the iteration limit at 1, 2, 3:
python scripts/setup_cost.py  (RADEX_EMCEE_AMD_LIB selects the build)"""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
e = Engine()
for n in (1024, 32768):
    cfg = workloads.config2(n, seed=5678)
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    P = torch.from_numpy(cfg["walkers"]).cuda()
    o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
    res = {}
    for mx in (1, 2, 3, 11):
        e.set_iteration_limits(0, mx)
        e.time_lnprob_torch(P, *o, reps=2)
        res[mx] = e.time_lnprob_torch(P, *o, reps=10)
    e.set_iteration_limits(10, 200)
    per = res[3] - res[2]
    waves = 1024 if n == 1024 else 2048
    rounds = n / waves
    print("N=%d: maxiter 1/2/3/11: %.4f %.4f %.4f %.4f ms; per round: iteration %.2f us, fixed %.2f us"
          % (n, res[1], res[2], res[3], res[11], per / rounds * 1e3, (res[1] - per) / rounds * 1e3))
