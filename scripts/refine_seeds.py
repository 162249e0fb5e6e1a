"""GPU: the 1024-walker launch with the refinement on / off for several prior-box draws (the headline is the draw of seed 1234;
a launch lasts as long as its slowest walker, so the gain depends on which walkers a draw holds)."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
eng = Engine()
tot = {False: [], True: []}
for seed in (1234, 777, 1, 2, 3, 4, 5, 6, 7, 8):
    cfg = workloads.config2(1024, seed)
    eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    P = torch.from_numpy(cfg["walkers"]).cuda()
    o = [torch.empty(1024, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
    r = {}
    for on in (False, True):
        eng.set_refinement(on)
        eng.time_lnprob_torch(P, *o, reps=3)
        r[on] = float(np.median([eng.time_lnprob_torch(P, *o, reps=1) for _ in range(15)]))
        tot[on].append(r[on])
    print("seed %5d: %d walkers at maxiter; pivoted every iteration %.3f ms, refinement %.3f ms (%+.1f %%)"
          % (seed, int((o[1] == 1).sum()), r[False], r[True], 100 * (r[True] / r[False] - 1)), flush=True)
print("mean: %.3f -> %.3f ms" % (np.mean(tot[False]), np.mean(tot[True])))
