"""Does a never-converging walker run slower when another one shares its CU?  1024-walker launches in which
every walker but two is rejected by the prior (finishes at once); the two slow ones sit at chosen queue
positions (workgroup = position // 4 when the wavefronts take their items in launch order)."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
n = 1024
cfg = workloads.config2(n, seed=1234)
e = Engine(); e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
lnp, st, nit = e.lnprob_batch(cfg["walkers"], return_info=True)
slow = cfg["walkers"][st == 1]
print("maxiter walkers:", len(slow))
out = cfg["walkers"][0].copy(); out[0] = 99.0                       # outside the box
o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
def run(positions, label):
    W = np.tile(out, (n, 1))
    for k, p in enumerate(positions): W[p] = slow[k % len(slow)]
    P = torch.from_numpy(W).cuda()
    e.time_lnprob_torch(P, *o, reps=5)
    ts = [e.time_lnprob_torch(P, *o, reps=10) for _ in range(5)]
    print("%-44s %s  niter %s" % (label, " ".join("%.4f" % t for t in ts), o[2][list(positions)].tolist()), flush=True)
run([0], "one slow walker")
run([0, 1], "two, positions 0,1 (same workgroup)")
run([0, 2], "two, positions 0,2 (same workgroup)")
run([0, 4], "two, positions 0,4 (next workgroup)")
run([0, 32], "two, positions 0,32 (workgroup 8: same XCD?)")
run([0, 1, 2, 3], "four in one workgroup")
run([0, 4, 8, 12], "four in four workgroups")
run(list(range(0, 92, 4)), "23 slow, one per workgroup")
run(list(range(0, 23)), "23 slow, packed 4 per workgroup")
run(list(np.random.RandomState(1).choice(1024, 23, replace=False)), "23 slow, random positions")
