"""Launch time of N prior-box walkers with one and with two wavefronts per SIMD (issue order on)."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
e = Engine()
for n in (1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768):
    cfg = workloads.config2(n, seed=4242)
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    P = torch.from_numpy(cfg["walkers"]).cuda()
    o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
    r = []
    for occ in (1, 2):
        e.set_waves_per_simd(occ)
        e.time_lnprob_torch(P, *o, reps=2)
        r.append(e.time_lnprob_torch(P, *o, reps=6))
    e.set_waves_per_simd(0)
    print("N=%6d  1 wave/SIMD %.3f ms (%.2f M/s)   2 waves/SIMD %.3f ms (%.2f M/s)   maxiter walkers %d"
          % (n, r[0], n / r[0] / 1e3, r[1], n / r[1] / 1e3, int((o[1] == 1).sum())), flush=True)
