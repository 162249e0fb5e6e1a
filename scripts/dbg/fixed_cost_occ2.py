"""The throughput regime's fixed cost per walker: launches of 32768 walkers (two wavefronts per SIMD) forced to EXACTLY k iterations
(miniter = maxiter = k), k = 1 .. 24: intercept = set-up + epilogue, slope = an early (pivoted) iteration."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
e = Engine()
cfg = workloads.config2(n, seed=5678)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
P = torch.from_numpy(cfg["walkers"]).cuda()
o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
ks = (1, 2, 4, 8, 12, 16, 24)
t = []
for k in ks:
    e.set_iteration_limits(k, k)
    e.time_lnprob_torch(P, *o, reps=2)
    t.append(np.median([e.time_lnprob_torch(P, *o, reps=1) for _ in range(7)]))
    print("exactly %2d iterations: %.3f ms (niter mean %.2f)" % (k, t[-1], float(o[2].double().mean())), flush=True)
e.set_iteration_limits(10, 200)
full = np.median([e.time_lnprob_torch(P, *o, reps=1) for _ in range(5)])
nit = float(o[2].double().mean())
a = np.polyfit(ks[:5], t[:5], 1)
print("k <= 12: %.4f ms per iteration, intercept %.3f ms; k 16 -> 24: %.4f ms per iteration" % (a[0], a[1], (t[-1] - t[-2]) / 8))
print("full run: %.3f ms, %.1f iterations per walker: intercept = %.0f %% of it" % (full, nit, 100 * a[1] / full))
