"""Kernel time of alternative builds on the bench workload.  Usage: python scripts/ablate.py scripts/abl/*.so"""
import os, subprocess, sys
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
cfg = workloads.config2(1024)
e = Engine()
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
P = torch.from_numpy(cfg["walkers"]).cuda()
lnp = torch.empty(1024, dtype=torch.float64, device="cuda"); st = torch.empty(1024, dtype=torch.int32, device="cuda"); nit = torch.empty_like(st)
e.time_lnprob_torch(P, lnp, st, nit, reps=3)
ms = e.time_lnprob_torch(P, lnp, st, nit, reps=20)
print("RESULT %.4f ms  sum(lnp finite)=%.6e niter_sum=%d" % (ms, float(lnp[torch.isfinite(lnp)].sum()), int(nit.sum())))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, RADEX_EMCEE_AMD_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print("%-28s %s" % (os.path.basename(lib), line[0] if line else r.stderr[-300:]))
