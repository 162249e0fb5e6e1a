"""Kernel time of alternative builds on the bench workloads: 1024 config-2 walkers (latency regime, one
wavefront per SIMD) and 32768 (throughput regime, two per SIMD).  Usage: python scripts/ablate.py lib.so ..."""
import os, subprocess, sys
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
e = Engine()
out = []
for n, seed in ((1024, 1234), (32768, 5678)):
    cfg = workloads.config2(n, seed=seed)
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    P = torch.from_numpy(cfg["walkers"]).cuda()
    lnp = torch.empty(n, dtype=torch.float64, device="cuda"); st = torch.empty(n, dtype=torch.int32, device="cuda"); nit = torch.empty_like(st)
    e.time_lnprob_torch(P, lnp, st, nit, reps=2)
    ms = e.time_lnprob_torch(P, lnp, st, nit, reps=10)
    out.append("%d: %.4f ms (%.0f k/s) chk=%.9e nit=%d" % (n, ms, n / ms, float(lnp[torch.isfinite(lnp)].sum()), int(nit.sum())))
print("RESULT " + " | ".join(out))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, RADEX_EMCEE_AMD_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print("%-20s %s" % (os.path.basename(lib), line[0][7:] if line else r.stderr[-300:]))
