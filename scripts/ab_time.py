"""Careful A/B of builds on one workload: python scripts/ab_time.py N reps a.so b.so [a.so b.so ...]
Each build runs in its own process (the library is chosen at import); per build: median and minimum of `reps`
launches over N config-2 prior-box walkers, after a warm-up.  Alternate the builds on the command line."""
import os, subprocess, sys
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
n, reps = int(sys.argv[1]), int(sys.argv[2])
import os
cfg = workloads.config2(n, seed=int(os.environ.get("RX_AB_SEED", "5678" if n != 1024 else "1234")))
e = Engine(); e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
P = torch.from_numpy(cfg["walkers"]).cuda()
o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
e.time_lnprob_torch(P, *o, reps=10)
ts = sorted(e.time_lnprob_torch(P, *o, reps=1) for _ in range(reps))
print("RESULT median %.4f ms  min %.4f ms  max %.4f ms" % (ts[len(ts) // 2], ts[0], ts[-1]))
'''
n, reps = sys.argv[1], sys.argv[2]
for lib in sys.argv[3:]:
    env = dict(os.environ, RADEX_EMCEE_AMD_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD, n, reps], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print("%-14s N=%s %s" % (os.path.basename(lib)[:-3], n, line[0][7:] if line else r.stderr[-400:]), flush=True)
