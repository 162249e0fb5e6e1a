"""How often the elimination's out-of-line pivot path (`settle`) runs: a -DRX_STAMPS build counts, per walker, the steps that took it
(slot 61) and the solves with at least one such step (slot 62).  usage: python scripts/dbg/settle_count.py lib_stamps.so [N=1024]"""
import os, sys
os.environ["RADEX_EMCEE_AMD_LIB"] = os.path.abspath(sys.argv[1]); os.environ["RX_STAMP_FILE"] = "/tmp/stamps.bin"
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = workloads.config2(N); e = Engine(); e.set_source(cfg["tbg"]); W = cfg["walkers"]; n = 10 ** W[:, 0]
r = e.solve_batch(10 ** W[:, 1], 10 ** W[:, 2], np.stack([0.25 * n, 0.75 * n], 1))
d = np.fromfile("/tmp/stamps.bin").reshape(-1, 64)[:N]
nit = np.asarray(r["niter"]); steps, solves = d[:, 61], d[:, 62]
ok = nit > 0
print("walkers %d, solves %d, elimination steps %d" % (ok.sum(), nit[ok].sum(), 41 * nit[ok].sum()))
print("steps through settle: %d (%.4f %% of the steps); solves with one: %d (%.2f %% of the solves)"
      % (steps[ok].sum(), 100 * steps[ok].sum() / (41.0 * nit[ok].sum()), solves[ok].sum(), 100 * solves[ok].sum() / nit[ok].sum()))
slow = ok & (nit >= 200)
print("the %d walkers that run into maxiter: solves with a settle %.2f %% (per walker: min %d, median %d, max %d of 200)"
      % (slow.sum(), 100 * solves[slow].sum() / nit[slow].sum(), solves[slow].min(), np.median(solves[slow]), solves[slow].max()))
first = solves[ok] > 0
print("walkers with at least one: %d; solves-with-settle per walker: median %d, 90th pct %d, max %d" % (first.sum(), np.median(solves[ok]), np.percentile(solves[ok], 90), solves[ok].max()))
