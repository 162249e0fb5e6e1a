import os, sys
os.environ["RADEX_EMCEE_AMD_LIB"] = os.path.abspath(sys.argv[1]); os.environ["RX_STAMP_FILE"] = "/tmp/stamps.bin"
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
N = 1024
cfg = workloads.config2(N); e = Engine(); e.set_source(cfg["tbg"]); W = cfg["walkers"]; n = 10 ** W[:, 0]
r = e.solve_batch(10 ** W[:, 1], 10 ** W[:, 2], np.stack([0.25 * n, 0.75 * n], 1))
d = np.fromfile("/tmp/stamps.bin").reshape(-1, 64)[:N]
slow = np.asarray(r["niter"]) >= 200
ok = np.isfinite(d[:, 2]) & np.isfinite(d[:, 4]) & (d[:, 4] > d[:, 2])
x = (d[:, 4] - d[:, 2])[ok & slow]
print("slow walkers: solve section ticks sorted:", np.sort(x).astype(int))
x = (d[:, 4] - d[:, 2])[ok & ~slow]
print("others: percentiles 5 25 50 75 95:", np.percentile(x, [5, 25, 50, 75, 95]).astype(int), len(x))
