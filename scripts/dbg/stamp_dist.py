"""Stamped build (-DRX_STAMPS -DRX_STAMP_IT=n -DRX_STAMP_MASK=0x77ull): the segments of iteration n of every walker that never
converges, one line per walker (solve section = slot 2 -> 4: the refinement's loads and corrections, or the elimination)."""
import os, sys
os.environ["RADEX_EMCEE_AMD_LIB"] = os.path.abspath(sys.argv[1]); os.environ["RX_STAMP_FILE"] = "/tmp/stamps.bin"
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
N = 1024
cfg = workloads.config2(N); e = Engine(); e.set_source(cfg["tbg"]); W = cfg["walkers"]; n = 10 ** W[:, 0]
if os.environ.get("RX_REFINE") == "0": e.set_refinement(False)
r = e.solve_batch(10 ** W[:, 1], 10 ** W[:, 2], np.stack([0.25 * n, 0.75 * n], 1))
d = np.fromfile("/tmp/stamps.bin").reshape(-1, 64)[:N]
slow = np.flatnonzero(np.asarray(r["niter"]) >= 200)
print("walker: phase A | A->solve | solve section | Tex/tau | tail | whole iteration")
for w in slow:
    s = d[w]
    if not np.all(np.isfinite(s[[0, 1, 2, 4, 5, 6]])):
        continue
    print("%4d: %5d %4d %6d %5d %5d | %6d" % (w, s[1] - s[0], s[2] - s[1], s[4] - s[2], s[5] - s[4], s[6] - s[5], s[6] - s[0]))
