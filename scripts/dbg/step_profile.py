"""Per-phase and per-elimination-step ticks of iteration RX_STAMP_IT for chosen walkers of the 1024-walker batch
(-DRX_STAMPS build with every stamp on).  python scripts/dbg/step_profile.py lib.so 741,724,..."""
import os, sys
os.environ["RADEX_EMCEE_AMD_LIB"] = os.path.abspath(sys.argv[1]); os.environ["RX_STAMP_FILE"] = "/tmp/stamps.bin"
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
N = 1024
cfg = workloads.config2(N, seed=1234); e = Engine(); e.set_source(cfg["tbg"]); W = cfg["walkers"]; n = 10 ** W[:, 0]
r = e.solve_batch(10 ** W[:, 1], 10 ** W[:, 2], np.stack([0.25 * n, 0.75 * n], 1))
d = np.fromfile("/tmp/stamps.bin").reshape(-1, 64)[:N]
NL = 41
for w in [int(x) for x in sys.argv[2].split(",")]:
    s = d[w]
    ph = [s[k + 1] - s[k] for k in range(0, 6)]
    steps = [s[8 + k + 1] - s[8 + k] for k in range(0, NL)]      # 8+k .. 8+NL, then 9+NL
    print("walker %d (niter %d): phases 0-6: %s" % (w, r["niter"][w], " ".join("%.0f" % x for x in ph)))
    print("   steps: %s" % " ".join("%.0f" % x for x in steps))
    print("   solve total %.0f, iteration total %.0f" % (s[9 + NL] - s[8], s[6] - s[0]))
