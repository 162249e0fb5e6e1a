import os, sys
os.environ["RADEX_EMCEE_AMD_LIB"] = os.path.abspath(sys.argv[1]); os.environ["RX_STAMP_FILE"] = "/tmp/stamps.bin"
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
N = int(os.environ.get("RX_STAMP_N", "1024"))        # fewer walkers = fewer wavefronts starting in lock step
cfg = workloads.config2(N); e = Engine(); e.set_source(cfg["tbg"]); W = cfg["walkers"]; n = 10 ** W[:, 0]
r = e.solve_batch(10 ** W[:, 1], 10 ** W[:, 2], np.stack([0.25 * n, 0.75 * n], 1))
d = np.fromfile("/tmp/stamps.bin").reshape(-1, 64)[:N]
slots = [int(x) for x in sys.argv[2].split(",")]
ok = np.all(np.isfinite(d[:, slots]), axis=1) & (d[:, slots[-1]] > d[:, slots[0]])
if len(sys.argv) > 3 and sys.argv[3] == "slow":       # only the walkers that run into maxiter (they set the launch time)
    ok &= np.asarray(r["niter"]) >= 200
print("%d of %d walkers carry stamps" % (ok.sum(), len(d)))
for a, b in zip(slots[:-1], slots[1:]):
    print("slot %d -> %d: median %.0f ticks" % (a, b, np.median(d[ok, b] - d[ok, a])))
