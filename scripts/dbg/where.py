"""Where the never-converging walkers of the 1024-walker launch run (XCC, SE, CU, SIMD from HW_ID / XCC_ID,
-DRX_STAMPS build) and how long their iterations take, against the number of other such walkers on the same CU,
the neighbouring CU, the same shader engine and the same XCD.  python scripts/dbg/where.py scripts/abl/stamps.so"""
import os, sys
os.environ["RADEX_EMCEE_AMD_LIB"] = os.path.abspath(sys.argv[1]); os.environ["RX_STAMP_FILE"] = "/tmp/stamps.bin"
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
N = 1024
cfg = workloads.config2(N, seed=1234); e = Engine(); e.set_source(cfg["tbg"]); W = cfg["walkers"]; n = 10 ** W[:, 0]
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    r = e.solve_batch(10 ** W[:, 1], 10 ** W[:, 2], np.stack([0.25 * n, 0.75 * n], 1))
    d = np.fromfile("/tmp/stamps.bin").reshape(-1, 64)[:N]
    hw = d[:, 59].astype(np.int64); xcc = d[:, 60].astype(np.int64) & 15
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    nit = np.asarray(r["niter"])
    slow = np.where(nit >= 200)[0]
    per_it = (d[:, 58] - d[:, 55]) / np.maximum(nit, 1)
    key_cu = xcc * 1000 + se * 100 + sh * 50 + cu
    key_pair = xcc * 1000 + se * 100 + sh * 50 + cu // 2
    key_se = xcc * 10 + se
    print("launch %d: %d slow walkers; distinct (xcc,se,sh,cu) over all walkers: %d" % (rep, len(slow), len(set(key_cu))))
    for w in slow[np.argsort(per_it[slow])]:
        same_cu = int(np.sum(key_cu[slow] == key_cu[w])) - 1
        same_pair = int(np.sum(key_pair[slow] == key_pair[w])) - 1 - same_cu
        same_se = int(np.sum(key_se[slow] == key_se[w])) - 1
        same_xcc = int(np.sum(xcc[slow] == xcc[w])) - 1
        print("  walker %4d xcc %d se %d sh %d cu %2d simd %d: %.0f ticks/iteration; other slow on the CU %d, on the neighbour CU %d, in the SE %d, in the XCD %d"
              % (w, xcc[w], se[w], sh[w], cu[w], simd[w], per_it[w], same_cu, same_pair, same_se, same_xcc))
