"""Which phase of an iteration slows down when a second never-converging walker runs on the same CU?
-DRX_STAMPS build (all stamps, iteration RX_STAMP_IT); every other walker of the launch is invalid (no solve)."""
import os, sys
os.environ["RADEX_EMCEE_AMD_LIB"] = os.path.abspath(sys.argv[1]); os.environ["RX_STAMP_FILE"] = "/tmp/stamps.bin"
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
N = 1024
cfg = workloads.config2(N, seed=1234); e = Engine(); e.set_source(cfg["tbg"]); W = cfg["walkers"]; n = 10 ** W[:, 0]
slow = [741, 854, 143, 158]           # four of the fast class (identical phase times alone)
NL = 41
def run(positions, label):
    tk = np.full(N, 100.0); cd = np.full(N, 1e30); dn = np.tile([1e3, 3e3], (N, 1))
    for k, p in enumerate(positions):
        w = slow[k]; tk[p] = 10 ** W[w, 1]; cd[p] = 10 ** W[w, 2]; dn[p] = [0.25 * n[w], 0.75 * n[w]]
    for rep in range(2):
        r = e.solve_batch(tk, cd, dn)
    d = np.fromfile("/tmp/stamps.bin").reshape(-1, 64)[:N]
    hw = d[:, 59].astype(np.int64); xcc = d[:, 60].astype(np.int64) & 15
    for p in positions:
        s = d[p]
        ph = [s[k + 1] - s[k] for k in range(0, 6)]
        print("%-28s pos %3d xcc %d se %d cu %2d simd %d niter %3d: A %.0f B %.0f C %.0f solve %.0f D %.0f E %.0f | first 4 steps %s"
              % (label, p, xcc[p], (hw[p] >> 13) & 7, (hw[p] >> 8) & 15, (hw[p] >> 4) & 3, r["niter"][p], *ph,
                 " ".join("%.0f" % (s[9 + k] - s[8 + k]) for k in range(4))))
run([0], "alone")
run([0, 1], "two, positions 0,1")
run([0, 1, 2, 3], "four, positions 0-3")
run([0, 32], "two, positions 0,32")
