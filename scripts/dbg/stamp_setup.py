"""Stamped build (-DRX_STAMPS -DRX_STAMP_MASK=0x7f8000000000000ull: the set-up slots 51..58 only): where a walker's time goes
before its first and after its last iteration, medians over the 1024 walkers of the headline batch (cycles of s_memtime).
usage: python scripts/dbg/stamp_setup.py lib.so [N=1024: the first N walkers of the batch] [maxiter]"""
import os, sys
os.environ["RADEX_EMCEE_AMD_LIB"] = os.path.abspath(sys.argv[1]); os.environ["RX_STAMP_FILE"] = "/tmp/stamps.bin"
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = workloads.config2(1024); e = Engine(); e.set_source(cfg["tbg"]); W = cfg["walkers"][:N]; n = 10 ** W[:, 0]
if len(sys.argv) > 3: e.set_iteration_limits(0, int(sys.argv[3]))        # (e.g. 1: set-up + one iteration + epilogue)
r = e.solve_batch(10 ** W[:, 1], 10 ** W[:, 2], np.stack([0.25 * n, 0.75 * n], 1))
d = np.fromfile("/tmp/stamps.bin").reshape(-1, 64)[:N]
names = {56: "kernel: tables staged, item taken", 57: "in front of solve_wave", 51: "rates: start", 52: "rates: table sums done",
         53: "detailed balance done", 54: "static part of the matrix written", 55: "first iteration starts", 58: "behind solve_wave"}
order = [56, 57, 51, 52, 53, 54, 55, 58]
t0 = d[:, 56].min()
print("earliest walker's slot 56 = 0; medians over %d walkers (cycles), and the step from the previous slot" % N)
prev = None
for s in order:
    v = d[:, s]
    print("  slot %d %-40s median %9.0f  min %9.0f  max %9.0f   step %s" % (s, names[s], np.median(v - t0), (v - t0).min(), (v - t0).max(),
          "" if prev is None else "%8.0f" % np.median(v - d[:, prev])))
    prev = s
nit = np.asarray(r["niter"])
print("iterations: median %d; time 55 -> 58 per iteration, median %.0f cycles" % (np.median(nit), np.median((d[:, 58] - d[:, 55]) / nit)))
# the clock all XCDs share (s_memrealtime, 100 MHz): kernel entry of the walker's workgroup, its first item taken, its results stored
rt = d[:, [61, 62, 63]] * 0.01                                   # us
t00 = rt[:, 0].min()
print("shared clock, us after the first workgroup entered the kernel: workgroup entry median %.1f max %.1f | item taken (tables staged) median %.1f max %.1f | "
      "results stored median %.1f max %.1f" % (np.median(rt[:, 0] - t00), (rt[:, 0] - t00).max(), np.median(rt[:, 1] - t00), (rt[:, 1] - t00).max(),
                                             np.median(rt[:, 2] - t00), (rt[:, 2] - t00).max()))
print("entry -> item taken: median %.1f us, max %.1f us" % (np.median(rt[:, 1] - rt[:, 0]), (rt[:, 1] - rt[:, 0]).max()))
print("item taken -> results stored: median %.1f us, min %.1f, max %.1f; cycles 56 -> 58 (s_memtime): median %.0f -> %.2f GHz if that is the shader clock"
      % (np.median(rt[:, 2] - rt[:, 1]), (rt[:, 2] - rt[:, 1]).min(), (rt[:, 2] - rt[:, 1]).max(), np.median(d[:, 58] - d[:, 56]),
         np.median(d[:, 58] - d[:, 56]) / np.median(rt[:, 2] - rt[:, 1]) * 1e-3))
# the average clock over a walker's life, by how long it lived
ghz = (d[:, 58] - d[:, 56]) / (rt[:, 2] - rt[:, 1]) * 1e-3
for lo, hi in ((0, 20), (20, 30), (30, 50), (50, 100), (100, 199), (200, 201)):
    m = (nit >= lo) & (nit < hi)
    if m.any():
        print("walkers with %3d..%3d iterations: %4d, alive for %6.1f us (median), average clock over that time %.2f GHz" % (lo, hi - 1, m.sum(), np.median((rt[:, 2] - rt[:, 1])[m]), np.median(ghz[m])))
