"""Diagnostic: per-phase cycle stamps of one iteration (needs scripts/abl/stamps.so built with -DRX_STAMPS)."""
import os, sys
os.environ["RADEX_EMCEE_AMD_LIB"] = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "scripts/abl/stamps.so")
os.environ["RX_STAMP_FILE"] = "/tmp/stamps.bin"
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
cfg = workloads.config2(1024)
e = Engine()
e.set_source(cfg["tbg"])
W = cfg["walkers"]
n = 10 ** W[:, 0]
r = e.solve_batch(10 ** W[:, 1], 10 ** W[:, 2], np.stack([0.25 * n, 0.75 * n], 1))
d = np.fromfile("/tmp/stamps.bin").reshape(1024, 64)
NL = 41
names = {0: "A", 1: "B", 2: "C", 3: "LU-pre", 4: "post-LU", 5: "conv", 6: "end"}
dt = np.diff(d[:, :7], axis=1)
print("phase cycles (median over walkers): A %.0f  B %.0f  C %.0f  LU(total) %.0f  norm+D %.0f  conv+relax %.0f" % tuple(np.median(dt, axis=0)))
lu = d[:, 8:8 + NL + 2]
st = np.diff(lu, axis=1)
med = np.median(st, axis=0)
print("LU prologue (stamp3 -> step0): %.0f" % np.median(d[:, 8] - d[:, 3]))
print("LU step cycles k=0..40:", " ".join("%.0f" % x for x in med[:NL]))
print("back-substitution: %.0f ; scatter: %.0f" % (med[NL], np.median(d[:, 4] - d[:, 9 + NL])))
print("iteration total: median %.0f  p10 %.0f p90 %.0f" % (np.median(d[:, 6] - d[:, 0]), *np.percentile(d[:, 6] - d[:, 0], [10, 90])))

sub = d[:, 52:57] - d[:, 8 + 5][:, None]
print("step 5 sub-stamps (cycles from step start): bookkeeping %.0f | after 6 DPP stages+groups %.0f | p known %.0f | rcp issued %.0f | NR done %.0f | step end %.0f" % (*np.median(sub, axis=0), np.median(d[:, 8 + 6] - d[:, 8 + 5])))
