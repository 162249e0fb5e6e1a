"""Outputs of two builds on the same walkers, compared value by value (which walkers differ, by how much, in what).
usage: python scripts/dbg/cmp_libs.py a.so b.so [N=1024]"""
import os, subprocess, sys
import numpy as np
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
n = int(sys.argv[1]); e = Engine()
cfg = workloads.config2(n, seed=1234)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
lp, st, ni = e.lnprob_batch(cfg["walkers"], return_info=True)
fl = e.model_flux_batch(cfg["walkers"])
np.savez(sys.argv[2], lp=lp, st=st, ni=ni, fl=fl, w=cfg["walkers"])
'''
n = sys.argv[3] if len(sys.argv) > 3 else "1024"
res = []
for k, lib in enumerate(sys.argv[1:3]):
    out = "/tmp/cmp_%d.npz" % k
    r = subprocess.run([sys.executable, "-c", CHILD, n, out], env=dict(os.environ, RADEX_EMCEE_AMD_LIB=os.path.abspath(lib)),
                       capture_output=True, text=True)
    if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
    res.append(np.load(out))
a, b = res
d = ~((a["lp"] == b["lp"]) | (np.isnan(a["lp"]) & np.isnan(b["lp"])))
print("walkers", len(d), "lnprob differs in", d.sum(), " niter differs in", (a["ni"] != b["ni"]).sum(), " status differs in", (a["st"] != b["st"]).sum())
idx = np.nonzero(d | (a["ni"] != b["ni"]))[0]
for i in idx[:12]:
    print(i, a["w"][i], "niter", a["ni"][i], b["ni"][i], "st", a["st"][i], b["st"][i], "lp", a["lp"][i], b["lp"][i])
rel = np.abs(a["fl"] - b["fl"]) / np.maximum(np.abs(a["fl"]), 1e-300)
print("flux: max rel diff", np.nanmax(rel), " walkers with any flux difference", (np.nan_to_num(rel).max(axis=1) > 0).sum())
