"""GPU against the CPU checker's reference arithmetic on draws of 131 072 prior-box walkers (status, iteration counts, lnprob per tier).
usage: python scripts/big_parity_seeds.py [--norefine] [seed ...]   (--norefine: every solve the pivoted elimination)"""
import sys, time; sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as O
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
eng = Engine(); mol = O.Molecule(eng.molfile)
if "--norefine" in sys.argv:
    sys.argv.remove("--norefine"); eng.set_refinement(False); print("refinement off")
for seed in ([int(x) for x in sys.argv[1:]] or (11, 222, 3333, 44444)):
    cfg = workloads.config2(131072, seed=seed)
    eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
    eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
    src = O.Source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
    lnp, st, nit = eng.lnprob_batch(cfg["walkers"], return_info=True)
    rl, rst, rnit = O.lnprob_batch(mol, src, cfg["walkers"], nthreads=16)
    fin = np.isfinite(rl) & np.isfinite(lnp)
    d = np.abs(lnp - rl) / np.maximum(np.abs(rl), 1.0)
    print("seed %d: status equal %d/%d, niter equal %d, finite mismatch %d, lnprob converged max %.2e, maxiter max %.2e (%d walkers, %d above 1e-4)"
          % (seed, (st == rst).sum(), len(st), (nit == rnit).sum(), (np.isfinite(rl) != np.isfinite(lnp)).sum(),
             d[fin & (rst == 0) & (nit == rnit)].max(), d[fin & (rst == 1)].max(), (rst == 1).sum(), (d[fin & (rst == 1)] > 1e-4).sum()), flush=True)
