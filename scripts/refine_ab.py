"""GPU: the refinement of the kept solution (rx_set_refinement) against the pivoted solve every iteration, same binary.

    python scripts/refine_ab.py [N=1024] [seed=1234]

Status / iteration counts / lnprob of the two against each other and against the oracle, the refinement's counters, and the
time of a launch (median of 20) either way.
"""
import sys, time; sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as O
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
cfg = workloads.config2(N, seed=seed)
eng = Engine(); mol = O.Molecule(eng.molfile)
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
eng.set_refinement(False)
tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
src = O.Source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
rl, rst, rnit = O.lnprob_batch(mol, src, cfg["walkers"], nthreads=16)


def rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0)


res = {}
for on in (False, True):
    eng.set_refinement(on)
    eng.set_refinement_counting(True)
    eng.refinement_counters(reset=True)
    lnp, st, nit = eng.lnprob_batch(cfg["walkers"], return_info=True)
    cnt = eng.refinement_counters(reset=True)
    eng.set_refinement_counting(False)
    ts = []
    for _ in range(20):
        t = time.perf_counter(); eng.lnprob_batch(cfg["walkers"]); ts.append(time.perf_counter() - t)
    res[on] = (lnp, st, nit)
    fin = np.isfinite(rl) & np.isfinite(lnp)
    print("refinement %-3s: status == oracle %d/%d, niter == oracle %d/%d, lnprob vs oracle: converged %.2e maxiter %.2e | "
          "host-timed call median %.3f ms | counters %s" % (
              "on" if on else "off", (st == rst).sum(), N, (nit == rnit).sum(), N,
              rel(lnp, rl)[fin & (rst == 0)].max(), rel(lnp, rl)[fin & (rst == 1)].max() if (fin & (rst == 1)).any() else 0.0,
              1e3 * np.median(ts), cnt))
a, b = res[False], res[True]
fin = np.isfinite(a[0]) & np.isfinite(b[0])
print("on vs off: status equal %d/%d, niter equal %d/%d, lnprob %.2e" % ((a[1] == b[1]).sum(), N, (a[2] == b[2]).sum(), N,
                                                                          rel(b[0], a[0])[fin].max()))
