"""One-off: GPU vs oracle on a large config-2 draw (iteration counts, status, lnprob)."""
import sys, time; sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as O
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = workloads.config2(N, seed=24680)
eng = Engine(); mol = O.Molecule(eng.molfile)
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
src = O.Source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
lnp, st, nit = eng.lnprob_batch(cfg["walkers"], return_info=True)
t = time.time(); rl, rst, rnit = O.lnprob_batch(mol, src, cfg["walkers"], nthreads=16); print("oracle %.1f s" % (time.time() - t))
print("status equal: %d / %d" % ((st == rst).sum(), N))
print("niter equal:  %d / %d ; max |dniter| %d" % ((nit == rnit).sum(), N, np.abs(nit - rnit).max()))
fin = np.isfinite(rl) & np.isfinite(lnp)
print("finite both: %d ; finite mismatch: %d" % (fin.sum(), (np.isfinite(rl) != np.isfinite(lnp)).sum()))
same = fin & (nit == rnit)
d = np.abs(lnp[same] - rl[same]) / np.maximum(np.abs(rl[same]), 1.0)
print("max rel dev of lnprob (same niter): %.3e ; 99.9th pct %.3e" % (d.max(), np.percentile(d, 99.9)))
for name, code in (("converged (RX_OK)", 0), ("maxiter (RX_MAXITER)", 1)):
    m = same & (rst == code)
    dd = np.abs(lnp[m] - rl[m]) / np.maximum(np.abs(rl[m]), 1.0)
    print("  %-22s %7d walkers: max %.3e ; 99.9th pct %.3e ; above 1e-4: %d ; above 1e-5: %d"
          % (name, m.sum(), dd.max(), np.percentile(dd, 99.9), int((dd > 1e-4).sum()), int((dd > 1e-5).sum())))
diff = fin & (nit != rnit)
if diff.any():
    d2 = np.abs(lnp[diff] - rl[diff]) / np.maximum(np.abs(rl[diff]), 1.0)
    print("walkers with different niter: %d ; their max rel dev of lnprob %.3e ; niter pairs %s" % (diff.sum(), d2.max(), list(zip(nit[diff][:8], rnit[diff][:8]))))
