"""Trace of the prior-box sampler for schedule simulations: per proposal the walker, its complement, the
iteration count, the status and whether it was accepted (host-driven half-steps through the numpy mirror of the
device kernels; the chain is the device sampler's).  Writes gpurun_out/sampler_trace.npz."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import stretch_propose, stretch_accept, walker_permutation, philox4x32_10, u53, PURPOSE_PROPOSE
nw, nst, seed = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 170, 7
cfg = workloads.config2(nw)
eng = Engine()
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
X = cfg["walkers"].copy()
lnp = eng.lnprob_batch(X)
h = nw // 2
rec = {k: [] for k in ("ws", "wc", "niter", "status", "acc")}
for step in range(nst):
    for split in range(2):
        q, factor, widx = stretch_propose(X, 1, nw, 2.0, seed, step, split)
        perm = walker_permutation(nw, seed, step, 0)
        r = philox4x32_10(np.arange(h), 0, step, PURPOSE_PROPOSE + 16 * split, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        ri = np.minimum((u53(r[2], r[3]) * float(h)).astype(np.int64), h - 1)
        wc = perm[(1 - split) * h + ri]
        lq, st, nit = eng.lnprob_batch(q, return_info=True)
        before = lnp.copy()
        na = np.zeros(nw, dtype=np.int64)
        stretch_accept(X, lnp, na, q, lq, factor, widx, 1, nw, seed, step, split)
        rec["ws"].append(widx.copy()); rec["wc"].append(wc.astype(np.int32)); rec["niter"].append(nit.copy())
        rec["status"].append(st.copy()); rec["acc"].append(na[widx] > 0)
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/sampler_trace.npz", **{k: np.array(v) for k, v in rec.items()})
a = np.array(rec["acc"]); n = np.array(rec["niter"]); s = np.array(rec["status"])
print("steps %d: acceptance %.3f, niter mean (solved) %.1f, maxiter %.4f, prior %.3f" % (nst, a.mean(), n[s < 2].mean(), (s == 1).mean(), (s == 3).mean()))
