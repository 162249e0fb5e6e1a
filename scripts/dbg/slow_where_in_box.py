"""Which prior-box walkers run into maxiter: parameter ranges (config 2, 65536 draws)."""
import sys; sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
n = 65536
cfg = workloads.config2(n, seed=777)
e = Engine(); e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
lnp, st, nit = e.lnprob_batch(cfg["walkers"], return_info=True)
W = cfg["walkers"]
print("bounds", cfg["bounds"].tolist())
slow = st == 1
print("maxiter: %d of %d (%.2f %%)" % (slow.sum(), n, 100 * slow.mean()))
names = ["log n", "log T", "log N", "log size"]
for k in range(4):
    q = np.quantile(W[slow, k], [0, .01, .05, .5, .95, .99, 1])
    print("%-8s slow quantiles 0/1/5/50/95/99/100: %s" % (names[k], np.round(q, 2)))
d = W[:, 2] - W[:, 0]
print("log N - log n: slow quantiles", np.round(np.quantile(d[slow], [0, .01, .05, .5, .95, .99, 1]), 2))
# 2-D histogram of the slow fraction in (log T, log N)
for lo_t in np.arange(0.5, 3.01, 0.25):
    row = []
    for lo_n in np.arange(12, 19.6, 0.5):
        m = (W[:, 1] >= lo_t) & (W[:, 1] < lo_t + 0.25) & (W[:, 2] >= lo_n) & (W[:, 2] < lo_n + 0.5)
        row.append("%3d" % (100 * slow[m].mean()) if m.sum() > 20 else "  .")
    print("logT %.2f: %s" % (lo_t, " ".join(row)))
print("(columns: log N from 12 in steps of 0.5; entries: %% of walkers that reach maxiter)")
# niter classes
for lo_t in np.arange(0.5, 3.01, 0.25):
    row = []
    for lo_d in np.arange(1.5, 7.1, 0.5):
        m = (W[:, 1] >= lo_t) & (W[:, 1] < lo_t + 0.25) & (W[:, 0] >= lo_d) & (W[:, 0] < lo_d + 0.5)
        row.append("%3d" % (100 * slow[m].mean()) if m.sum() > 20 else "  .")
    print("logT %.2f: %s" % (lo_t, " ".join(row)))
print("(columns: log n from 1.5 in steps of 0.5)")
