"""Do the never-converging walkers reach a bitwise period-2 state?  For each, the first iteration limit k at which
the results with limits k and k+2 are bit-identical (and stay so), from launches with maxiter = 12, 14, ..., 200."""
import sys; sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = workloads.config2(N, seed=1234); e = Engine(); e.set_source(cfg["tbg"]); W = cfg["walkers"]; n = 10 ** W[:, 0]
tk, cd, dn = 10 ** W[:, 1], 10 ** W[:, 2], np.stack([0.25 * n, 0.75 * n], 1)
r = e.solve_batch(tk, cd, dn)
slow = np.where(np.asarray(r["niter"]) >= 200)[0]
print("%d walkers, %d reach maxiter" % (N, len(slow)))
tk, cd, dn = tk[slow], cd[slow], dn[slow]
outs = {}
for k in list(range(11, 202)):
    e.set_iteration_limits(10, k)
    r = e.solve_batch(tk, cd, dn)
    outs[k] = np.concatenate([r["xpop"], r["tex"], r["tau"]], axis=1).view(np.uint64)
e.set_iteration_limits(10, 200)
first = []
for i in range(len(slow)):
    same = {k: np.array_equal(outs[k][i], outs[k + 2][i]) for k in range(11, 199)}
    # first k from which on every later comparison is equal
    ks = [k for k in range(11, 199) if all(same[j] for j in range(k, 199))]
    first.append(ks[0] if ks else -1)
first = np.array(first)
print("bitwise period 2 reached (limit k == limit k+2 from then on): %d of %d walkers" % ((first > 0).sum(), len(slow)))
if (first > 0).any():
    print("onset iteration quantiles 0/25/50/75/100:", np.quantile(first[first > 0], [0, .25, .5, .75, 1]))
print("never:", slow[first < 0][:20], "...")
# period 1 (a fixed point that the convergence test does not accept)?  and longer periods
for p in (1, 3, 4, 6, 8):
    cnt = 0
    for i in np.where(first < 0)[0]:
        if all(np.array_equal(outs[k][i], outs[k + p][i]) for k in range(190, 200 - p + 1)): cnt += 1
    print("of the rest, period %d at the end: %d" % (p, cnt))
