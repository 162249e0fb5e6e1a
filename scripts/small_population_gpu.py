"""Level populations, T_ex and tau of the device kernels against the reference arithmetic, COMPONENTWISE, per decade of population.

    python scripts/small_population_gpu.py [N=4096]        (GPU; the oracle is the checker)

For N prior-box walkers (BASELINE config 2's draw, T_bg = 9.56 K) and N walkers of the wider box of the routine tests at
T_bg = 2.73 K: rx_solve_batch with the refinement ON (the default) and OFF (a pivoted elimination in every iteration) against
oracle/radex_oracle.c (= radex.so's arithmetic), over the walkers that converge with the same iteration count.  Per decade of
the reference population: the largest relative deviation and the largest ABSOLUTE one.  What the reference's own solve is worth
in each decade (against the exact solution of its own linear system) is in profiles/r6_small_population_accuracy.txt.
"""
import sys

sys.path.insert(0, ".")
import numpy as np                                   # noqa: E402

from oracle import oracle as O                       # noqa: E402
from radex_emcee_amd import workloads                # noqa: E402
from radex_emcee_amd.engine import Engine            # noqa: E402

EDGES = [0.0, 1e-19, 1e-17, 1e-15, 1e-13, 1e-11, 1e-9, 1e-6, 1e-3, 2.0]


def oracle_all(mol, tbg, tkin, cd, dens):
    out = []
    for w in range(len(tkin)):
        out.append(O.solve_state(mol, tbg, {2: dens[w, 0], 3: dens[w, 1]}, tkin[w], cd[w]))
    return out


def table(tag, got, ref, iupp, ilow):
    same = np.array([r["niter"] for r in ref]) == got["niter"]
    conv = same & (got["niter"] < 200)
    X = np.array([r["xpop"] for r in ref])[conv]
    G = got["xpop"][conv]
    print("# %s: %d walkers, %d converge with the reference's iteration count" % (tag, len(ref), conv.sum()))
    print("%-22s %9s | %-12s %-12s" % ("reference population", "levels", "max rel dev", "max abs dev"))
    for k in range(len(EDGES) - 1):
        m = (X > EDGES[k]) & (X <= EDGES[k + 1])
        if m.any():
            d = np.abs(G[m] - X[m])
            print("%8.0e .. %-8.0e %9d | %-12.2e %-12.2e" % (EDGES[k], EDGES[k + 1], m.sum(), (d / X[m]).max(), d.max()))
    # lines: T_ex and tau, by the smaller of the two level populations
    T = np.array([r["tex"] for r in ref])[conv]
    TA = np.array([r["tau"] for r in ref])[conv]
    small = np.minimum(X[:, iupp - 1], X[:, ilow - 1])
    with np.errstate(all="ignore"):
        dt = np.abs(got["tex"][conv] - T) / np.abs(T)
        dta = np.abs(got["tau"][conv] - TA) / np.abs(TA)
    print("%-22s %9s | %-12s %-12s" % ("smaller level of line", "lines", "max rel T_ex", "max rel tau"))
    for k in range(len(EDGES) - 1):
        m = (small > EDGES[k]) & (small <= EDGES[k + 1])
        if m.any():
            print("%8.0e .. %-8.0e %9d | %-12.2e %-12.2e" % (EDGES[k], EDGES[k + 1], m.sum(), np.nanmax(dt[m]), np.nanmax(dta[m])))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    eng = Engine(device=0)
    mol = O.Molecule(eng.molfile)
    cfg = workloads.config2(N, seed=1234)
    W = cfg["walkers"]
    rng = np.random.default_rng(77)
    sets = [("config 2 prior box, T_bg %.2f K" % cfg["tbg"], cfg["tbg"], 10.0 ** W[:, 1], 10.0 ** W[:, 2],
             np.stack([0.25 * 10.0 ** W[:, 0], 0.75 * 10.0 ** W[:, 0]], axis=1)),
            ("wide box, T_bg 2.73 K", 2.73, 10.0 ** rng.uniform(0.6, 2.9, N), 10.0 ** rng.uniform(12.0, 18.5, N),
             10.0 ** rng.uniform(1.5, 6.5, (N, 2)))]
    assert eng.partner_ids == [2, 3]
    for tag, tbg, tkin, cd, dens in sets:
        eng.set_source(tbg)
        ref = oracle_all(mol, tbg, tkin, cd, dens)
        for on in (True, False):
            eng.set_refinement(on)
            got = eng.solve_batch(tkin, cd, dens)
            table("%s, refinement %s" % (tag, "ON" if on else "OFF"), got, ref, eng.iupp, eng.ilow)
        eng.set_refinement(True)


if __name__ == "__main__":
    main()
