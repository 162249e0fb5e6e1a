"""CPU gate for the ACCEPTANCE RULE of the refinement: what it does to the small level populations.

    python scripts/refine_componentwise.py [N=4096] [--seed 1234] [--tbg T]

The refinement (rx_refine.hip.inc; restated in oracle/radex_oracle.c: rf_solve) accepted an iterate on an ABSOLUTE bound of
the correction (2^-43; populations sum to 1), so levels with populations below ~1e-13 lost their relative accuracy.  This
script runs the reference's arithmetic (pivoted solve every iteration) and the variants on the same walkers and prints, per
variant, the componentwise relative deviation of xpop / T_ex / tau of the walkers that converge with the same iteration
count, next to what the rule costs (corrections per attempt, failed attempts, modelled cost).
Test infrastructure: oracle against oracle, nothing here touches the product.
"""
import argparse
import sys

sys.path.insert(0, ".")
import numpy as np                                   # noqa: E402

from oracle import oracle as O                       # noqa: E402
from radex_emcee_amd import workloads                # noqa: E402
from radex_emcee_amd.molecule import default_molfile  # noqa: E402
from scripts.refine_gate import cost_model           # noqa: E402


def solve_all(mol, tbg, W):
    out = []
    for p in W:
        n = 10.0 ** p[0]
        out.append(O.solve_state(mol, tbg, {3: 0.75 * n, 2: 0.25 * n}, 10.0 ** p[1], 10.0 ** p[2]))
    return out


def dev(a, b, floor=0.0):
    with np.errstate(all="ignore"):
        d = np.abs(a - b) / np.maximum(np.abs(b), floor)
    d = d[np.isfinite(d)]
    return float(d.max()) if len(d) else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("N", nargs="?", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--tbg", type=float, default=None)
    ap.add_argument("--variants", default="")
    a = ap.parse_args()
    mol = O.Molecule(default_molfile("co"))
    cfg = workloads.config2(a.N, seed=a.seed)
    tbg = a.tbg if a.tbg is not None else cfg["tbg"]
    W = cfg["walkers"]
    O.set_refine(0)
    base = solve_all(mol, tbg, W)
    nit0 = np.array([b["niter"] for b in base])
    print("# %d walkers, seed %d, tbg %.4f: mean niter %.2f, %d at maxiter" % (a.N, a.seed, tbg, nit0.mean(), (nit0 >= 200).sum()))
    # (first, tol, max, lag, crit, d1max, loose, backoff, cw_rel, cw_floor, cw_loose)
    T, L = 2.0 ** -40, 2.0 ** -33
    variants = [("absolute 2^-43 (round 5)", (12, T, 8, 2, 1, 2.0 ** 13, L, 1), None)]
    for rel, floor, lrel, ms in ((2.0 ** -20, 1e-20, 2.0 ** -13, 8), (2.0 ** -26, 1e-20, 2.0 ** -19, 8), (2.0 ** -30, 1e-20, 2.0 ** -23, 8),
                                 (2.0 ** -36, 1e-20, 2.0 ** -29, 8), (2.0 ** -43, 1e-20, 2.0 ** -36, 8),
                                 (2.0 ** -30, 1e-24, 2.0 ** -23, 8), (2.0 ** -30, 1e-20, 2.0 ** -23, 12)):
        variants.append(("componentwise rel 2^%d floor %.0e loose 2^%d max %d" % (np.log2(rel), floor, np.log2(lrel), ms),
                         (12, T, ms, 2, 2, 2.0 ** 13, L, 1), (rel, floor, lrel)))
    for name, rf, cw in variants:
        O.set_refine(*rf)
        if cw:
            O.set_refine_componentwise(*cw)
        O.refine_counters(reset=True)
        var = solve_all(mol, tbg, W)
        cnt = O.refine_counters(reset=True)
        O.set_refine(0)
        nit1 = np.array([v["niter"] for v in var])
        same = (nit0 == nit1) & (nit0 < 200)
        dx = max(dev(v["xpop"], b["xpop"]) for v, b, s in zip(var, base, same) if s)
        dx13 = max(dev(v["xpop"][b["xpop"] > 1e-13], b["xpop"][b["xpop"] > 1e-13]) for v, b, s in zip(var, base, same) if s)
        dt = max(dev(v["tex"], b["tex"]) for v, b, s in zip(var, base, same) if s)
        dtau = max(dev(v["tau"], b["tau"]) for v, b, s in zip(var, base, same) if s)
        nbad = sum(dev(v["xpop"], b["xpop"]) > 1e-4 for v, b, s in zip(var, base, same) if s)
        # walkers that stop at maxiter: the populated levels only (the iteration is chaotic there)
        mx = (nit0 >= 200) & (nit1 >= 200)
        dmx = max([dev(v["xpop"], b["xpop"], 1e-6) for v, b, s in zip(var, base, mx) if s] or [0.0])
        att = max(cnt["refined"] + cnt["failed"], 1)
        print("%-58s niter equal %d/%d | converged, same niter: xpop %.1e (levels > 1e-13: %.1e; walkers > 1e-4: %d) tex %.1e tau %.1e |"
              " maxiter xpop(>1e-6) %.1e | replaced %.1f %% steps %.2f failed %.1f %% cost %.3f"
              % (name, (nit0 == nit1).sum(), a.N, dx, dx13, nbad, dt, dtau, dmx,
                 100 * cnt["refined"] / max(cnt["full"] + cnt["refined"], 1), cnt["steps"] / att, 100 * cnt["failed"] / att,
                 cost_model(cnt)), flush=True)


if __name__ == "__main__":
    main()
