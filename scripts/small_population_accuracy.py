"""How accurate are the SMALL level populations -- in the reference's own solve, and in the refinement?

    python scripts/small_population_accuracy.py [N=256] [--seed 1234] [--tbg T]   (CPU only; needs mpmath)

Round 5's advisor measured that with the refinement (rx_refine.hip.inc) level populations below ~1e-13 differ from the
reference arithmetic's by more than 1e-4 relative (by 100 % at 1e-18).  This script asks the question behind that number:
what are those populations in the reference?  For N prior-box walkers (BASELINE config 2's draw) it takes the linear system
of the LAST iteration of the reference arithmetic (oracle/radex_oracle.c, pinned bit for bit to radex.so) and solves it
  (a) as the reference does: LINPACK sgefa/sgesl in double (the oracle's own right-hand side),
  (b) as the refinement does: start vector and single-precision inverse of two iterations back, corrections until every
      component is below 2^-43 (or two in a row below 2^-36), at most 8,
  (c) exactly: LU in 60-digit arithmetic (mpmath) on the same double-precision matrix,
and prints, per decade of the exact population, the relative error of (a) and of (b) against (c) and the relative
difference (b) - (a) -- the quantity the advisor measured.
Test infrastructure: nothing here touches the product.
"""
import argparse
import sys

sys.path.insert(0, ".")
import numpy as np                                   # noqa: E402
import mpmath as mp                                  # noqa: E402

from oracle import oracle as O                       # noqa: E402
from radex_emcee_amd import workloads                # noqa: E402
from radex_emcee_amd.molecule import default_molfile  # noqa: E402

THR, LOOSE, MAXSTEPS = 2.0 ** -43, 2.0 ** -36, 8


def history(mol, tbg, p, keep=3):
    """(A, x) of the last `keep` iterations of a cold-start run_radex in the reference's arithmetic; niter."""
    n = 10.0 ** p[0]
    st = O.State(mol)
    st.set_density({3: 0.75 * n, 2: 0.25 * n})
    st.s.tkin, st.s.cdmol, st.s.totdens = 10.0 ** p[1], 10.0 ** p[2], n
    st.rates()
    st.backrad(tbg)
    N = mol.nlev
    out = []
    for it in range(200):
        conv = st.matrix(it)
        A = st.arr("yrate").reshape(N, N).T.copy()
        A[N - 1, :] = 1.0                                   # lubksb_: the last balance equation <- sum x = 1
        out.append((A, st.arr("rhs").copy()))
        out = out[-keep:]
        if conv and it >= 10:
            return out, it + 1
    return out, 200


def refine(A, A_kept, x_kept):
    """The device rule (rf_refine) in numpy: float inverse, float correction, double residual."""
    Mf = np.linalg.inv(A_kept).astype(np.float32)
    b = np.zeros(len(x_kept))
    b[-1] = 1.0
    x = x_kept.copy()
    lprev = True
    for st in range(MAXSTEPS):
        r = A @ x - b
        d = (Mf @ r.astype(np.float32)).astype(np.float64)
        x = x - d
        if np.all(np.abs(d) < THR):
            return x, st + 1, True
        bigl = not np.all(np.abs(d) < LOOSE)
        if not bigl and not lprev:
            return x, st + 1, True
        lprev = bigl
    return x, MAXSTEPS, False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("N", nargs="?", type=int, default=256)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--tbg", type=float, default=None)
    a = ap.parse_args()
    mp.mp.dps = 60
    mol = O.Molecule(default_molfile("co"))
    cfg = workloads.config2(a.N, seed=a.seed)
    tbg = a.tbg if a.tbg is not None else cfg["tbg"]
    O.set_refine(0)
    edges = [1e-21, 1e-19, 1e-17, 1e-15, 1e-13, 1e-11, 1e-9, 1e-6, 1e-3, 2.0]
    acc = {k: [[] for _ in edges[:-1]] for k in ("ref", "rf", "diff")}
    used = refused = 0
    for p in cfg["walkers"]:
        h, nit = history(mol, tbg, p)
        if nit >= 200 or len(h) < 3:
            continue
        (A2, x2), _, (A0, x_ref) = h
        try:
            x_rf, steps, ok = refine(A0, A2, x2)
        except np.linalg.LinAlgError:                       # (a NaN / singular system: the maser branch)
            continue
        if not ok or not np.all(np.isfinite(x_ref)):
            refused += 1
            continue
        used += 1
        b = [0] * (mol.nlev - 1) + [1]
        x_true = np.array([float(v) for v in mp.lu_solve(mp.matrix(A0.tolist()), mp.matrix(b))])
        for i in range(mol.nlev):
            k = np.searchsorted(edges, abs(x_true[i])) - 1
            if 0 <= k < len(edges) - 1:
                acc["ref"][k].append(abs(x_ref[i] - x_true[i]) / abs(x_true[i]))
                acc["rf"][k].append(abs(x_rf[i] - x_true[i]) / abs(x_true[i]))
                acc["diff"][k].append(abs(x_rf[i] - x_ref[i]) / abs(x_ref[i]))
    print("# %d walkers of config2(seed %d), tbg %.4f K: %d converged and refined at their last iteration (%d attempts given up)"
          % (a.N, a.seed, tbg, used, refused))
    print("# relative error of a level population against the 60-digit solution of the same double-precision system")
    print("%-22s %8s | %-23s | %-23s | %-23s" % ("exact population", "levels", "reference LU: med / max", "refinement: med / max",
                                                   "refinement - reference"))
    for k in range(len(edges) - 1):
        if not acc["ref"][k]:
            continue
        r, f, d = (np.array(acc[n][k]) for n in ("ref", "rf", "diff"))
        print("%8.0e .. %-8.0e %8d | %10.1e / %-10.1e | %10.1e / %-10.1e | %10.1e / %-10.1e"
              % (edges[k], edges[k + 1], len(r), np.median(r), r.max(), np.median(f), f.max(), np.median(d), d.max()))


if __name__ == "__main__":
    main()
