"""Are the GPU's `maxiter` outliers the walkers the REFERENCE BINARY itself is most sensitive on?   (GPU; fixture from the container)

    python scripts/maxiter_vs_ref_sensitivity.py [--norefine]

tests/golden/ref_sensitivity.npz (scripts/ref_sensitivity.py) holds, for every walker of sixteen 131 072-walker prior-box draws
that radex.so runs to maxiter = 200, how far the binary's own answer moves when its exp / log are one ulp off.  This script
evaluates the same walkers on the GPU against the CPU checker's reference arithmetic (= the binary's, bit for bit) and prints
  * per draw: the GPU's deviation and the binary's 1-ulp response, both in units of the flux tolerance
    (1e-4 |F| + 1e-10 F_bg: tests/test_gpu_parity.py:_flux_ok) and as relative lnprob;
  * for every walker the GPU has beyond the tolerance: where that walker ranks among the binary's responses of its draw;
  * the smallest K with  GPU deviation <= K x (binary's 1-ulp response) + tolerance  on every walker -- the bound
    tests/test_gpu_round2.py asserts for its own batches.
"""
import os
import sys

sys.path.insert(0, ".")
import numpy as np                                   # noqa: E402

from oracle import oracle as O                       # noqa: E402
from radex_emcee_amd import workloads                # noqa: E402
from radex_emcee_amd.engine import Engine            # noqa: E402

SEEDS = (11, 222, 3333, 44444, 5, 66, 777, 8888, 99999, 101, 2020, 30303, 4, 55, 606, 7070)


def main():
    fx = np.load(os.path.join("tests", "golden", "ref_sensitivity.npz"))
    eng = Engine()
    mol = O.Molecule(eng.molfile)
    if "--norefine" in sys.argv:
        eng.set_refinement(False)
        print("refinement off")
    st0 = O.State(mol)
    Kf, Kl, out_total, out_sensitive = 0.0, 0.0, 0, 0
    for seed in SEEDS:
        name = "big_%d" % seed
        if name + "_walker" not in fx.files:
            continue
        cfg = workloads.config2(131072, seed=seed)
        idx = fx[name + "_walker"]
        W = cfg["walkers"][idx]
        eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
        tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
        eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
        src = O.Source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
        lnp, st, nit = eng.lnprob_batch(W, return_info=True)
        fl = eng.model_flux_batch(W)
        rl, rst, rnit = O.lnprob_batch(mol, src, W, nthreads=16)
        rf = O.model_flux_batch(mol, src, W, nthreads=16)[0]
        assert np.array_equal(st, rst) and (rst == 1).all()
        st0.backrad(cfg["tbg"])
        bmax = st0.arr("backi").max()
        tol = 1e-4 * np.abs(rf) + 1e-10 * (bmax * 10.0 ** W[:, 3] * 1e23)[:, None]
        with np.errstate(all="ignore"):
            dev = np.nanmax(np.abs(fl - rf) / tol, axis=1)
            dl = np.abs(lnp - rl) / np.maximum(np.abs(rl), 1.0)
        dev = np.where(np.isfinite(dev), dev, 0.0)
        dl = np.where(np.isfinite(dl), dl, 0.0)
        resp, rlnp = fx[name + "_resp_sb"].astype(np.float64), fx[name + "_resp_lnp"].astype(np.float64)
        order = np.argsort(np.argsort(-resp))                      # rank 0 = the binary's most sensitive walker
        bad = np.flatnonzero(dev > 1.0)
        print("seed %d: %d maxiter walkers | GPU vs reference arithmetic: flux dev max %.2e tol (%d beyond), lnprob %.2e | binary's own 1-ulp "
              "response: max %.2e tol (%d beyond), lnlike %.2e" % (seed, len(idx), dev.max(), len(bad), dl.max(), resp.max(),
                                                                   int((resp > 1).sum()), rlnp.max()), flush=True)
        for w in bad:
            out_total += 1
            out_sensitive += int(order[w] < 0.01 * len(idx))
            print("    walker %6d: GPU %.2e tol, lnprob %.2e | binary's response %.2e tol (rank %d of %d: top %.2f %%), lnlike %.2e"
                  % (idx[w], dev[w], dl[w], resp[w], order[w] + 1, len(idx), 100.0 * (order[w] + 1) / len(idx), rlnp[w]))
        over = dev > 1.0
        if over.any():
            Kf = max(Kf, float(np.max((dev[over] - 1.0) / np.maximum(resp[over], 1e-300))))
        overl = dl > 1e-4
        if overl.any():
            Kl = max(Kl, float(np.max((dl[overl] - 1e-4) / np.maximum(rlnp[overl], 1e-300))))
    print("GPU outliers (beyond the flux tolerance): %d, of which among the binary's most sensitive 1 %% of their draw: %d" % (out_total, out_sensitive))
    print("smallest K with GPU deviation <= K x binary's 1-ulp response + tolerance: flux %.1f, lnprob %.1f" % (Kf, Kl))


if __name__ == "__main__":
    main()
