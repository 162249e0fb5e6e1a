#!/usr/bin/env python3
"""How far do the `maxiter` walkers of the REFERENCE BINARY move when its libm is one ulp off?  (container only)

    python scripts/ref_sensitivity.py [--procs 8] [--batches name,...]   ->  tests/golden/ref_sensitivity.npz

Walkers that stop at maxiter = 200 never settle; a few of them are chaotic, and any two implementations of the same arithmetic
end up apart there.  Round 5 argued that from the HIP path against itself.  This script measures it on the reference's own machine
code: /root/reference/emcee/pyradex/radex/radex.so is mapped by oracle/macho_ref.py as for every other fixture, but its `_exp`
and `_log` imports are bound to glibc's results moved by -1, 0 or +1 ulp (a seeded sequence, probabilities 1/4, 1/2, 1/4) --
the difference between two correctly working libms.  For every walker of the batches below that the binary itself runs to
maxiter, matrix_ is driven as emcee/pyradex/core.py:903-920 drives it, once unperturbed and NPERT times perturbed, and stored:

    resp_sb   max over perturbations and lines J_up = 1..10 of |dS| / (1e-4 |S| + 1e-10 max backi)       (S: line surface brightness,
              core.py:986-1003; the unit is the GPU tests' flux tolerance, tests/test_gpu_parity.py:_flux_ok)
    resp_lnp  max over perturbations of |d lnlike| / max(|lnlike|, 1)  (the GPU tests' measure on lnprob; the batch's data: the
              model at the batch's truth, sigma = 10 %)

Batches: the sixteen 131 072-walker prior-box draws of scripts/big_parity_seeds.py (profiles/r5_big_parity_seeds.txt), and the
batches of the GPU tests whose maxiter tier used to have the flat ceiling MAXITER_CEIL = 3e-3 (tests/test_gpu_round2.py):
config2(65536, seed 5678), config3(512), config4(4096) with its prior-box half, config4(2048); and EVERY walker of the bench
headline (config2(1024, seed 1234): `headline_1024_all`, with the iteration counts and whether a perturbation moved them).
The oracle only CHOOSES (which walkers to run: its iteration counts equal the binary's, tests/test_oracle_ref_configs.py -- and every
chosen walker is checked to reach maxiter in the binary too) and supplies the batches' synthetic data (flux at the truth).
"""
import argparse
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.macho_ref import RefRadex                     # noqa: E402
from radex_emcee_amd import workloads                     # noqa: E402
from radex_emcee_amd.molecule import SYNTH_CO_PATH        # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_sensitivity.npz")
NLEV, NLINE, NKEEP = 41, 40, 11
THC_PY, FK_PY = 3.9728917142978573e-16, 1.4387768775039338   # core.py:981-984 (astropy CODATA-2018)
NPERT = 2
BIG_SEEDS = (11, 222, 3333, 44444, 5, 66, 777, 8888, 99999, 101, 2020, 30303, 4, 55, 606, 7070)


class PerturbedRefRadex(RefRadex):
    """RefRadex whose exp / log imports return glibc's value moved by -1 / 0 / +1 ulp (self.pert_seed; None: unperturbed)."""

    def __init__(self):
        self.pert_seed = None
        self._state = 0
        super().__init__()

    def reseed(self, seed):
        self.pert_seed = seed
        self._state = (0x9E3779B97F4A7C15 * (int(seed) + 1)) & 0xFFFFFFFFFFFFFFFF if seed is not None else 0

    def _wrap(self, fn):
        fn.restype = C.c_double
        fn.argtypes = [C.c_double]

        def f(x):
            y = fn(x)
            if self.pert_seed is None:
                return y
            s = self._state = (self._state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
            k = (s >> 62) & 3                      # 0: down, 1, 2: unchanged, 3: up
            if k == 0:
                return float(np.nextafter(y, -np.inf))
            if k == 3:
                return float(np.nextafter(y, np.inf))
            return y
        cb = C.CFUNCTYPE(C.c_double, C.c_double)(f)
        self._keep.append(cb)
        return C.cast(cb, C.c_void_p).value

    def _bind(self, libc):
        super()._bind(libc)
        import ctypes.util
        libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        from oracle.macho_ref import S_LAZY_SYMBOL_POINTERS, S_NON_LAZY_SYMBOL_POINTERS, INDIRECT_SYMBOL_LOCAL, INDIRECT_SYMBOL_ABS
        wrapped = {"exp": self._wrap(libm.exp), "log": self._wrap(libm.log)}
        n = 0
        for _sect, addr, size, flags, res1, _r2 in self.m.sections:
            if (flags & 0xFF) not in (S_LAZY_SYMBOL_POINTERS, S_NON_LAZY_SYMBOL_POINTERS):
                continue
            for i in range(size // 8):
                isym = self.m.indirect[res1 + i]
                if isym & (INDIRECT_SYMBOL_LOCAL | INDIRECT_SYMBOL_ABS):
                    continue
                name = self.m.symbols[isym][0]
                bare = name[1:] if name.startswith("_") else name
                if bare in wrapped and name not in self.m.defined:
                    C.c_uint64.from_address(self.base + addr + 8 * i).value = wrapped[bare]
                    n += 1
        assert n >= 2, "exp / log imports not found"


_R = None


def _image():
    global _R
    if _R is None:
        _R = PerturbedRefRadex()
    return _R


def setup(p0, p1, p2, tbg):
    """readdata_ + backrad_ for one walker (unperturbed: the perturbation is the ITERATION's).  readdata_ costs 120 ms through the
    loader's Fortran I/O, so a walker's runs share it: matrix_ does not touch what it leaves in COMMON."""
    R = _image()
    v = R.views()
    R.reseed(None)
    n_h2 = 10.0 ** p0
    R.readdata(SYNTH_CO_PATH, 10.0 ** p1, {2: 0.25 * n_h2, 3: 0.75 * n_h2})
    v["method"][0] = 2
    v["cdmol"][0] = 10.0 ** p2
    v["deltav"][0] = 1e5
    v["tbg"][0] = tbg
    R.backrad()


def iterate(pert):
    """One cold-start run_radex of the binary on the walker set up (core.py:896-925): (niter, S[NKEEP], max backi)."""
    R = _image()
    v = R.views()
    v["xpop"][:NLEV] = 0.0
    v["tex"][:NLINE] = 0.0
    v["taul"][:NLINE] = 0.0
    R.reseed(pert)
    it, conv = 0, 0
    last = v["xpop"][:NLEV].copy()
    while not conv:                                        # core.py:903-920
        if it >= 200:
            break
        conv = R.matrix(it, conv)
        x = v["xpop"][:NLEV]
        if np.abs(last - x).sum() < 1e-16 and it > 10:
            break
        last = x.copy()
        it += 1
    R.reseed(None)
    if R.trap_log:
        raise RuntimeError("trap fired: %r" % (R.trap_log,))
    tex, taul, backi, xnu = (v[k][:NKEEP].copy() for k in ("tex", "taul", "backi", "xnu"))
    with np.errstate(all="ignore"):                        # core.py:986-1003
        ftau = np.exp(-taul)
        bnutex = THC_PY * xnu ** 3 / (np.exp(FK_PY * xnu / tex) - 1.0)
        sb = backi * ftau + bnutex * (1.0 - ftau) - backi
    return it, sb, float(v["backi"][:NLINE].max())


def lnlike(model, flux, esig):
    r = (flux - model) / esig
    return -0.5 * (np.sum(r * r) + 2.0 * np.sum(np.log(esig)))


def job(args):
    """One walker (1 or 2 components) of a batch: its response to the perturbations."""
    w, p, tbg, jidx, flux, esig = args
    ncomp = len(p) // 4
    base, pert = [], [[] for _ in range(NPERT)]
    nit, nit_moved = 0, False
    for c in range(ncomp):
        q = p[4 * c:4 * c + 4]
        setup(q[0], q[1], q[2], tbg)
        it, sb, bmax = iterate(None)
        nit = max(nit, it)
        base.append((sb, bmax, q[3]))
        for k in range(NPERT):
            itk, sbk, _ = iterate(1000 * k + 17 + c)
            nit_moved = nit_moved or itk != it
            pert[k].append(sbk)
    resp_sb, resp_lnp = 0.0, 0.0
    m0 = sum(sb[jidx] * 10.0 ** s * 1e23 for sb, _, s in base)
    for k in range(NPERT):
        for (sb, bmax, _), sp in zip(base, pert[k]):
            with np.errstate(all="ignore"):
                d = np.abs(sp[:10] - sb[:10]) / (1e-4 * np.abs(sb[:10]) + 1e-10 * bmax)
            d = d[np.isfinite(d)]
            resp_sb = max(resp_sb, float(d.max()) if len(d) else 0.0)
        mk = sum(sp[jidx] * 10.0 ** s * 1e23 for sp, (_, _, s) in zip(pert[k], base))
        with np.errstate(all="ignore"):
            l0 = lnlike(m0, flux, esig)
            dl = abs(lnlike(mk, flux, esig) - l0) / max(abs(l0), 1.0)
        resp_lnp = max(resp_lnp, float(dl) if np.isfinite(dl) else 0.0)
    return w, nit, resp_sb, resp_lnp, float(nit_moved)


def batches():
    """name -> (walkers [N, 4 ncomp], per-walker (tbg, jidx, flux, esig) through a source list and index)."""
    from oracle import oracle as O
    mol = O.Molecule(SYNTH_CO_PATH)

    def truth(cfg, ncomp=1):
        s0 = O.Source(cfg["tbg"], cfg["Jup"], np.ones(len(cfg["Jup"])), np.ones(len(cfg["Jup"])), cfg["bounds"], ncomp=ncomp,
                      T_d=cfg.get("T_d"))
        tf = O.model_flux_batch(mol, s0, cfg["truth"][None, :])[0][0]
        return O.Source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"], ncomp=ncomp, T_d=cfg.get("T_d")), tf

    out = {}
    for seed in BIG_SEEDS:
        def mk(seed=seed):
            cfg = workloads.config2(131072, seed=seed)
            src, tf = truth(cfg)
            return cfg["walkers"], [src], np.zeros(131072, dtype=np.int32), [(cfg["tbg"], np.asarray(cfg["Jup"]) - 1, tf, 0.1 * tf)]
        out["big_%d" % seed] = mk

    def headline():
        cfg = workloads.config2(1024, seed=1234)
        src, tf = truth(cfg)
        return cfg["walkers"], [src], np.zeros(1024, dtype=np.int32), [(cfg["tbg"], np.asarray(cfg["Jup"]) - 1, tf, 0.1 * tf)]
    out["headline_1024_all"] = headline                    # (EVERY walker of the bench headline, converged ones included)

    def c5():
        cfg = workloads.config2(65536, seed=5678)
        src, tf = truth(cfg)
        return cfg["walkers"], [src], np.zeros(65536, dtype=np.int32), [(cfg["tbg"], np.asarray(cfg["Jup"]) - 1, tf, 0.1 * tf)]
    out["config5_65536"] = c5

    def c3():
        cfg = workloads.config3(512)
        srcs = [O.Source(s["tbg"], s["Jup"], s["flux"], s["eflux"], s["bounds"]) for s in cfg["sources"]]
        data = [(s["tbg"], np.asarray(s["Jup"]) - 1, np.asarray(s["flux"], dtype=np.float64),
                 np.maximum(np.abs(np.asarray(s["eflux"], dtype=np.float64)), 1e-12)) for s in cfg["sources"]]
        return cfg["walkers"].reshape(-1, 4), srcs, np.asarray(cfg["src_index"], dtype=np.int32), data
    out["config3_512"] = c3

    def c4mix():
        cfg = workloads.config4(4096)
        W = cfg["walkers"].copy()
        W[2048:] = workloads.draw_prior_2comp(cfg["bounds"], 2048, 91)
        src, tf = truth(cfg, 2)
        return W, [src], np.zeros(4096, dtype=np.int32), [(cfg["tbg"], np.asarray(cfg["Jup"]) - 1, tf, 0.1 * tf)]
    out["config4_4096_mixed"] = c4mix

    def c4():
        cfg = workloads.config4(2048)
        src, tf = truth(cfg, 2)
        return cfg["walkers"], [src], np.zeros(2048, dtype=np.int32), [(cfg["tbg"], np.asarray(cfg["Jup"]) - 1, tf, 0.1 * tf)]
    out["config4_2048"] = c4
    return mol, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--batches", default="")
    ap.add_argument("--limit", type=int, default=0, help="at most so many maxiter walkers per batch (smoke runs)")
    a = ap.parse_args()
    from oracle import oracle as O
    from tests.golden.make_ref_configs import save_deterministic
    mol, B = batches()
    names = [n for n in B if not a.batches or n in a.batches.split(",")]
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    t0 = time.time()
    with mp.Pool(a.procs) as pool:
        for name in names:
            W, srcs, sidx, data = B[name]()
            st = np.zeros(len(W), dtype=np.int32)
            for k, src in enumerate(srcs):                 # the oracle CHOOSES: status 1 = some component stopped at maxiter
                m = sidx == k
                st[m] = O.lnprob_batch(mol, src, W[m], nthreads=a.procs)[1]
            idx = np.flatnonzero(st == 1) if not name.endswith("_all") else np.flatnonzero(st != 3)
            if a.limit:
                idx = idx[:a.limit]
            jobs = [(int(w), W[w], data[sidx[w]][0], data[sidx[w]][1], data[sidx[w]][2], data[sidx[w]][3]) for w in idx]
            rows = []
            for k, r in enumerate(pool.imap(job, jobs, chunksize=4)):
                rows.append(r)
                if (k + 1) % 500 == 0:
                    print("  %s: %d / %d walkers, %.0f s" % (name, k + 1, len(jobs), time.time() - t0), flush=True)
            rows.sort()
            r = np.array(rows, dtype=np.float64).reshape(-1, 5)
            if name.endswith("_all"):
                out[name + "_niter"] = r[:, 1].astype(np.int32)
                out[name + "_niter_moved"] = r[:, 4].astype(np.int8)
                conv = r[:, 1] < 200
                print("%s: %d walkers, %d converge: their 1-ulp response max %.2e of the flux tolerance, relative |d lnlike| max %.2e; "
                      "iteration counts moved by the perturbation: %d" % (name, len(r), int(conv.sum()), r[conv, 2].max(), r[conv, 3].max(),
                                                                           int(r[:, 4].sum())), flush=True)
            else:
                assert np.all(r[:, 1] >= 200), "%s: a chosen walker does not reach maxiter in the binary" % name
            out[name + "_walker"] = r[:, 0].astype(np.int32)
            out[name + "_resp_sb"] = r[:, 2].astype(np.float32)
            out[name + "_resp_lnp"] = r[:, 3].astype(np.float32)
            q = r[:, 2]
            if not name.endswith("_all"):
                print("%s: %d maxiter walkers; 1-ulp response in units of the flux tolerance: median %.2e, 99th pct %.2e, max %.2e, "
                      "above 1: %d; relative |d lnlike| max %.2e  (%.0f s)"
                      % (name, len(q), np.median(q), np.percentile(q, 99), q.max(), int((q > 1).sum()), r[:, 3].max(), time.time() - t0),
                      flush=True)
            save_deterministic(OUT, out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
