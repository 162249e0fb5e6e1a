"""The refinement form of matrix_'s linear solve (the device kernels' scheme, rx_refine.hip.inc) restated on the CPU
(oracle.set_refine -> radex_oracle.c: rf_solve) against the reference's arithmetic, which pivots every iteration.

The gate the kernels were built behind (scripts/refine_gate.py, profiles/r5_refine_gate_*.txt), at a size the CPU suite
affords: status and iteration counts must not move, lnprob must stay orders of magnitude inside north_star's 1e-4, and with
the variant switched off the oracle must be the reference's arithmetic bit for bit again."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from radex_emcee_amd import workloads

DEVICE_RULE = dict(first_iter=12, tol=2.0 ** -40, max_steps=8, lag=2, crit=1, d1max=2.0 ** 13, loose=2.0 ** -33, backoff=1)


@pytest.fixture(scope="module")
def mol(co_path):
    return O.Molecule(co_path)


@pytest.fixture(autouse=True)
def _refinement_off_afterwards():
    yield
    O.set_refine(0)


def _config2(mol, n, seed):
    cfg = workloads.config2(n, seed=seed)
    src0 = O.Source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    tf = O.model_flux_batch(mol, src0, cfg["truth"][None, :])[0][0]
    return cfg, O.Source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])


def test_refinement_keeps_status_and_iteration_counts(mol):
    cfg, src = _config2(mol, 4096, 97531)
    O.set_refine(0)
    lnp0, st0, nit0 = O.lnprob_batch(mol, src, cfg["walkers"], nthreads=4)
    O.set_refine(**DEVICE_RULE)
    O.refine_counters(reset=True)
    lnp1, st1, nit1 = O.lnprob_batch(mol, src, cfg["walkers"], nthreads=4)
    cnt = O.refine_counters(reset=True)
    assert np.array_equal(st0, st1)
    assert (nit0 != nit1).sum() <= 1, np.flatnonzero(nit0 != nit1)          # (262 144 walkers: none, profiles/r5_refine_gate_262144.txt)
    fin = np.isfinite(lnp0)
    assert np.array_equal(fin, np.isfinite(lnp1))
    dev = np.abs(lnp1[fin] - lnp0[fin]) / np.maximum(np.abs(lnp0[fin]), 1.0)
    assert dev[st0[fin] == 0].max() < 1e-6 and dev.max() < 1e-5, (dev[st0[fin] == 0].max(), dev.max())
    # it does replace most of the solves, in few corrections, and rarely gives an attempt up
    tot = cnt["full"] + cnt["refined"]
    assert cnt["refined"] > 0.5 * tot and cnt["steps"] < 6 * (cnt["refined"] + cnt["failed"])
    assert cnt["failed"] < 0.1 * (cnt["refined"] + cnt["failed"])
    assert (st0 == 1).sum() > 50                                             # the draw holds walkers that never converge


def test_switched_off_it_is_the_reference_arithmetic_again(mol, golden_dir):
    """After a run with the variant on, a state created with it off reproduces the reference binary's histories bit for bit."""
    cfg, src = _config2(mol, 64, 5)
    O.set_refine(**DEVICE_RULE)
    O.lnprob_batch(mol, src, cfg["walkers"])
    O.set_refine(0)
    g = json.load(open(os.path.join(golden_dir, "ref_matrix.json")))
    n = 0
    for c in g["cases"]:
        if c["mol"] != "co_synth" or c["method"] != 2:
            continue
        r = O.solve_state(mol, c["tbg"], {int(k): v for k, v in c["density"].items()}, c["tkin"], c["cdmol"])
        assert r["niter"] == c["niter"]
        assert np.array_equal(r["xpop"], np.array(c["xpop"])) or np.all(np.isnan(r["xpop"]) == np.isnan(np.array(c["xpop"])))
        n += 1
    assert n >= 10


def test_refined_histories_stay_on_the_reference_binarys(mol, golden_dir):
    """The 26 iteration histories of the reference's own matrix_ (ref_matrix.json; three of them exhaust maxiter): with the
    refinement the CO / LVG ones end after the same number of iterations, on populations within 1e-6 x + 1e-14 of the binary's."""
    g = json.load(open(os.path.join(golden_dir, "ref_matrix.json")))
    O.set_refine(**DEVICE_RULE)
    n = 0
    for c in g["cases"]:
        if c["mol"] != "co_synth" or c["method"] != 2:
            continue
        r = O.solve_state(mol, c["tbg"], {int(k): v for k, v in c["density"].items()}, c["tkin"], c["cdmol"])
        assert r["niter"] == c["niter"], (c["tkin"], c["cdmol"], r["niter"], c["niter"])
        want = np.array(c["xpop"])
        if np.any(np.isnan(want)):
            continue
        tol = 1e-6 * np.abs(want) + 1e-14
        if c["niter"] >= 200:
            tol = 1e-3 * np.abs(want) + 1e-12          # walkers that never settle amplify round-off over 200 iterations
        assert np.all(np.abs(r["xpop"] - want) <= tol), (c["tkin"], c["cdmol"], np.max(np.abs(r["xpop"] - want) / tol))
        n += 1
    assert n >= 10
