"""BASELINE's own workloads against the reference BINARY (tests/golden/ref_configs.npz, made by
tests/golden/make_ref_configs.py: radex.so's readdata_ / backrad_ / matrix_ run on the 1024 config-2 walkers of the bench.py
headline, both components of config 4's 2048 walkers and 64 walkers per source of config 3).  Here: the oracle on the same
walkers -- iteration counts, conv flags, T_ex, tau and the line surface brightness of J_up = 1..11 equal the binary's bit for
bit.  (tests/test_gpu_ref_configs.py holds the HIP kernels to the same numbers, north_star's "within 1e-4 of the reference
Fortran RADEX on the same inputs".)"""
import os

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_configs.npz"))


@pytest.fixture(scope="module")
def mol(co_path):
    return O.Molecule(co_path)


def _check(mol, params, tbg, niter, conv, tex, taul):
    bad = 0
    for k, p in enumerate(params):
        n_h2 = 10.0 ** p[0]
        r = O.solve_state(mol, tbg[k] if np.ndim(tbg) else tbg, {2: 0.25 * n_h2, 3: 0.75 * n_h2}, 10.0 ** p[1], 10.0 ** p[2])
        assert r["niter"] == niter[k] and int(r["converged"]) == conv[k], (k, r["niter"], niter[k])
        nk = tex.shape[1]
        same = (np.array_equal(r["tex"][:nk], tex[k]) or np.array_equal(np.isnan(r["tex"][:nk]), np.isnan(tex[k]))) and \
               (np.array_equal(r["tau"][:nk], taul[k]) or np.array_equal(np.isnan(r["tau"][:nk]), np.isnan(taul[k])))
        bad += not same
    assert bad == 0


def test_config2_headline_batch_is_the_binarys(ref, mol):
    from radex_emcee_amd import workloads
    cfg = workloads.config2(1024, 1234)
    assert np.array_equal(cfg["walkers"], ref["c2_params"])              # the fixture is OF the bench.py batch
    _check(mol, ref["c2_params"], cfg["tbg"], ref["c2_niter"], ref["c2_conv"], ref["c2_tex"], ref["c2_taul"])
    assert (ref["c2_niter"] >= 200).sum() >= 20                          # the walkers that set the length of the launch are in it


def test_config4_components_are_the_binarys(ref, mol):
    from radex_emcee_amd import workloads
    cfg = workloads.config4(2048)
    assert np.array_equal(cfg["walkers"], ref["c4_params"])
    comps = ref["c4_params"].reshape(-1, 4)[::8]                        # every eighth solve: the CPU suite's share (512)
    _check(mol, comps, cfg["tbg"], ref["c4_niter"][::8], ref["c4_conv"][::8], ref["c4_tex"][::8], ref["c4_taul"][::8])


def test_config3_sources_are_the_binarys(ref, mol):
    tbg = ref["c3_tbg"][ref["c3_src"]]
    _check(mol, ref["c3_params"], tbg, ref["c3_niter"], ref["c3_conv"], ref["c3_tex"], ref["c3_taul"])


def test_sensitivity_fixture_names_the_maxiter_walkers_of_the_test_batches(mol, golden_dir):
    """tests/golden/ref_sensitivity.npz (scripts/ref_sensitivity.py: the reference binary with its exp / log one ulp off) is
    what the GPU tests derive the ceiling of their maxiter walkers from: its walker lists must be exactly the walkers of the
    batches that stop at maxiter (checked here for the two small ones; the binary itself confirmed maxiter for every one of
    them when the fixture was made), and the summary the README quotes must be what the fixture holds."""
    from radex_emcee_amd import workloads
    f = np.load(os.path.join(golden_dir, "ref_sensitivity.npz"))
    c3 = workloads.config3(512)
    st = np.concatenate([O.lnprob_batch(mol, O.Source(s["tbg"], s["Jup"], s["flux"], s["eflux"], s["bounds"]), c3["walkers"][k], nthreads=8)[1]
                         for k, s in enumerate(c3["sources"])])
    assert np.array_equal(np.asarray(c3["src_index"]).reshape(16, 512)[:, 0], np.arange(16))
    assert np.array_equal(np.flatnonzero(st == 1), f["config3_512_walker"])
    big = [k[:-7] for k in f.files if k.startswith("big_") and k.endswith("_walker")]
    assert len(big) == 16
    resp = np.concatenate([f[b + "_resp_sb"] for b in big])
    assert len(resp) == 49947
    assert int((resp > 1.0).sum()) == 13 and 28.0 < resp.max() < 29.5                # the binary itself: 13 walkers beyond the tolerance, worst 29 x
    assert np.median(resp) < 1e-7
    # every walker of the bench headline, converged ones included: a libm that differs by one ulp moves no iteration count (they
    # are the binary's own of ref_configs.npz) and the converged walkers by less than 1e-3 of the flux tolerance
    ref = np.load(os.path.join(golden_dir, "ref_configs.npz"))
    assert np.array_equal(f["headline_1024_all_walker"], np.arange(1024)) and np.array_equal(f["headline_1024_all_niter"], ref["c2_niter"])
    assert int(f["headline_1024_all_niter_moved"].sum()) == 0
    conv = f["headline_1024_all_niter"] < 200
    assert conv.sum() >= 1000 and f["headline_1024_all_resp_sb"][conv].max() < 1e-3 and f["headline_1024_all_resp_lnp"][conv].max() < 1e-10
