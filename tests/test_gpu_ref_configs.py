"""north_star: "line fluxes ... within 1e-4 relative tolerance on flux against the reference Fortran RADEX on the same inputs" --
on BASELINE's own workloads, with no oracle in between: the HIP kernels (through the C ABI: rx_solve_batch,
rx_model_flux_batch) against tests/golden/ref_configs.npz, the numbers /root/reference/emcee/pyradex/radex/radex.so itself
computed for the 1024 config-2 walkers of the bench.py headline, both components of config 4's 2048 walkers and 64 walkers per
source of config 3 (tests/golden/make_ref_configs.py: its readdata_ / backrad_ / matrix_ driven as core.py:896-925 drives them,
cold start; surface brightness by core.py:986-1003 from the binary's T_ex / tau / backi).

Bars: iteration counts and conv flags equal (a convergence test may flip on the last bit: at most 0.5 % of the walkers, and those
still within the flux bar); line surface brightness / flux within 1e-4 relative + the background floor of
tests/test_gpu_parity.py (|dS| <= 1e-4 |S| + 1e-10 max_l backi_l) -- for the walkers the binary stops at maxiter too (their answer is a
snapshot of an iteration that never settles: they were held to 1e-3 while the kernels ran an elimination in every iteration;
observed since round 5's refinement: 1.1e-8 at worst)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from radex_emcee_amd import workloads               # noqa: E402
from radex_emcee_amd.engine import Engine           # noqa: E402

NK = 11


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_configs.npz"))


@pytest.fixture(scope="module")
def eng(co_path):
    return Engine(co_path)


def _solve(eng, params, tbg):
    n = 10.0 ** params[:, 0]
    eng.set_source(tbg)
    return eng.solve_batch(10.0 ** params[:, 1], 10.0 ** params[:, 2], np.stack([0.25 * n, 0.75 * n], 1))


def _check_solves(tag, r, niter, conv, sb, backi):
    nit = np.asarray(r["niter"])
    same = nit == niter
    assert same.mean() >= 0.995, (tag, (~same).sum())
    st = np.asarray(r["status"])
    assert np.array_equal(st[same] == 0, conv[same] == 1), tag
    got = np.asarray(r["sb"])[:, :NK]
    floor = 1e-10 * backi.max(axis=1, keepdims=True)
    d = np.abs(got - sb)
    both_nan = np.isnan(got) & np.isnan(sb)
    settled = (conv == 1)[:, None]
    tol = 1e-4 * np.abs(sb) + floor
    ok = (d <= tol) | both_nan
    assert ok.all(), (tag, np.argwhere(~ok)[:5], (d / tol)[~ok][:5])
    sig = settled & (1e-4 * np.abs(sb) > 100.0 * floor) & np.isfinite(sb)        # (where the background floor plays no part)
    rel = d[sig] / np.abs(sb[sig])
    print("%s: %d solves, iteration count equal on %d; surface brightness of the settled walkers vs the reference binary: "
          "median %.1e, worst %.1e; walkers at maxiter: %d, worst %.1e"
          % (tag, len(nit), same.sum(), np.median(rel), rel.max(), (conv == 0).sum(),
             np.nanmax((d / np.maximum(np.abs(sb), floor))[~settled[:, 0]]) if (conv == 0).any() else 0.0))
    return rel.max()


def test_config2_headline_batch_against_the_reference_binary(eng, ref):
    cfg = workloads.config2(1024, 1234)
    assert np.array_equal(cfg["walkers"], ref["c2_params"])
    r = _solve(eng, ref["c2_params"], cfg["tbg"])
    worst = _check_solves("config 2", r, ref["c2_niter"], ref["c2_conv"], ref["c2_sb"], ref["c2_backi"])
    assert worst < 1e-4
    # and as model_lvg's fluxes (emcee_radex.py:120-130): flux = S[Jup-1] * 10^size * 1e23, J-indexing included
    eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    flux, st, nit = eng.model_flux_batch(cfg["walkers"], return_info=True)
    want = ref["c2_sb"][:, cfg["Jup"] - 1] * (10.0 ** cfg["walkers"][:, 3:4]) * 1e23
    floor = 1e-10 * ref["c2_backi"].max(axis=1, keepdims=True) * (10.0 ** cfg["walkers"][:, 3:4]) * 1e23
    tol = 1e-4 * np.abs(want) + floor
    ok = (np.abs(flux - want) <= tol) | (np.isnan(flux) & np.isnan(want))
    assert ok.all(), np.argwhere(~ok)[:5]


def test_config4_components_against_the_reference_binary(eng, ref):
    cfg = workloads.config4(2048)
    comps = ref["c4_params"].reshape(-1, 4)
    r = _solve(eng, comps, cfg["tbg"])
    _check_solves("config 4 (components)", r, ref["c4_niter"], ref["c4_conv"], ref["c4_sb"], ref["c4_backi"])


def test_config3_sources_against_the_reference_binary(eng, ref):
    src = ref["c3_src"]
    for k in range(16):
        m = src == k
        r = _solve(eng, ref["c3_params"][m], float(ref["c3_tbg"][k]))
        _check_solves("config 3 source %d" % k, r, ref["c3_niter"][m], ref["c3_conv"][m], ref["c3_sb"][m], ref["c3_backi"][m])
