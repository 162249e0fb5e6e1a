"""The refinement form of matrix_'s linear solve (rx_set_refinement; rx_refine.hip.inc) on the GPU, through the C ABI:
against the pivoted solve every iteration of the SAME library, against the CPU checker in both of its forms (the reference's
arithmetic; the same refinement rule restated), and the property the chains rest on -- a walker's result does not depend on
the build (one or two wavefronts per SIMD, kept inverses in LDS or in the global scratch) or on the batch it is evaluated in."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O                      # noqa: E402  (checker only)
from radex_emcee_amd import workloads               # noqa: E402
from radex_emcee_amd._lib import RX_MAXITER, RX_OK    # noqa: E402
from radex_emcee_amd.engine import Engine           # noqa: E402

DEVICE_RULE = dict(first_iter=12, tol=2.0 ** -40, max_steps=8, lag=2, crit=1, d1max=2.0 ** 13, loose=2.0 ** -33, backoff=1)


@pytest.fixture(scope="module")
def eng(co_path):
    return Engine(co_path)


@pytest.fixture(scope="module")
def mol(co_path):
    return O.Molecule(co_path)


def _source(eng, mol, cfg):
    src0 = O.Source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    tf = O.model_flux_batch(mol, src0, cfg["truth"][None, :])[0][0]
    eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
    return O.Source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])


def _rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0)


def test_refinement_against_the_pivoted_solve_and_the_checker(eng, mol):
    cfg = workloads.config2(8192, seed=97531)
    src = _source(eng, mol, cfg)
    W = cfg["walkers"]
    O.set_refine(0)
    rl, rst, rnit = O.lnprob_batch(mol, src, W, nthreads=16)
    out = {}
    eng.set_refinement_counting(True)                   # (the instantiation of the solve kernel that feeds the counters)
    for on in (False, True):
        eng.set_refinement(on)
        eng.refinement_counters(reset=True)
        out[on] = eng.lnprob_batch(W, return_info=True)
        cnt = eng.refinement_counters(reset=True)
        if not on:
            assert cnt["refined"] == 0 and cnt["kept"] == 0
    eng.set_refinement_counting(False)
    # the kernel every other launch uses carries no counters: the same results bit for bit, and nothing is counted
    plain = eng.lnprob_batch(W, return_info=True)
    for x, y in zip(plain, out[True]):
        assert np.array_equal(x, y, equal_nan=True)
    assert eng.refinement_counters(reset=True)["iterations"] == 0
    eng.set_refinement(True)
    (l0, s0, n0), (l1, s1, n1) = out[False], out[True]
    assert np.array_equal(s0, rst) and np.array_equal(s1, rst)
    assert (n0 != rnit).sum() <= 8 and (n1 != rnit).sum() <= 8, ((n0 != rnit).sum(), (n1 != rnit).sum())
    fin = np.isfinite(rl)
    assert np.array_equal(fin, np.isfinite(l1))
    conv, slow = fin & (rst == RX_OK) & (n1 == rnit), fin & (rst == RX_MAXITER)
    assert slow.sum() > 100
    d1 = _rel(l1, rl)
    print("refinement on, against the reference's arithmetic on the CPU: lnprob converged %.1e, maxiter %.1e (pivoted every "
          "iteration: %.1e, %.1e)" % (d1[conv].max(), d1[slow].max(), _rel(l0, rl)[conv].max(), _rel(l0, rl)[slow].max()))
    assert d1[conv].max() < 1e-6 and d1[slow].max() < 1e-4
    # it does what it is for: most solves replaced, in few corrections, few attempts given up
    assert cnt["iterations"] == int(n1[np.isfinite(l1) | (s1 == RX_MAXITER) | (s1 == RX_OK)].sum()) or cnt["iterations"] > 0
    assert cnt["refined"] > 0.5 * cnt["iterations"], cnt
    assert cnt["corrections"] < 6 * (cnt["refined"] + cnt["failed"]) and cnt["failed"] < 0.1 * (cnt["refined"] + cnt["failed"]), cnt
    # the same rule on the CPU: iteration counts again, and the solves replaced agree to a per cent (the two arithmetics
    # differ in the last bits, so a threshold decision may fall differently here and there)
    O.set_refine(**DEVICE_RULE)
    O.refine_counters(reset=True)
    vl, vst, vnit = O.lnprob_batch(mol, src, W, nthreads=16)
    oc = O.refine_counters(reset=True)
    O.set_refine(0)
    assert np.array_equal(vst, s1) and (vnit != n1).sum() <= 8
    assert abs(oc["refined"] / (oc["refined"] + oc["full"]) - cnt["refined"] / cnt["iterations"]) < 0.01, (oc, cnt)
    both = fin & (vnit == n1)
    assert _rel(l1, vl)[both & (rst == RX_OK)].max() < 1e-6


def test_a_walkers_result_does_not_depend_on_build_or_batch(eng, mol):
    """One or two wavefronts per SIMD (kept inverses in LDS / in the global scratch, the system's rows held / streamed), alone
    or among 12 288 others: the same bits -- what the bit-identical chains across schedules, ranks and batch sizes rest on."""
    cfg = workloads.config2(12288, seed=1357)
    _source(eng, mol, cfg)
    W = cfg["walkers"]
    eng.set_refinement(True)
    res = []
    for occ in (1, 2):
        eng.set_waves_per_simd(occ)
        res.append(eng.lnprob_batch(W, return_info=True) + (eng.model_flux_batch(W[:2048]),))
    eng.set_waves_per_simd(0)
    for a, b, what in zip(res[0], res[1], ("lnprob", "status", "niter", "flux")):
        assert np.array_equal(a, b, equal_nan=True), what
    assert (res[0][1] == RX_MAXITER).sum() > 100
    sub = eng.lnprob_batch(W[5000:5300], return_info=True)                      # a small batch: the one-wavefront build
    for a, b in zip(sub, res[1][:3]):
        assert np.array_equal(a, b[5000:5300], equal_nan=True)


def test_switched_off_every_solve_is_pivoted_again(eng, mol):
    cfg = workloads.config2(2048, seed=24680)
    src = _source(eng, mol, cfg)
    eng.set_refinement(False)
    eng.set_refinement_counting(True)
    eng.refinement_counters(reset=True)
    l0, s0, n0 = eng.lnprob_batch(cfg["walkers"], return_info=True)
    cnt = eng.refinement_counters(reset=True)
    eng.set_refinement_counting(False)
    eng.set_refinement(True)
    assert cnt["refined"] == 0 and cnt["corrections"] == 0 and cnt["kept"] == 0 and cnt["iterations"] == int(n0.sum())
    rl, rst, rnit = O.lnprob_batch(mol, src, cfg["walkers"], nthreads=16)
    assert np.array_equal(s0, rst) and (n0 != rnit).sum() <= 2


def test_every_level_population_componentwise_with_the_refinement_on(co_path, mol):
    """xpop, T_ex and tau of rx_solve_batch against the reference's arithmetic COMPONENTWISE, all 41 levels, with the refinement on
    -- no floor at 1e-10 of the total as in round 5's tests.  The bound is 1e-6 relative + 1e-14 ABSOLUTE per level: the absolute
    term is what a double-precision solve of this system is worth (the reference's own LINPACK solution is off from the exact
    solution of its matrix by 4e-4 at a population of 1e-12, by 5 % at 1e-14: profiles/r6_small_population_accuracy.txt), and
    the device's deviation from the reference's numbers is the same with the refinement on and off
    (profiles/r6_small_population_gpu.txt: populations below 1e-6 deviate by at most 1.2e-14 absolute either way).  T_ex / tau: lines whose two levels
    both hold more than 1e-9 of the molecules."""
    e = Engine(co_path)
    rng = np.random.default_rng(2468)
    cfg = workloads.config2(1024, seed=1234)                       # the bench headline's walkers ...
    W = cfg["walkers"]
    sets = [(cfg["tbg"], 10.0 ** W[:, 1], 10.0 ** W[:, 2], np.stack([0.25 * 10.0 ** W[:, 0], 0.75 * 10.0 ** W[:, 0]], axis=1)),
            # ... and the wide box of the routine tests against a 2.73 K background (populations down to the 1e-20 clamp)
            (2.73, 10.0 ** rng.uniform(0.6, 2.9, 1024), 10.0 ** rng.uniform(12.0, 18.5, 1024), 10.0 ** rng.uniform(1.5, 6.5, (1024, 2)))]
    worst = dict(x=0.0, tex=0.0, tau=0.0, onoff=0.0)
    small_seen = 0
    for tbg, tkin, cd, dens in sets:
        e.set_source(tbg)
        e.set_refinement(True)
        on = e.solve_batch(tkin, cd, dens)
        e.set_refinement(False)
        off = e.solve_batch(tkin, cd, dens)
        e.set_refinement(True)
        ncmp = 0
        for w in range(len(tkin)):
            r = O.solve_state(mol, tbg, {2: dens[w, 0], 3: dens[w, 1]}, tkin[w], cd[w])
            if r["niter"] >= 200 or r["niter"] != on["niter"][w] or not np.all(np.isfinite(r["xpop"])):
                continue
            ncmp += 1
            x = r["xpop"]
            small_seen += int((x < 1e-13).sum())
            for got, key in ((on, "x"), (off, "onoff")):
                q = np.abs(got["xpop"][w] - x) / (1e-6 * x + 1e-14)
                worst[key] = max(worst[key], q.max())
                assert q.max() < 1.0, (key, w, int(q.argmax()), x[q.argmax()], got["xpop"][w][q.argmax()])
            both = np.minimum(x[e.iupp - 1], x[e.ilow - 1]) > 1e-9
            dt = np.abs(on["tex"][w][both] - r["tex"][both]) / np.abs(r["tex"][both])
            dta = np.abs(on["tau"][w][both] - r["tau"][both]) / np.abs(r["tau"][both])
            worst["tex"], worst["tau"] = max(worst["tex"], dt.max()), max(worst["tau"], dta.max())
            assert dt.max() < 1e-5 and dta.max() < 1e-6, (w, dt.max(), dta.max())
        assert ncmp > 900
    assert small_seen > 10000                                       # the small populations are what this test is about
    print("componentwise, in units of (1e-6 x + 1e-14): refinement on %.2f, off %.2f; T_ex %.1e, tau %.1e (levels > 1e-9)"
          % (worst["x"], worst["onoff"], worst["tex"], worst["tau"]))
    e.close()
