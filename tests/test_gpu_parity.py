"""GPU parity tests: the HIP path (through the C ABI of libradex_emcee_amd.so) against
(a) golden vectors computed by the reference's own radex.so and (b) the CPU oracle.

Tolerances (north_star: flux within 1e-4 relative of reference RADEX):
  * populations vs the reference binary: |dx| <= 1e-6*x + 1e-14 on every level (the
    absolute term covers levels below ~1e-8 where the reference's own LU sits on its
    round-off floor); Tex / tau: 1e-6 relative on lines whose two levels have x > 1e-8;
  * fluxes vs the oracle: |df| <= 1e-4*|f| + 1e-10*F_bg, where
    F_bg = max_l backi_l * size * 1e23 is the background flux through the source
    solid angle: source_line_surfbrightness is `toti - backi`, so absolute errors
    of order eps*backi are cancellation noise of the reference itself;
  * iteration counts and status codes must be identical.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O                      # noqa: E402  (checker only)
from radex_emcee_amd import workloads               # noqa: E402
from radex_emcee_amd.engine import Engine           # noqa: E402

FLUX_RTOL = 1e-4
BG_FLOOR = 1e-10


@pytest.fixture(scope="module")
def engines(co_path):
    return {m: Engine(co_path, escapeProbGeom=n) for m, n in ((1, "sphere"), (2, "lvg"), (3, "slab"))}


@pytest.fixture(scope="module")
def mol(co_path):
    return O.Molecule(co_path)


def _flux_ok(flux, ref, walkers, tbg, mol, ncomp=1):
    st = O.State(mol)
    st.backrad(tbg)
    bmax = st.arr("backi").max()
    size = np.zeros(len(walkers))
    for c in range(ncomp):
        size = np.maximum(size, 10.0 ** walkers[:, 4 * c + 3])
    tol = FLUX_RTOL * np.abs(ref) + BG_FLOOR * (bmax * size * 1e23)[:, None]
    d = np.abs(flux - ref)
    both_nan = np.isnan(flux) & np.isnan(ref)
    return (d <= tol) | both_nan, d


def test_lubksb_vs_reference_binary(engines, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "ref_lubksb.json")))
    for c in g["cases"]:
        n = c["n"]
        x = engines[2].lubksb_batch(np.array(c["A"]).reshape(n, n))[0]
        want = np.array(c["x"])
        assert np.max(np.abs(x - want) / np.abs(want)) < 1e-10, n


def test_backrad_vs_reference_binary(engines, golden_dir):
    """Row a7 on the product side [radex.so@0x1be30 backrad_, through the `tbg` setter emcee/pyradex/core.py:845-854]: the
    background table rx_set_source uploads -- read back from the DEVICE by rx_background -- equals the reference binary's
    backi / totalb / trj for all 26 (molecule, T_bg) vectors of ref_backrad.json, bit for bit (host arithmetic in the
    binary's operand order on the same libm: pow(xnu, 3.0), the 160 guard and its 1e-30 floor)."""
    g = json.load(open(os.path.join(golden_dir, "ref_backrad.json")))
    guard = json.load(open(os.path.join(golden_dir, "ref_backrad_guard.json")))     # T_bg 0.24 .. 2 K: the exp-guard branch
    toy = Engine(os.path.join(golden_dir, "toy6.dat"))
    assert len(g["cases"]) == 26
    floor_seen = False
    for k, c in enumerate(g["cases"] + guard["cases"]):
        eng = engines[2] if c["mol"] == "co_synth" else toy
        slot = k % 5                                                   # (slots are independent tables)
        eng.set_source(c["tbg"], src=slot)
        backi, trj = eng.background(slot)
        assert np.array_equal(backi, np.array(c["backi"])), (c["mol"], c["tbg"])
        assert np.array_equal(backi, np.array(c["totalb"]))            # totalb = backi on this path
        assert np.all(np.array(c["trj"]) == trj) and trj == c["tbg"]
        floor_seen |= bool(np.any(backi == 1e-30))
    assert floor_seen                                                  # the exp-guard branch is among the vectors
    with pytest.raises(Exception):
        toy.background(63)                                             # a slot that was never set


def test_lamda_tables_vs_reference_readdata(engines, golden_dir):
    """Row a6 on the product side: what the library's own LAMDA parser (rx_create) holds for the two committed
    files equals what the reference binary's readdata_ parsed from them (ref_readdata.json: xnu = E_up - E_low,
    spfreq, iupp, ilow), bit for bit.  The rate arithmetic of readdata_ is pinned through the oracle
    (tests/test_oracle_golden.py) and through the matrix_ histories below, whose crate / ctot are the binary's own."""
    g = json.load(open(os.path.join(golden_dir, "ref_readdata.json")))
    toy = Engine(os.path.join(golden_dir, "toy6.dat"))
    for name, eng in (("co_synth", engines[2]), ("toy6", toy)):
        t = g["molecules"][name]
        assert (eng.nlev, eng.nline) == (t["nlev"], t["nline"])
        assert np.array_equal(eng.xnu, np.array(t["xnu"])) and np.array_equal(eng.spfreq, np.array(t["spfreq"]))
        assert list(eng.iupp) == t["iupp"] and list(eng.ilow) == t["ilow"]
    toy.close()


def test_escprob_vs_reference_binary(engines, golden_dir):
    """The device escprob_ against the 88 values the reference binary's own escprob_ returned
    (sphere / LVG / slab, the branch boundaries, the NaN of the LVG maser branch)."""
    g = json.load(open(os.path.join(golden_dir, "ref_escprob.json")))
    for method in (1, 2, 3):
        cs = [c for c in g["cases"] if c["method"] == method]
        tau = np.array([c["tau"] for c in cs])
        want = np.array([np.nan if c["beta"] is None else c["beta"] for c in cs], dtype=float)
        got = engines[2].escprob_batch(tau, method)
        assert np.array_equal(np.isnan(got), np.isnan(want)), method
        f = ~np.isnan(want)
        # OCML exp / the kernel's log against the binary's libm: 1 ulp of exp(-x), amplified by the
        # cancellation in 1 - exp(-x) near the branch boundaries (x ~ 0.02: x100)
        dev = np.abs(got[f] - want[f]) / np.abs(want[f])
        assert dev.max() < 2e-13, (method, tau[f][np.argmax(dev)], dev.max())
        assert np.median(dev) < 3e-16, method


def test_kernel_logarithm(engines):
    """rx_log (plain-double argument reduction + polynomial, used by the LVG escape probability and the
    excitation temperatures): <= 2 ulp from libm over the whole range, special operands as libm."""
    rng = np.random.default_rng(7)
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 20000), rng.uniform(0.5, 2.0, 20000),
                        1.0 + rng.uniform(-1e-3, 1e-3, 5000), np.array([1.0, 2.0, 0.5, 4.9e-324, 2.2e-308, 1.7e308])])
    got = engines[2].escprob_batch(x, 0)
    want = np.log(x)
    ulp = np.abs(got - want) / np.spacing(np.abs(want) + 5e-324)
    assert ulp.max() <= 2.0, (ulp.max(), x[np.argmax(ulp)])
    with np.errstate(all="ignore"):
        sp = np.array([0.0, -0.0, -1.0, -1e-300, np.inf, np.nan])
        g2 = engines[2].escprob_batch(sp, 0)
    assert g2[0] == -np.inf and g2[1] == -np.inf and np.isnan(g2[2]) and np.isnan(g2[3]) and g2[4] == np.inf and np.isnan(g2[5])


def test_pivot_choices_vs_reference_binary(engines, golden_dir, tmp_path):
    """Pivot row of every elimination step against the ipvt of the reference's own sgefa_
    (tests/golden/ref_sgefa.json): gaussian systems, exact ties at step 0, an exact tie at step 1
    that sgefa_ resolves by POSITION after the interchange of step 0.  Integer output: equality.
    (Small-integer systems are skipped: there FMA contraction can turn an exact tie of the
    reference's arithmetic into a near-tie.)"""
    from radex_emcee_amd.molecule import synth_co_text
    g = json.load(open(os.path.join(golden_dir, "ref_sgefa.json")))
    eng_by_n = {41: engines[2]}
    checked = 0
    for c in g["cases"]:
        n = c["n"]
        if c["kind"] == "int" or c["info"] != 0:
            continue
        if n not in eng_by_n:
            path = tmp_path / ("rotor%d.dat" % n)
            path.write_text(synth_co_text(nlev=n))
            eng_by_n[n] = Engine(str(path))
        _x, piv = eng_by_n[n].lubksb_batch(np.array(c["A"]).reshape(n, n), return_pivots=True)
        order, rows = list(range(n)), []
        for k in range(n):
            l = c["ipvt"][k] if k < n - 1 else n - 1
            order[k], order[l] = order[l], order[k]
            rows.append(order[k])
        assert list(piv[0]) == rows, (c["kind"], n)
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("n", [3, 8, 9, 20, 21, 32, 41, 45, 48, 64])
def test_lubksb_random_vs_oracle(engines, n, tmp_path):
    """The LU alone on matrices that are nothing like a rate matrix: an interchange at almost every
    step (gaussian), exact ties between candidates (small integers; isamax's first-maximum rule),
    singular systems (sgeir_ returns without solving: x stays e_last).  n <= 41 runs in the CO
    engine's padded <41> kernel, the other sizes in the instantiation of a rotor with n levels."""
    rng = np.random.RandomState(1000 + n)
    eng = engines[2]
    if n in (8, 20, 32, 45, 48, 64):
        from radex_emcee_amd.molecule import synth_co_text
        path = tmp_path / ("rotor%d.dat" % n)
        path.write_text(synth_co_text(nlev=n))
        eng = Engine(str(path))
    mats = []
    for _ in range(12):
        mats.append(rng.randn(n, n))
    for _ in range(12):
        mats.append(rng.randint(-2, 3, size=(n, n)).astype(float))
    for _ in range(4):                                   # rate-matrix like: dominant diagonal
        a = -rng.rand(n, n) * 10.0 ** rng.uniform(-12, 0, size=(n, n))
        a[np.arange(n), np.arange(n)] = -a.sum(axis=0) + a.diagonal()
        mats.append(a)
    if n >= 8:
        for _ in range(4):        # exact ties at step 0: three rows share the largest |a(i,0)|
            a = rng.randn(n, n) * 0.3
            a[rng.choice(n - 1, 3, replace=False), 0] = 5.0 * rng.choice([-1.0, 1.0], 3)
            mats.append(a)
        for _ in range(4):        # exact tie at step 1 between row 0 -- moved to position r0 by the
            a = rng.randn(n, n) * 0.3          # interchange of step 0 -- and row 2: sgefa_ takes row 2
            r0 = int(rng.randint(3, n - 1))    # (lower POSITION), not the lower row index
            a[r0, 0] = 9.0
            a[0, :2] = a[2, :2] = (0.3, 7.0)
            mats.append(a)
    nexact = len(mats)
    for _ in range(4):                                   # exactly singular: a zero balance equation
        a = rng.randn(n, n)                              #   (stays zero through the elimination)
        a[rng.randint(0, n - 1)] = 0.0
        mats.append(a)
    A = np.array(mats)
    x, piv = eng.lubksb_batch(A, return_pivots=True)
    for m, a in enumerate(mats):
        want, info, ipvt = O.lubksb(a, return_ipvt=True)
        if m >= nexact:
            assert info != 0
            assert np.array_equal(x[m], want), (n, m, "singular system must return e_last")
            continue
        if info == 0:
            # integer output, bar = equality: the pivot ROW of every step (sgefa_'s ipvt replayed),
            # exact ties between candidates included.  Where FMA contraction can flip a comparison
            # between two nearly equal candidates the sequences may part; that needs |a|'s equal to
            # ~1e-16, which the gaussian and rate-like draws do not produce.
            order = list(range(n))
            rows = []
            for k in range(n):
                l = int(ipvt[k]) if k < n - 1 else n - 1
                order[k], order[l] = order[l], order[k]
                rows.append(order[k])
            if not (12 <= m < 24):        # (integer draws: rounding may turn a tie into a near-tie)
                assert list(piv[m]) == rows, (n, m, list(piv[m]), rows)
        aa = a.copy()
        aa[n - 1] = 1.0
        cond = np.linalg.cond(aa)
        if info != 0 or cond > 1e10:      # numerically singular integer draws: rounding decides, no parity claim
            continue
        err = np.max(np.abs(x[m] - want)) / np.max(np.abs(want))
        assert err <= 1e-13 * cond + 1e-12, (n, m, err, cond)


def test_solve_vs_reference_binary(engines, golden_dir, toy_path):
    g = json.load(open(os.path.join(golden_dir, "ref_matrix.json")))
    toy = {m: Engine(toy_path, escapeProbGeom=n) for m, n in ((1, "sphere"), (2, "lvg"))}
    worst = 0.0
    for c in g["cases"]:
        e = engines[c["method"]] if c["mol"] == "co_synth" else toy[c["method"]]
        e.set_source(c["tbg"])
        dens = [[c["density"].get(str(pid), 0.0) for pid in e.partner_ids]]
        r = e.solve_batch([c["tkin"]], [c["cdmol"]], dens)
        assert r["niter"][0] == c["niter"], (c["mol"], c["tkin"], r["niter"][0], c["niter"])
        assert r["status"][0] == (0 if c["conv"] else 1)
        x = np.array(c["xpop"])
        # populations: 1e-6 relative; below ~1e-8 the reference's own LU sits on its round-off
        # floor (absolute errors ~1e-16), hence the small absolute term
        ex = np.abs(r["xpop"][0] - x)
        assert np.all(ex <= 1e-6 * x + 1e-14), (c["mol"], c["method"], c["tkin"], np.max(ex / x))
        big = x > 1e-8
        dx = np.max(ex[big] / x[big])
        iu, il = e.iupp - 1, e.ilow - 1
        lines = (x[iu] > 1e-8) & (x[il] > 1e-8)
        tex, tau = np.array(c["tex"]), np.array(c["taul"])
        dt = np.max(np.abs(r["tex"][0][lines] - tex[lines]) / np.abs(tex[lines]))
        dtau = np.max(np.abs(r["tau"][0][lines] - tau[lines]) / np.abs(tau[lines]))
        worst = max(worst, dx, dt, dtau)
        assert dx < 1e-6 and dt < 1e-6 and dtau < 1e-6, (c["mol"], c["method"], c["tkin"], dx, dt, dtau)
    print("worst deviation from the reference binary: %.2e" % worst)


def _truth_source(eng, mol, cfg):
    src0 = O.Source(cfg["tbg"], cfg["Jup"], np.ones(len(cfg["Jup"])), np.ones(len(cfg["Jup"])),
                    cfg["bounds"], cfg["ncomp"], cfg["T_d"])
    truth_flux = O.model_flux_batch(mol, src0, cfg["truth"][None, :])[0][0]
    eflux = 0.1 * truth_flux
    eng.set_source(cfg["tbg"], cfg["Jup"], truth_flux, eflux, cfg["bounds"], cfg["ncomp"], cfg["T_d"])
    return O.Source(cfg["tbg"], cfg["Jup"], truth_flux, eflux, cfg["bounds"], cfg["ncomp"], cfg["T_d"])


def test_config2_flux_and_lnprob_vs_oracle(engines, mol):
    """BASELINE config 2: 1024 walkers uniform in the prior box, J=1..10, 1 component."""
    eng = engines[2]
    cfg = workloads.config2(1024)
    src = _truth_source(eng, mol, cfg)
    flux, st, nit = eng.model_flux_batch(cfg["walkers"], return_info=True)
    rf, rst, rnit = O.model_flux_batch(mol, src, cfg["walkers"], nthreads=8)
    assert np.array_equal(st, rst)
    assert (nit == rnit).mean() >= 0.995            # a convergence test may flip on the last bit
    ok, d = _flux_ok(flux, rf, cfg["walkers"], cfg["tbg"], mol)
    assert ok.all(), (np.argwhere(~ok)[:5], d[~ok][:5])
    sig = np.abs(rf) > 1e-6 * np.max(np.abs(rf), axis=1, keepdims=True)
    rel = d[sig] / np.abs(rf[sig])
    assert np.median(rel) < 1e-12 and np.percentile(rel, 99) < 1e-6
    # J-indexing: flux column j is line Jup[j]-1
    assert (st == 1).sum() > 0 and (st == 0).sum() > 900

    lnp, st2, nit2 = eng.lnprob_batch(cfg["walkers"], return_info=True)
    rl, rst2, rnit2 = O.lnprob_batch(mol, src, cfg["walkers"], nthreads=8)
    assert np.array_equal(st2, rst2)
    fin = np.isfinite(rl)
    assert np.array_equal(fin, np.isfinite(lnp))
    assert np.all(lnp[~fin] == -np.inf)
    rel = np.abs(lnp[fin] - rl[fin]) / np.maximum(np.abs(rl[fin]), 1.0)
    assert rel.max() < 1e-6, rel.max()


def test_prior_edges_invalid_and_floor(engines, mol):
    eng = engines[2]
    cfg = workloads.config2(8)
    src = _truth_source(eng, mol, cfg)
    b = cfg["bounds"]
    mid = 0.5 * (b[:, 0] + b[:, 1])
    P = np.tile(np.array([4.0, 1.8, 17.0, mid[3]]), (10, 1))
    P[1, 0] = b[0, 1] + 1e-9                 # above the box
    P[2, 1] = b[1, 0] - 1e-9                 # below the box
    P[3] = [4.0, 1.8, 14.0 + 4.0, mid[3]]; P[3, 2] = P[3, 0] + 10.0      # p2-p0 == 10.0 -> -inf
    P[4, 0], P[4, 2] = 2.0, 19.5             # p2-p0 == 17.5 -> -inf
    P[5, 2] = np.nan                         # NaN passes the prior, dies in lnlike
    P[6, 0] = b[0, 0]                        # exactly on the box edge is allowed
    lnp, st, _ = eng.lnprob_batch(P, return_info=True)
    rl, rst, _ = O.lnprob_batch(mol, src, P)
    assert np.array_equal(st, rst), (st, rst)
    assert np.array_equal(np.isfinite(lnp), np.isfinite(rl))
    assert list(st[1:5]) == [3, 3, 3, 3] and st[5] == 2 and st[0] == 0 and st[6] in (0, 1)
    # ValueError paths of the setters: T > 1e4 K, column > 1e25 (needs a wide box)
    wide = np.array([[-8, 12.0], [-1, 6.0], [0.0, 30.0], [-30, 0.0]])
    eng.set_source(cfg["tbg"], cfg["Jup"], src.flux, src.eflux, wide)
    srcw = O.Source(cfg["tbg"], cfg["Jup"], src.flux, src.eflux, wide)
    Q = np.array([[4.0, 4.5, 15.0, -10.0], [10.0, 2.0, 25.5, -10.0], [-6.0, 2.0, 4.9, -10.0], [4.0, 2.0, 16.0, -10.0]])
    lnp, st, _ = eng.lnprob_batch(Q, return_info=True)
    rl, rst, _ = O.lnprob_batch(mol, srcw, Q)
    assert list(st) == [2, 2, 2, 0] and np.array_equal(st, rst)
    flux, fst, _ = eng.model_flux_batch(Q, return_info=True)
    assert np.isnan(flux[:3]).all() and np.isfinite(flux[3]).all()
    # sigma floor: eflux = 0 -> e = 1e-12 (emcee_radex.py:147); non-finite data -> -inf everywhere
    eng.set_source(cfg["tbg"], cfg["Jup"], src.flux, np.zeros(10), wide)
    srcz = O.Source(cfg["tbg"], cfg["Jup"], src.flux, np.zeros(10), wide)
    lnp = eng.lnprob_batch(Q[3:])
    rl = O.lnprob_batch(mol, srcz, Q[3:])[0]
    assert np.isfinite(lnp[0]) and abs(lnp[0] - rl[0]) <= 1e-6 * abs(rl[0])
    bad = src.flux.copy(); bad[2] = np.nan
    eng.set_source(cfg["tbg"], cfg["Jup"], bad, src.eflux, wide)
    assert eng.lnprob_batch(Q[3:])[0] == -np.inf


def test_two_component_vs_oracle(engines, mol):
    eng = engines[2]
    cfg = workloads.config4(256)
    src = _truth_source(eng, mol, cfg)
    W = cfg["walkers"].copy()
    W[5, 5] = W[5, 1] - 0.01         # T2 <= T1            -> -inf
    W[6, 3] = W[6, 7] - 0.5          # size1 < size2       -> -inf
    W[7, 2] = W[7, 0] + 18.0         # N1-n1 >= 18         -> -inf
    lnp, st, nit = eng.lnprob_batch(W, return_info=True)
    rl, rst, rnit = O.lnprob_batch(mol, src, W, nthreads=8)
    assert np.array_equal(st, rst)
    assert (nit == rnit).mean() > 0.99
    fin = np.isfinite(rl)
    assert np.array_equal(fin, np.isfinite(lnp)) and not fin[5] and not fin[6] and not fin[7]
    rel = np.abs(lnp[fin] - rl[fin]) / np.maximum(np.abs(rl[fin]), 1.0)
    assert rel.max() < 1e-6, rel.max()
    flux = eng.model_flux_batch(cfg["walkers"][:64])
    rf = O.model_flux_batch(mol, src, cfg["walkers"][:64])[0]
    ok, d = _flux_ok(flux, rf, cfg["walkers"][:64], cfg["tbg"], mol, ncomp=2)
    assert ok.all()
    # T_d = None: flat prior on log T_cold too
    eng.set_source(cfg["tbg"], cfg["Jup"], src.flux, src.eflux, cfg["bounds"], 2, None)
    srcn = O.Source(cfg["tbg"], cfg["Jup"], src.flux, src.eflux, cfg["bounds"], 2, None)
    a = eng.lnprob_batch(cfg["walkers"][:64])
    b = O.lnprob_batch(mol, srcn, cfg["walkers"][:64])[0]
    fin = np.isfinite(b)
    assert fin.sum() > 10 and np.array_equal(fin, np.isfinite(a))
    assert np.max(np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1.0)) < 1e-6


def test_two_component_flux_is_sum_of_single_component_fluxes(engines, mol):
    """emcee_radex_2comp.py:144-147: intensity = comp1 + comp2, bit-exact on the same kernel."""
    eng = engines[2]
    c2, c1 = workloads.config4(64), workloads.config2(4)
    eng.set_source(c2["tbg"], c2["Jup"], np.ones(10), np.ones(10), c2["bounds"], 2, 40.0, src=0)
    eng.set_source(c2["tbg"], c2["Jup"], np.ones(10), np.ones(10), c1["bounds"], 1, None, src=1)
    f2 = eng.model_flux_batch(c2["walkers"], src=0)
    fa = eng.model_flux_batch(c2["walkers"][:, :4].copy(), src=1)
    fb = eng.model_flux_batch(c2["walkers"][:, 4:].copy(), src=1)
    assert np.array_equal(f2, fa + fb)


def test_multi_source_batch(engines, mol):
    """BASELINE config 3 in miniature: several sources advanced by one launch."""
    eng = engines[2]
    zs, jups = [3.6345, 3.0413, 2.951], [[3, 4, 5, 6, 7], [1, 3, 5, 8, 10], [5]]
    fl = [[5.699, 7.8, 9.734, 9.979, 7.962], [1.456, 7.008, 10.039, 9.3, 3.2], [9.89]]
    ef = [[2.248, 1.5, 1.188, 1.672, 0.915], [0.463, 1.193, 4.17, 0.4, 0.2], [0.618]]
    srcs, P, idx = [], [], []
    for k, z in enumerate(zs):
        b = workloads.bounds_1comp(z)
        tbg = workloads.T_CMB0 * (1 + z)
        eng.set_source(tbg, jups[k], fl[k], ef[k], b, src=k)
        srcs.append(O.Source(tbg, jups[k], fl[k], ef[k], b))
        P.append(workloads.draw_prior_1comp(b, 40, 100 + k))
        idx += [k] * 40
    P = np.concatenate(P)
    idx = np.array(idx, dtype=np.int32)
    perm = np.random.default_rng(5).permutation(len(P))
    lnp = eng.lnprob_batch(P[perm], src_index=idx[perm])
    ref = np.concatenate([O.lnprob_batch(mol, srcs[k], P[idx == k])[0] for k in range(3)])[perm]
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(lnp))
    assert np.max(np.abs(lnp[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1.0)) < 1e-6


def test_general_molecule_toy6(toy_path):
    """Non-ladder line list, one 'H2' partner, 6 levels -> the NL=8 kernel instantiation."""
    eng = Engine(toy_path)
    mol = O.Molecule(toy_path)
    assert eng.nlev == 6 and eng.nline == 7 and eng.partner_ids == [1]
    b = np.array([[2.0, 7.0], [0.5, 2.4], [12.0, 18.0], [-12.0, -8.0]])
    rng = np.random.default_rng(11)
    P = b[:, 0] + (b[:, 1] - b[:, 0]) * rng.random((200, 4))
    P[:, 2] = np.clip(P[:, 2], P[:, 0] + 10.01, P[:, 0] + 17.49)
    P[:, 2] = np.clip(P[:, 2], 12.0, 18.0)
    Jup = [1, 2, 3, 5, 7]
    eng.set_source(2.73, Jup, np.ones(5), 0.3 * np.ones(5), b)
    src = O.Source(2.73, Jup, np.ones(5), 0.3 * np.ones(5), b)
    flux, st, nit = eng.model_flux_batch(P, return_info=True)
    rf, rst, rnit = O.model_flux_batch(mol, src, P)
    assert np.array_equal(st, rst) and (nit == rnit).mean() > 0.99
    ok, d = _flux_ok(flux, rf, P, 2.73, mol)
    assert ok.all()
    lnp = eng.lnprob_batch(P)
    rl = O.lnprob_batch(mol, src, P)[0]
    fin = np.isfinite(rl)
    assert np.array_equal(fin, np.isfinite(lnp))
    assert np.max(np.abs(lnp[fin] - rl[fin]) / np.maximum(np.abs(rl[fin]), 1.0)) < 1e-6


def test_full_size_properties(engines, mol):
    """Size-independent properties at the stress size (65536 walkers)."""
    import torch
    eng = engines[2]
    cfg = workloads.config2(65536, seed=5678)
    _truth_source(eng, mol, cfg)
    dev = torch.device("cuda:0")
    P = torch.from_numpy(cfg["walkers"]).to(dev)
    a, sa, na = eng.lnprob_batch_torch(P)
    torch.cuda.synchronize()
    b, sb, nb = eng.lnprob_batch_torch(P)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(sa, sb) and torch.equal(na, nb)       # deterministic
    perm = torch.randperm(P.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    c, sc, nc = eng.lnprob_batch_torch(P[perm].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(c, a[perm]) and torch.equal(nc, na[perm])                   # order independent
    sub, _, _ = eng.lnprob_batch_torch(P[1000:1512].contiguous())                  # batch-size independent
    torch.cuda.synchronize()
    assert torch.equal(sub, a[1000:1512])
    an = a.cpu().numpy()
    assert np.isfinite(an).mean() > 0.9 and (na >= 11).all() and (na <= 200).all()
    # spot-check 256 of them against the oracle
    pick = np.random.default_rng(9).choice(P.shape[0], 256, replace=False)
    src = _truth_source(eng, mol, cfg)
    rl = O.lnprob_batch(mol, src, cfg["walkers"][pick], nthreads=8)[0]
    fin = np.isfinite(rl)
    assert np.max(np.abs(an[pick][fin] - rl[fin]) / np.maximum(np.abs(rl[fin]), 1.0)) < 1e-6


def test_issue_order_changes_no_result(engines, mol):
    """Batches larger than the resident wavefronts are handed out hottest walkers first
    (rx_order_*_kernel): a scheduling decision -- every output must be bit-identical without it."""
    eng = engines[2]
    cfg = workloads.config2(6000, seed=77)
    src = _truth_source(eng, mol, cfg)
    eng.set_issue_order(True)
    lnp, st, nit = eng.lnprob_batch(cfg["walkers"], return_info=True)
    eng.set_issue_order(False)
    lnp0, st0, nit0 = eng.lnprob_batch(cfg["walkers"], return_info=True)
    eng.set_issue_order(True)
    assert np.array_equal(st, st0) and np.array_equal(nit, nit0)
    assert np.array_equal(lnp, lnp0, equal_nan=True)
    assert (st == 1).sum() > 0


def test_two_handles_on_two_streams(engines, mol):
    """The ABI allows several handles to be used concurrently (one per source / per chain): two engines,
    two HIP streams, launches in flight together -- each must return what it returns alone."""
    import torch
    from radex_emcee_amd.engine import Engine
    eng = engines[2]
    cfgA, cfgB = workloads.config2(1024, seed=11), workloads.config2(1024, seed=12)
    _truth_source(eng, mol, cfgA)
    dev = torch.device("cuda:0")
    PA, PB = (torch.from_numpy(c["walkers"]).to(dev) for c in (cfgA, cfgB))
    refA = [t.clone() for t in eng.lnprob_batch_torch(PA)]
    refB = [t.clone() for t in eng.lnprob_batch_torch(PB)]
    torch.cuda.synchronize()
    eng2 = Engine(eng.molfile)
    src = _truth_source(eng2, mol, cfgA)
    sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outA = [torch.empty(1024, dtype=t, device=dev) for t in (torch.float64, torch.int32, torch.int32)]
    outB = [torch.empty(1024, dtype=t, device=dev) for t in (torch.float64, torch.int32, torch.int32)]
    for _ in range(5):
        eng.lnprob_batch_torch(PA, *outA, stream=sA.cuda_stream)
        eng2.lnprob_batch_torch(PB, *outB, stream=sB.cuda_stream)
    torch.cuda.synchronize()
    for got, want in zip(outA + outB, refA + refB):
        assert torch.equal(got, want) or (got.dtype == torch.float64 and torch.equal(torch.nan_to_num(got, neginf=-1e308), torch.nan_to_num(want, neginf=-1e308)))
    eng2.close()


def test_device_pointer_api_on_side_stream(engines, mol):
    import torch
    eng = engines[2]
    cfg = workloads.config2(512)
    _truth_source(eng, mol, cfg)
    dev = torch.device("cuda:0")
    host = eng.lnprob_batch(cfg["walkers"])
    s = torch.cuda.Stream(device=dev)
    P = torch.from_numpy(cfg["walkers"]).to(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        lnp, st, nit = eng.lnprob_batch_torch(P, stream=s.cuda_stream)
    s.synchronize()
    assert np.array_equal(lnp.cpu().numpy(), host)
    ms = eng.time_lnprob_torch(P, lnp, st, nit, reps=2)
    assert 0.0 < ms < 1000.0


@pytest.mark.parametrize("nlev", [5, 8, 12, 20, 27, 32, 33, 45, 48, 64])
def test_every_kernel_instantiation(nlev, tmp_path):
    """Synthetic rotor ladders of different sizes exercise each general instantiation (NL = 8, 20, 32, 48, 64), with padding
    levels (5, 12, 27, 33, 45) and filled exactly (8, 20, 32, 48, 64), against the oracle.  (The specialised ladder form exists
    for the 41 levels of CO only and is what every other test runs; the general 41-level form runs in
    test_line_order_that_is_not_a_ladder_... and in the sphere / slab tests.)"""
    from radex_emcee_amd.molecule import synth_co_text
    path = tmp_path / ("rotor%d.dat" % nlev)
    path.write_text(synth_co_text(nlev=nlev))
    eng = Engine(str(path))
    mol = O.Molecule(str(path))
    assert eng.nlev == nlev and eng.nline == nlev - 1
    z = 2.5
    b = workloads.bounds_1comp(z)
    tbg = workloads.T_CMB0 * (1 + z)
    P = workloads.draw_prior_1comp(b, 48, 900 + nlev)
    Jup = [j for j in (1, 2, 3, 4) if j < nlev]
    eng.set_source(tbg, Jup, np.ones(len(Jup)), 0.2 * np.ones(len(Jup)), b)
    src = O.Source(tbg, Jup, np.ones(len(Jup)), 0.2 * np.ones(len(Jup)), b)
    flux, st, nit = eng.model_flux_batch(P, return_info=True)
    rf, rst, rnit = O.model_flux_batch(mol, src, P)
    assert np.array_equal(st, rst), nlev
    assert (nit == rnit).mean() >= 0.95
    same = nit == rnit
    ok, d = _flux_ok(flux[same], rf[same], P[same], tbg, mol)
    assert ok.all(), (nlev, d[~ok][:4])
    lnp = eng.lnprob_batch(P)
    rl = O.lnprob_batch(mol, src, P)[0]
    fin = np.isfinite(rl) & same
    assert np.array_equal(np.isfinite(rl), np.isfinite(lnp))
    assert np.max(np.abs(lnp[fin] - rl[fin]) / np.maximum(np.abs(rl[fin]), 1.0)) < 1e-6


def test_line_order_that_is_not_a_ladder_runs_the_general_form(tmp_path):
    """The ladder form of the 41-level instantiation assumes line l = (level l+1 -> level l) in FILE order (a line and its
    lower level share a lane).  The same molecule with two lines listed the other way round must not take it: it runs the general
    instantiation (incidence lists through LDS), and matches the oracle on THAT file -- the accumulation order of matrix_ follows
    the line order, so the sums differ in the last bits from the ladder file's, and `niter` of a walker on a threshold may too."""
    from radex_emcee_amd.molecule import synth_co_text
    lines = synth_co_text(nlev=41).split("\n")
    i0 = next(k for k, l in enumerate(lines) if l.startswith("!TRANS + UP + LOW + EINSTEINA")) + 1
    a, b = lines[i0 + 6].split(), lines[i0 + 7].split()
    a[0], b[0] = b[0], a[0]                                   # (the running index stays 1..40)
    lines[i0 + 6], lines[i0 + 7] = " ".join(b), " ".join(a)
    path = tmp_path / "co_swapped.dat"
    path.write_text("\n".join(lines))
    eng = Engine(str(path))
    mol = O.Molecule(str(path))
    assert eng.nlev == 41 and eng.nline == 40 and eng.kernel_name == "rx_solve_kernel<41, 1, false>"
    cfg = workloads.config2(256, seed=4242)
    eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), 0.2 * np.ones(10), cfg["bounds"])
    src = O.Source(cfg["tbg"], cfg["Jup"], np.ones(10), 0.2 * np.ones(10), cfg["bounds"])
    P = cfg["walkers"]
    flux, st, nit = eng.model_flux_batch(P, return_info=True)
    rf, rst, rnit = O.model_flux_batch(mol, src, P)
    assert np.array_equal(st, rst) and (nit == rnit).mean() >= 0.98
    same = nit == rnit
    ok, d = _flux_ok(flux[same], rf[same], P[same], cfg["tbg"], mol)
    assert ok.all(), d[~ok][:4]
    # and the two files are the same physics: against the ladder file's engine the fluxes agree to rounding
    eng0 = Engine()
    assert eng0.kernel_name == "rx_solve_kernel<41, 1, true>"
    eng0.set_refinement(False)            # (the general form pivots every solve: rounding-level agreement needs the ladder form to do so too)
    eng0.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), 0.2 * np.ones(10), cfg["bounds"])
    f0, st0, nit0 = eng0.model_flux_batch(P, return_info=True)
    both = (nit == nit0) & (st == 0) & (st0 == 0)
    assert both.mean() > 0.9
    # (model_lvg picks result[Jup - 1] by POSITION in the file, emcee/emcee_radex.py:129: the swapped file's columns for
    # Jup = 7 and 8 are each other's lines)
    col = list(cfg["Jup"])
    k7, k8 = col.index(7), col.index(8)
    f0[:, [k7, k8]] = f0[:, [k8, k7]]
    assert np.nanmax(np.abs(flux[both] - f0[both]) / np.maximum(np.abs(f0[both]), 1e-300)) < 1e-6
    eng.close(); eng0.close()


def test_batch_shapes_and_limits(engines, mol):
    eng = engines[2]
    cfg = workloads.config2(1030)
    src = _truth_source(eng, mol, cfg)
    full = eng.lnprob_batch(cfg["walkers"])
    assert eng.lnprob_batch(np.empty((0, 4))).shape == (0,)                # empty batch
    for n in (1, 3, 5, 255, 257, 1030):                                     # ragged sizes, tail of the work queue
        assert np.array_equal(eng.lnprob_batch(cfg["walkers"][:n]), full[:n]), n
    # nJ = 0: likelihood is the prior only (chi2 = 0, log term = 0)
    eng.set_source(cfg["tbg"], [], [], [], cfg["bounds"])
    lp = eng.lnprob_batch(cfg["walkers"][:8])
    assert np.all(lp == 0.0)
    # all source slots can be resident at once; out-of-range slots are rejected by the host API
    for k in range(64):
        eng.set_source(cfg["tbg"] * (1 + 0.001 * k), cfg["Jup"], src.flux, src.eflux, cfg["bounds"], src=k)
    idx = (np.arange(128) % 64).astype(np.int32)
    a = eng.lnprob_batch(cfg["walkers"][:128], src_index=idx)
    assert np.isfinite(a).sum() > 100 and not np.array_equal(a[:64], a[64:128])
    from radex_emcee_amd.engine import EngineError
    with pytest.raises(EngineError):
        eng.set_source(cfg["tbg"], cfg["Jup"], src.flux, src.eflux, cfg["bounds"], src=64)
    with pytest.raises(EngineError):
        eng.set_source(cfg["tbg"], [99], [1.0], [1.0], cfg["bounds"])       # Jup beyond the line list
    with pytest.raises(EngineError):
        eng.set_source(-1.0, cfg["Jup"], src.flux, src.eflux, cfg["bounds"])
    with pytest.raises(EngineError):
        eng.lnprob_batch(cfg["walkers"][:4], src_index=np.array([0, 1, 2, 77], dtype=np.int32))
