"""CPU tests of the host-side logic and of the oracle's driver restatement
(lnprior/lnlike/lnprob/model_lvg of emcee_radex.py and emcee_radex_2comp.py).  The reference has
no tests for these functions; expected values are hand-derived from the cited lines."""
import hashlib
import math
import os

import numpy as np
import pytest

from oracle import oracle as O
from radex_emcee_amd import cosmology, molecule, workloads


def test_cosmology_golden_values():
    # astropy 4.3.1 FlatLambdaCDM(H0=67.8, Om0=0.308), SURVEY.md section 8c
    gold = {3.6345: (1525.2170523973487, -9.17931735162758, 1.1024040913453224),
            2.0924: (1760.763378351088, -9.304056041005104, 0.9266968694491308),
            3.911: (1481.9490657587899, -9.15432060187837, 1.1275711389822052),
            4.243: (1431.8648915206384, -9.124458128423141, 1.1559810625643245)}
    for z, (da, lra, ltcmb) in gold.items():
        assert abs(cosmology.angular_diameter_distance(z) - da) < 1e-9
        assert abs(cosmology.log10_R_angle(z) - lra) < 1e-13
        assert abs(math.log10(2.7315 * (1 + z)) - ltcmb) < 1e-15
        b = workloads.bounds_1comp(z)
        assert b[3, 0] == pytest.approx(lra - 4) and b[3, 1] == pytest.approx(lra + 4)
        b2 = workloads.bounds_2comp(z)
        assert b2.shape == (8, 2) and b2[7, 1] == pytest.approx(lra + 9) and b2[4, 0] == 1.5


def test_synthetic_molecule_is_deterministic(co_path):
    txt = molecule.synth_co_text()
    assert open(molecule.SYNTH_CO_PATH).read() == txt
    m = O.Molecule(co_path)
    assert (m.nlev, m.nline, m.npart, m.part_id) == (41, 40, 2, [2, 3])
    assert m.xnu[0] == pytest.approx(3.845033419, abs=1e-9)        # CO 1-0
    assert np.all(m.iupp == np.arange(2, 42)) and np.all(m.ilow == np.arange(1, 41))


def test_workload_shapes():
    c2 = workloads.config2(1024)
    assert c2["walkers"].shape == (1024, 4) and list(c2["Jup"]) == list(range(1, 11))
    d = c2["walkers"][:, 2] - c2["walkers"][:, 0]
    assert np.all((d > 10.0) & (d < 17.5))
    assert np.all(c2["walkers"] >= c2["bounds"][:, 0]) and np.all(c2["walkers"] <= c2["bounds"][:, 1])
    assert hashlib.sha1(c2["walkers"].tobytes()).hexdigest() == hashlib.sha1(workloads.config2(1024)["walkers"].tobytes()).hexdigest()
    c4 = workloads.config4(2048)
    assert c4["walkers"].shape == (2048, 8) and c4["ncomp"] == 2 and c4["T_d"] == 40.0


@pytest.fixture(scope="module")
def mol(co_path):
    return O.Molecule(co_path)


def _src(ncomp=1, T_d=None, eflux=None):
    z = 2.5
    b = workloads.bounds_1comp(z) if ncomp == 1 else workloads.bounds_2comp(z)
    Jup = np.array([1, 3, 5, 8, 10], dtype=np.int32)
    flux = np.array([1.456, 7.008, 10.039, 9.3, 3.2])
    e = np.array([0.463, 1.193, 4.17, 0.4, 0.2]) if eflux is None else eflux
    return O.Source(2.7315 * (1 + z), Jup, flux, e, b, ncomp, T_d), b


def test_lnprior_1comp_edges():
    src, b = _src()
    mid = [4.0, 1.8, 17.0, 0.5 * (b[3, 0] + b[3, 1])]
    assert O.lnprior(src, mid) == 0.0
    assert O.lnprior(src, [b[0, 0], 1.8, 15.5, mid[3]]) == 0.0           # on the box edge: allowed
    for k in range(4):
        p = list(mid); p[k] = b[k, 1] + 1e-12
        assert O.lnprior(src, p) == -np.inf
        p = list(mid); p[k] = b[k, 0] - 1e-12
        assert O.lnprior(src, p) == -np.inf
    assert O.lnprior(src, [6.0, 1.8, 16.0, mid[3]]) == -np.inf           # p2-p0 == 10.0 (<=)
    assert O.lnprior(src, [2.0, 1.8, 19.5, mid[3]]) == -np.inf           # p2-p0 == 17.5 (>=)
    assert O.lnprior(src, [2.0, 1.8, 19.4999, mid[3]]) == 0.0
    assert O.lnprior(src, [4.0, 1.8, float("nan"), mid[3]]) == 0.0       # NaN slips through the prior


def test_lnprior_2comp_formula():
    T_d = 40.0
    src, b = _src(2, T_d)
    p = np.array([1.9, 1.2, 16.4, -12.1, 3.9, 2.5, 17.5, -12.1])
    flat = sum(-(b[k, 1] - b[k, 0]) for k in range(8) if k != 1)
    gauss = -0.5 * ((10 ** 1.2 - T_d) / T_d) ** 2 - math.log(T_d * math.sqrt(2 * math.pi))
    assert O.lnprior(src, p) == pytest.approx(flat + gauss, rel=1e-14)
    q = p.copy(); q[5] = q[1]
    assert O.lnprior(src, q) == -np.inf                                  # T2 <= T1
    q = p.copy(); q[3] = q[7] - 1e-9
    assert O.lnprior(src, q) == -np.inf                                  # size1 < size2
    q = p.copy(); q[3] = q[7]
    assert np.isfinite(O.lnprior(src, q))                                # size1 == size2 allowed
    q = p.copy(); q[2] = q[0] + 9.0
    assert O.lnprior(src, q) == -np.inf
    q = p.copy(); q[6] = q[4] + 18.0
    assert O.lnprior(src, q) == -np.inf
    srcn, _ = _src(2, None)
    assert O.lnprior(srcn, p) == pytest.approx(flat - (b[1, 1] - b[1, 0]), rel=1e-14)
    src0, _ = _src(2, 0.0)
    assert O.lnprior(src0, p) == -np.inf                                 # T_d <= 0


def test_lnlike_and_unit_factor(mol):
    src, b = _src()
    p = np.array([[3.5, 2.0, 17.5, -9.5]])
    f, st, nit = O.model_flux_batch(mol, src, p)
    # flux = S[Jup-1] * 10**log_size * 1e23 (emcee_radex.py:129), J-indexing via Jup-1
    s = O.State(mol); s.backrad(src.tbg)
    s.set_density({2: 0.25 * 10 ** 3.5, 3: 0.75 * 10 ** 3.5}); s.s.tkin = 100.0; s.s.cdmol = 10 ** 17.5
    s.rates(); it, conv = s.run()
    sb = s.surfbrightness()
    assert nit[0] == it
    want = sb[src.Jup - 1] * 10 ** -9.5 * 1.0 * 1e23
    assert np.array_equal(f[0], want)
    lnp, st, _ = O.lnprob_batch(mol, src, p)
    e = np.maximum(np.abs(src.eflux), 1e-12)
    r = (src.flux - want) / e
    assert lnp[0] == pytest.approx(-0.5 * (np.dot(r, r) + 2 * np.sum(np.log(e))), rel=1e-13)
    # sigma floor 1e-12
    srcz, _ = _src(eflux=np.zeros(5))
    lz = O.lnprob_batch(mol, srcz, p)[0][0]
    rz = (src.flux - want) / 1e-12
    assert lz == pytest.approx(-0.5 * (np.dot(rz, rz) + 2 * 5 * math.log(1e-12)), rel=1e-13)
    # setter ValueErrors -> -inf (T > 1e4, column out of range)
    wide = np.array([[-8, 12.0], [-1, 6.0], [0.0, 30.0], [-30, 0.0]])
    srcw = O.Source(src.tbg, src.Jup, src.flux, src.eflux, wide)
    Q = np.array([[4.0, 4.5, 15.0, -10.0], [10.0, 2.0, 25.5, -10.0], [-6.0, 2.0, 4.9, -10.0]])
    lnp, st, _ = O.lnprob_batch(mol, srcw, Q)
    assert np.all(lnp == -np.inf) and list(st) == [2, 2, 2]


def test_oracle_physical_limits(mol):
    """Analytic limits the restated RADEX must hit (golden kind ii)."""
    # LTE at very high density, optically thin: Tex -> Tkin
    r = O.solve_state(mol, 2.73, {2: 0.25e11, 3: 0.75e11}, 40.0, 1e10)
    assert np.allclose(r["tex"][:8], 40.0, rtol=2e-4)
    # no collisions worth mentioning: Tex -> T_bg
    r = O.solve_state(mol, 2.73, {2: 0.25e-4, 3: 0.75e-4}, 40.0, 1e10)
    assert np.allclose(r["tex"][:3], 2.73, rtol=1e-3)
    assert abs(r["xpop"].sum() - 1.0) < 1e-12
    # detailed balance of the interpolated rates
    st = O.State(mol); st.set_density({2: 1e3, 3: 3e3}); st.s.tkin = 77.0; st.rates()
    cr = st.arr("crate").reshape(41, 41)
    FK = 1.4387809925261357
    for (u, l) in ((1, 0), (5, 2), (20, 19)):
        want = mol.gstat[u] / mol.gstat[l] * math.exp(-FK * (mol.eterm[u] - mol.eterm[l]) / 77.0) * cr[u, l]
        assert cr[l, u] == pytest.approx(want, rel=1e-14)
    assert np.allclose(st.arr("ctot"), cr.sum(1), rtol=1e-14)


def test_reference_kats_when_real_co_dat_is_supplied():
    """emcee/pyradex/tests/test_radex.py:99-115, 175-200 need the LAMDA co.dat that the
    reference does not ship; activated by RADEX_DATAPATH."""
    dp = os.getenv("RADEX_DATAPATH")
    if not dp or not os.path.exists(os.path.join(dp, "co.dat")):
        pytest.skip("real LAMDA co.dat not available (set RADEX_DATAPATH)")
    m = O.Molecule(os.path.join(dp, "co.dat"))
    opr = min(3.0, 9.0 * math.exp(-170.6 / 30.0)); fo = opr / (1 + opr)
    r = O.solve_state(m, 2.73, {2: 1e4 * (1 - fo), 3: 1e4 * fo}, 30.0, 1e14)
    assert r["tex"][0] == pytest.approx(56.131, rel=1e-4)
    assert r["tau"][0] == pytest.approx(1.786e-3, rel=1e-3)


def test_linpack_positions_can_be_replayed_from_the_pivot_history():
    """The kernel moves pivot rows physically into lane k and does not track the position LINPACK's
    sgefa_ would have each row at; pivot_exact (rx_kernel.hip.inc) replays the positions from the
    pivot history when two candidates tie exactly.  Same replay in Python against a plain
    simulation of sgefa_'s interchanges (oracle/radex_oracle.c lin_gefa)."""
    rng = np.random.RandomState(5)
    for n in (3, 7, 41, 64):
        for _ in range(50):
            # an arbitrary pivot history: the pivot ROW of each step (any permutation is reachable)
            order = list(rng.permutation(n))
            for kk in (0, 1, n // 2, n - 1):
                # plain sgefa_: rows start at position = index; step j swaps positions j and l
                at = list(range(n))                   # at[position] = row
                for j in range(kk):
                    l = at.index(order[j])
                    at[j], at[l] = at[l], at[j]
                want = {row: posn for posn, row in enumerate(at)}
                # kernel-style replay: lane j < kk holds the pivot row of step j, the other rows
                # sit in arbitrary lanes >= kk (here: in a random arrangement)
                rest = [r for r in range(n) if r not in order[:kk]]
                rng.shuffle(rest)
                lrow = order[:kk] + rest              # lrow[lane] = row held by that lane
                pos = list(lrow)
                for j in range(kk):
                    l = pos[j]
                    pos = [l if p == j else p for p in pos]
                    pos[j] = j
                got = {lrow[lane]: pos[lane] for lane in range(n)}
                assert got == want
