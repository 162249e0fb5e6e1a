"""GPU tests added in round 2: BASELINE config 3 in one launch, the issue order with per-walker sources and
with two components, full-width parity at the stress sizes, the hardened device-pointer boundary, the
on-device stretch move against its host checker, and the reference's known-answer tests when a real
LAMDA co.dat is supplied.  Everything goes through the C ABI; the oracle is only the checker.

Tolerances: as tests/test_gpu_parity.py (flux 1e-4 relative + the background floor; status codes
equal).  Walkers that run into maxiter amplify round-off over their 200 iterations: their
deviation is reported separately and bounded more loosely than that of converged walkers."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O                      # noqa: E402  (checker only)
from radex_emcee_amd import workloads               # noqa: E402
from radex_emcee_amd.engine import Engine           # noqa: E402
from test_gpu_parity import _flux_ok, _truth_source   # noqa: E402

NTH = max(1, min(32, len(os.sched_getaffinity(0))))
RX_OK, RX_MAXITER, RX_INVALID, RX_PRIOR = 0, 1, 2, 3


@pytest.fixture(scope="module")
def eng(co_path):
    return Engine(co_path)


@pytest.fixture(scope="module")
def mol(co_path):
    return O.Molecule(co_path)


def _rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0)


# The ceiling of a walker that stops at maxiter is derived from the REFERENCE BINARY, not chosen: tests/golden/ref_sensitivity.npz
# (scripts/ref_sensitivity.py, container only) holds for every such walker of the batches below how far radex.so's own answer
# moves when its exp / log are one ulp off -- in units of the flux tolerance (resp_sb) and as relative lnlike (resp_lnp).  A GPU
# walker may deviate by  K_SENS x that response + north_star's 1e-4;  everywhere else the plain tolerance holds.  Measured over sixteen
# draws of 131 072 prior-box walkers (49 947 at maxiter, profiles/r6_maxiter_vs_ref_sensitivity.txt): the GPU has 8 walkers beyond
# the flux tolerance (worst 191 x, lnprob 1.2e-3) -- all 8 rank first to third among the ~3100 binary responses of their draw (the
# binary itself moves 13 walkers beyond the tolerance, worst 29 x), and the smallest K that covers every walker is 6.6.
K_SENS = 20.0
_SENS = {}


def _sensitivity(batch, walkers, missing_ok=False):
    """(resp_sb, resp_lnp) of the reference binary for the given walker indices of a batch.  All of them must be in the fixture
    unless missing_ok: a walker it does not hold (one the prior rejects, which only the flux call ever solves) gets response 0,
    i.e. the plain tolerance."""
    if not _SENS:
        f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_sensitivity.npz"))
        _SENS.update({k: f[k] for k in f.files})
    w = _SENS[batch + "_walker"]
    walkers = np.asarray(walkers, dtype=np.int64)
    pos = np.minimum(np.searchsorted(w, walkers), len(w) - 1)
    have = w[pos] == walkers
    assert len(w) and (missing_ok or have.all()), "maxiter walkers of %s missing from ref_sensitivity.npz" % batch
    return (np.where(have, _SENS[batch + "_resp_sb"][pos], 0.0).astype(np.float64),
            np.where(have, _SENS[batch + "_resp_lnp"][pos], 0.0).astype(np.float64))


def _maxiter_ceiling(batch, walkers):
    """per-walker bound on the relative lnprob deviation of walkers that stop at maxiter"""
    return K_SENS * _sensitivity(batch, walkers)[1] + 1e-4


def _report(tag, lnp, ref, st):
    """max relative lnprob deviation, separately for converged and maxiter walkers"""
    fin = np.isfinite(ref)
    ok, mx = fin & (st == RX_OK), fin & (st == RX_MAXITER)
    dok = _rel(lnp[ok], ref[ok]).max() if ok.any() else 0.0
    dmx = _rel(lnp[mx], ref[mx]) if mx.any() else np.zeros(1)
    # (the distribution goes into the pytest log -- GPUTEST carries it every round.  The maxiter tier has a heavy tail: these
    # walkers' iterations never settle, a few of them are chaotic, and ANY two implementations of the same arithmetic end up apart
    # there -- over 16 draws of 131 072 walkers (profiles/r5_big_parity_seeds.txt, r5_big_parity_seeds_norefine.txt) 5 of the
    # 49 800 maxiter walkers are beyond 1e-4 with the refinement and 9 without it, the worst at 1.2e-3 and 9.8e-4.  The tier is
    # therefore asserted as a distribution close to what is observed -- a regression of an order of magnitude fails -- plus, per
    # walker, K_SENS x the reference binary's own response to a 1-ulp libm perturbation + 1e-4: _maxiter_ceiling.)
    print("\n[%s] %d walkers: converged %d (max rel dev of lnprob %.2e); maxiter %d: 99th pct %.2e, 99.9th pct %.2e, "
          "max %.2e, above 1e-4: %d" % (tag, len(ref), ok.sum(), dok, mx.sum(), np.percentile(dmx, 99),
                                        np.percentile(dmx, 99.9), dmx.max(), int((dmx > 1e-4).sum())))
    return dok, dmx


def _set_config3(eng, cfg):
    srcs = []
    for s in cfg["sources"]:
        eng.set_source(s["tbg"], s["Jup"], s["flux"], s["eflux"], s["bounds"], src=s["slot"])
        srcs.append(O.Source(s["tbg"], s["Jup"], s["flux"], s["eflux"], s["bounds"]))
    return srcs


def test_config3_sixteen_sources_one_launch(eng, mol):
    """BASELINE configs[2]: the 16 sources of flux.dat with their own tbg / bounds / line lists, 512
    proposals each = one launch of 8192 walkers with a per-walker source slot; the batch is larger than
    twice the resident wavefronts, so the hottest-first issue order is live together with src_index
    [/root/reference/emcee/emcee_radex.py:389-442, data/flux.dat:8-23]."""
    cfg = workloads.config3(512)
    srcs = _set_config3(eng, cfg)
    P = cfg["walkers"].reshape(-1, 4)
    idx = cfg["src_index"]
    assert len(P) == 8192
    # interleave the sources: neighbouring walkers belong to different sources
    perm = np.random.default_rng(8).permutation(len(P))
    eng.set_issue_order(True)
    lnp, st, nit = eng.lnprob_batch(P[perm], src_index=idx[perm], return_info=True)
    eng.set_issue_order(False)
    lnp0, st0, nit0 = eng.lnprob_batch(P[perm], src_index=idx[perm], return_info=True)
    eng.set_issue_order(True)
    assert np.array_equal(lnp, lnp0, equal_nan=True) and np.array_equal(st, st0) and np.array_equal(nit, nit0)
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    lnp, st, nit = lnp[inv], st[inv], nit[inv]
    for k, s in enumerate(cfg["sources"]):
        sl = slice(512 * k, 512 * (k + 1))
        rl, rst, rnit = O.lnprob_batch(mol, srcs[k], P[sl], nthreads=NTH)
        assert np.array_equal(st[sl], rst), s["name"]
        assert (nit[sl] == rnit).mean() >= 0.995, s["name"]
        fin = np.isfinite(rl)
        assert np.array_equal(fin, np.isfinite(lnp[sl]))
        okm = fin & (rst == RX_OK)
        assert _rel(lnp[sl][okm], rl[okm]).max() < 1e-6, s["name"]
        mx = fin & (rst == RX_MAXITER)
        if mx.any():
            dsrc = _rel(lnp[sl][mx], rl[mx])                              # (observed: 1e-8 at worst)
            assert np.median(dsrc) < 1e-6 and (dsrc <= _maxiter_ceiling("config3_512", 512 * k + np.flatnonzero(mx))).all(), s["name"]
        # fluxes of this source's walkers (its own line list) against the oracle
        flux, fst, fnit = eng.model_flux_batch(P[sl], src=s["slot"], return_info=True)
        rf, rfst, _ = O.model_flux_batch(mol, srcs[k], P[sl], nthreads=NTH)
        assert flux.shape == (512, len(s["Jup"])) and np.array_equal(fst, rfst)
        ok, d = _flux_ok(flux, rf, P[sl], s["tbg"], mol)
        assert ok.all(), (s["name"], np.argwhere(~ok)[:5], d[~ok][:5])
    assert (st == RX_MAXITER).sum() > 0 and (st == RX_OK).sum() > 7000


def test_two_component_issue_order_against_oracle(eng, mol):
    """with the issue order active (N*ncomp items > twice the resident wavefronts) a
    2-component walker must receive ITS components' fluxes: with and without the order, and the oracle."""
    cfg = workloads.config4(4096)
    W = cfg["walkers"].copy()
    W[2048:] = workloads.draw_prior_2comp(cfg["bounds"], 2048, 91)     # spread over the box: the order permutes
    src = _truth_source(eng, mol, cfg)
    eng.set_issue_order(True)
    lnp, st, nit = eng.lnprob_batch(W, return_info=True)
    eng.set_issue_order(False)
    lnp0, st0, nit0 = eng.lnprob_batch(W, return_info=True)
    eng.set_issue_order(True)
    assert np.array_equal(st, st0) and np.array_equal(nit, nit0)
    assert np.array_equal(lnp, lnp0, equal_nan=True)
    rl, rst, rnit = O.lnprob_batch(mol, src, W, nthreads=NTH)
    assert np.array_equal(st, rst)
    assert (nit == rnit).mean() >= 0.995
    fin = np.isfinite(rl)
    assert np.array_equal(fin, np.isfinite(lnp)) and fin.sum() > 2000
    dok, dmx = _report("2-comp, 4096 walkers, issue order on", lnp, rl, rst)
    assert dok < 1e-6 and np.percentile(dmx, 99) < 1e-6                                  # (observed: 4.8e-10, 2.2e-8)
    assert (dmx <= _maxiter_ceiling("config4_4096_mixed", np.flatnonzero(fin & (rst == RX_MAXITER)))).all()
    flux = eng.model_flux_batch(W[2040:2300])
    rf = O.model_flux_batch(mol, src, W[2040:2300], nthreads=NTH)[0]
    ok, d = _flux_ok(flux, rf, W[2040:2300], cfg["tbg"], mol, ncomp=2)
    assert ok.all()


def test_full_width_parity_config5(eng, mol):
    """The stress shape at full width: 65536 config-5 walkers against the oracle on every host thread.
    Status must be identical; converged walkers within the flux-level tolerance (1e-4) on lnprob -- in
    practice 1e-6; walkers at maxiter are reported, not hidden."""
    cfg = workloads.config2(65536, seed=5678)
    src = _truth_source(eng, mol, cfg)
    lnp, st, nit = eng.lnprob_batch(cfg["walkers"], return_info=True)
    rl, rst, rnit = O.lnprob_batch(mol, src, cfg["walkers"], nthreads=NTH)
    assert np.array_equal(st, rst)
    same = (nit == rnit)
    print("\niteration counts identical: %d of %d" % (same.sum(), len(same)))
    assert same.mean() >= 0.999
    fin = np.isfinite(rl)
    assert np.array_equal(fin, np.isfinite(lnp))
    dok, dmx = _report("config 5, 65536 walkers", lnp, rl, rst)
    assert dok < 1e-4
    # maxiter tier (see _report): observed here 99th percentile 2.0e-8, 99.9th 2.7e-6, worst 1.6e-5
    assert np.percentile(dmx, 99) <= 1e-6 and np.percentile(dmx, 99.9) <= 2e-5
    mxw = np.flatnonzero(fin & (rst == RX_MAXITER))
    assert (dmx <= _maxiter_ceiling("config5_65536", mxw)).all(), float((dmx / _maxiter_ceiling("config5_65536", mxw)).max())
    # the fluxes themselves, at the same width (north_star's bar is stated on flux).  Two tiers, as README states
    # them: walkers that converge -- 1e-4 relative (+ the background floor) on every line; walkers that stop at
    # maxiter = 200 never settle and amplify round-off over their 200 iterations (in the reference their answer
    # even depends on the worker's previous walker, emcee/pyradex/core.py:896): 99.9 % of their fluxes within 1e-4,
    # none beyond (1 + K_SENS x the reference binary's own 1-ulp response) x the tolerance (observed worst: 7.0e-5, inside the
    # floor-augmented tolerance).
    flux, fst, _ = eng.model_flux_batch(cfg["walkers"], return_info=True)
    rflux, rfst, _ = O.model_flux_batch(mol, src, cfg["walkers"], nthreads=NTH)
    assert np.array_equal(fst, rfst)
    ok, d = _flux_ok(flux, rflux, cfg["walkers"], cfg["tbg"], mol)
    # (a line a million times fainter than the walker's brightest one is the difference of two nearly equal terms,
    # B(T_ex) - B(T_bg) with populations at the 1e-13 level: the relative bar gets an absolute floor of 1e-8 of the
    # brightest line of the same walker -- observed: one such line among 655 360, 2e-3 off at 8e-7 Jy km/s)
    ok |= d <= 1e-8 * np.nanmax(np.abs(rflux), axis=1, keepdims=True)
    conv, mx = rfst == RX_OK, rfst == RX_MAXITER
    assert ok[conv].all(), "converged walkers beyond 1e-4 on flux: %d" % int((~ok[conv]).any(axis=1).sum())
    rel = d[mx] / np.maximum(np.abs(rflux[mx]), 1e-300)
    frac_ok = ok[mx].mean()
    relw = np.where(ok[mx], 0.0, rel)                                # (entries inside the tolerance or under the floor count as 0)
    print("maxiter walkers: %d, flux entries within tolerance %.5f; relative deviation of all their flux entries: 99th pct %.2e, "
          "99.9th pct %.2e, max %.2e; worst beyond tolerance %.2e"
          % (int(mx.sum()), frac_ok, np.percentile(rel, 99), np.percentile(rel, 99.9), float(rel.max()), float(relw.max())))
    st0 = O.State(mol)
    st0.backrad(cfg["tbg"])
    W = cfg["walkers"]
    tol = 1e-4 * np.abs(rflux) + 1e-10 * (st0.arr("backi").max() * 10.0 ** W[:, 3] * 1e23)[:, None]      # (_flux_ok's)
    resp_sb = _sensitivity("config5_65536", np.flatnonzero(mx), missing_ok=True)[0]    # (mx: the flux call's status, no prior)
    assert frac_ok >= 0.999 and (ok[mx] | (d[mx] <= tol[mx] * (1.0 + K_SENS * resp_sb)[:, None])).all()


def test_full_width_parity_config4(eng, mol):
    """2048 two-component walkers (config 4's ensemble) against the oracle."""
    cfg = workloads.config4(2048)
    src = _truth_source(eng, mol, cfg)
    lnp, st, nit = eng.lnprob_batch(cfg["walkers"], return_info=True)
    rl, rst, rnit = O.lnprob_batch(mol, src, cfg["walkers"], nthreads=NTH)
    assert np.array_equal(st, rst) and (nit == rnit).mean() >= 0.995
    fin = np.isfinite(rl)
    assert np.array_equal(fin, np.isfinite(lnp)) and fin.sum() > 1000
    dok, dmx = _report("config 4, 2048 two-component walkers", lnp, rl, rst)
    assert dok < 1e-6 and np.percentile(dmx, 99) < 1e-6                                  # (observed: 9e-15, 1.4e-14)
    assert (dmx <= _maxiter_ceiling("config4_2048", np.flatnonzero(fin & (rst == RX_MAXITER)))).all()


def test_device_index_is_validated_not_substituted(co_path, mol):
    """A device-side source index the host cannot see: slots outside the table, never set, or of another
    ncomp give RX_INVALID / -inf for that walker -- slot 0 is not silently used; ncomp is the caller's."""
    import torch
    e = Engine(co_path)
    cfg = workloads.config2(64)
    dev = torch.device("cuda:0")
    P = torch.from_numpy(cfg["walkers"]).to(dev)
    # slot 0 unset, slot 5 set: ncomp comes from the argument (the tensor's width), not from slot 0
    src = _truth_source(e, mol, cfg)
    e.set_source(cfg["tbg"], cfg["Jup"], src.flux, src.eflux, cfg["bounds"], src=5)
    e2 = Engine(co_path)
    e2.set_source(cfg["tbg"], cfg["Jup"], src.flux, src.eflux, cfg["bounds"], src=5)
    c4 = workloads.config4(8)
    e2.set_source(c4["tbg"], c4["Jup"], np.ones(10), np.ones(10), c4["bounds"], 2, 40.0, src=6)
    idx = torch.full((64,), 5, dtype=torch.int32, device=dev)
    idx[3], idx[4], idx[5], idx[6] = 64, -1, 7, 6          # out of range (2x), never set, 2-component source
    lnp, st, nit = e2.lnprob_batch_torch(P, src_index=idx)
    torch.cuda.synchronize()
    ref = e.lnprob_batch(cfg["walkers"])
    lnp, st = lnp.cpu().numpy(), st.cpu().numpy()
    bad = np.array([3, 4, 5, 6])
    assert np.all(st[bad] == RX_INVALID) and np.all(lnp[bad] == -np.inf)
    good = np.setdiff1d(np.arange(64), bad)
    assert np.array_equal(lnp[good], ref[good])
    # without an index the batch addresses slot 0, which e2 never set: an error, not a guess
    from radex_emcee_amd.engine import EngineError
    with pytest.raises(EngineError):
        e2.lnprob_batch_torch(P)
    with pytest.raises(EngineError):
        e.lnprob_batch_torch(torch.zeros(4, 8, dtype=torch.float64, device=dev))   # slot 0 is 1-component
    # engine.lnprob_batch derives the layout from the slots the batch addresses
    P8 = c4["walkers"]
    e2.set_source(c4["tbg"], c4["Jup"], np.ones(10), 0.1 * np.ones(10), c4["bounds"], 2, 40.0, src=6)
    out = e2.lnprob_batch(P8, src_index=np.full(8, 6, dtype=np.int32))
    assert out.shape == (8,)
    with pytest.raises(ValueError):
        e2.lnprob_batch(P8, src_index=np.full(3, 6, dtype=np.int32))
    assert e.time_lnprob_torch(P[:0], lnp=torch.empty(0, dtype=torch.float64, device=dev),
                               status=torch.empty(0, dtype=torch.int32, device=dev),
                               niter=torch.empty(0, dtype=torch.int32, device=dev)) == 0.0
    e.close(); e2.close()


def test_one_handle_on_two_streams_is_ordered(eng, mol):
    """the handle owns its queue and scratch; launches of ONE handle on two
    non-blocking streams are ordered by an event instead of racing."""
    import torch
    dev = torch.device("cuda:0")
    cfgA, cfgB = workloads.config2(2048, seed=21), workloads.config4(1500)
    eng.set_source(cfgB["tbg"], cfgB["Jup"], np.ones(10), 0.1 * np.ones(10), cfgB["bounds"], 2, 40.0, src=1)
    _truth_source(eng, mol, cfgA)
    PA = torch.from_numpy(cfgA["walkers"]).to(dev)
    PB = torch.from_numpy(cfgB["walkers"]).to(dev)
    iB = torch.ones(1500, dtype=torch.int32, device=dev)
    refA = [t.clone() for t in eng.lnprob_batch_torch(PA)]
    refB = [t.clone() for t in eng.lnprob_batch_torch(PB, src_index=iB)]
    torch.cuda.synchronize()
    sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outA = [torch.empty(2048, dtype=t, device=dev) for t in (torch.float64, torch.int32, torch.int32)]
    outB = [torch.empty(1500, dtype=t, device=dev) for t in (torch.float64, torch.int32, torch.int32)]
    for _ in range(4):
        eng.lnprob_batch_torch(PA, *outA, stream=sA.cuda_stream)
        eng.lnprob_batch_torch(PB, *outB, src_index=iB, stream=sB.cuda_stream)
    torch.cuda.synchronize()
    for got, want in zip(outA + outB, refA + refB):
        assert torch.equal(torch.nan_to_num(got.double(), neginf=-1e308), torch.nan_to_num(want.double(), neginf=-1e308))


# ---- the stretch move on the device (f-1) ------------------------------------------------------------
def test_device_sampler_is_the_host_checker_bit_for_bit(eng, mol):
    """DeviceEnsembleSampler (propose / solve / accept kernels, state in HBM) against the host sampler
    replaying the same counter-based stream in numpy and calling rx_lnprob_batch for the proposals:
    identical chain and log-probabilities, reproducible per seed, different for another seed."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    cfg = workloads.config2(128)
    _truth_source(eng, mol, cfg)
    p0 = cfg["truth"] + 1e-2 * np.random.RandomState(5).randn(128, 4)
    d = DeviceEnsembleSampler(128, 4, engine=eng, seed=42)
    assert d.schedule == "dataflow"
    st = d.run_mcmc(p0, 30)
    h = EnsembleSampler(128, 4, eng.lnprob_batch, vectorize=True, seed=42, rng="philox")
    sh = h.run_mcmc(p0, 30)
    # the same chain under the half-step schedule (propose / solve / accept launches)
    dh = DeviceEnsembleSampler(128, 4, engine=eng, seed=42, schedule="halfsteps")
    dh.run_mcmc(p0, 30)
    assert np.array_equal(dh.get_chain(), d.get_chain()) and np.array_equal(dh.get_log_prob(), d.get_log_prob())
    assert np.array_equal(dh.acceptance_fraction, d.acceptance_fraction)
    assert np.array_equal(d.get_chain(), h.get_chain())
    assert np.array_equal(d.get_log_prob(), h.get_log_prob())
    assert np.array_equal(st.coords, sh.coords) and np.array_equal(st.log_prob, sh.log_prob)
    assert np.array_equal(d.acceptance_fraction, h.acceptance_fraction)
    assert 0.1 < d.acceptance_fraction.mean() < 0.9
    d2 = DeviceEnsembleSampler(128, 4, engine=eng, seed=42)
    st10 = d2.run_mcmc(p0, 10)
    d2.run_mcmc(st10, 20)                                   # resuming continues the step counter
    assert np.array_equal(d2.get_chain(), d.get_chain())
    d3 = DeviceEnsembleSampler(128, 4, engine=eng, seed=43)
    d3.run_mcmc(p0, 5)
    assert not np.array_equal(d3.get_chain(), d.get_chain()[:5])


def test_dataflow_schedule_is_the_same_chain_at_every_shape(eng, mol):
    """rx_sampler_run_async_device (one persistent kernel, tasks start when their two input walkers are
    final, positions versioned in a ring) against rx_sampler_run_device (half-steps): identical chains
    -- 1024 walkers over more steps than the ring holds versions, walkers spread over the prior box so
    that slow proposals (maxiter) make the wavefronts run ahead; 16 ensembles with their own sources in
    the two-waves-per-SIMD regime; two components; without chain storage; resumed."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
    cfg = workloads.config2(1024)
    _truth_source(eng, mol, cfg)
    out = {}
    for sched in ("dataflow", "halfsteps"):
        d = DeviceEnsembleSampler(1024, 4, engine=eng, seed=5, schedule=sched)
        st = d.run_mcmc(cfg["walkers"], 30)
        st2 = d.run_mcmc(State(st.coords, st.log_prob), 5, store=False)      # resumed, nothing stored
        out[sched] = (d.get_chain(), d.get_log_prob(), d.acceptance_fraction, st2.coords, st2.log_prob)
    for a, b in zip(out["dataflow"], out["halfsteps"]):
        assert np.array_equal(a, b)
    assert out["dataflow"][0].shape == (30, 1024, 4)
    # 16 sources x 640 walkers: 5120 tasks per half-step -> two wavefronts per SIMD
    c3 = workloads.config3(640)
    _set_config3(eng, c3)
    out = {}
    for sched in ("dataflow", "halfsteps"):
        d = DeviceEnsembleSampler(640, 4, engine=eng, nens=16, ens_src=np.arange(16), seed=6, schedule=sched)
        d.run_mcmc(c3["walkers"], 4)
        out[sched] = (d.get_chain(), d.get_log_prob())
    assert np.array_equal(out["dataflow"][0], out["halfsteps"][0])
    assert np.array_equal(out["dataflow"][1], out["halfsteps"][1])
    # two components
    c4 = workloads.config4(256)
    W = c4["walkers"].copy()
    W[128:] = workloads.draw_prior_2comp(c4["bounds"], 128, 17)
    _truth_source(eng, mol, c4)
    out = {}
    for sched in ("dataflow", "halfsteps"):
        d = DeviceEnsembleSampler(256, 8, engine=eng, seed=8, schedule=sched)
        d.run_mcmc(W, 14)
        out[sched] = (d.get_chain(), d.get_log_prob())
    assert np.array_equal(out["dataflow"][0], out["halfsteps"][0])
    assert np.array_equal(out["dataflow"][1], out["halfsteps"][1])
    _truth_source(eng, mol, workloads.config2(8))


@pytest.mark.parametrize("nw,nens,nsteps", [(8, 1, 40), (100, 1, 25), (30, 5, 15), (1500, 1, 3)])
def test_dataflow_schedule_odd_shapes(eng, mol, nw, nens, nsteps):
    """Ensemble sizes that are not powers of two (the walker permutation cycle-walks), the smallest legal
    ensemble, several small ensembles, more steps than ring slots: dataflow == half-steps == host replay."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    cfg = workloads.config2(8)
    _truth_source(eng, mol, cfg)
    rs = np.random.RandomState(nw)
    p0 = cfg["truth"] + 1e-2 * rs.randn(nens, nw, 4)
    out = {}
    for sched in ("dataflow", "halfsteps"):
        d = DeviceEnsembleSampler(nw, 4, engine=eng, nens=nens, seed=77, schedule=sched,
                                  ens_src=None if nens == 1 else np.zeros(nens, dtype=np.int32))
        d.run_mcmc(p0 if nens > 1 else p0[0], nsteps)
        out[sched] = (d.get_chain(), d.get_log_prob())
    assert np.array_equal(out["dataflow"][0], out["halfsteps"][0])
    assert np.array_equal(out["dataflow"][1], out["halfsteps"][1])
    if nens == 1 and nw <= 100:
        h = EnsembleSampler(nw, 4, eng.lnprob_batch, vectorize=True, seed=77, rng="philox")
        h.run_mcmc(p0[0], nsteps)
        assert np.array_equal(h.get_chain(), out["dataflow"][0])
    d = DeviceEnsembleSampler(nw, 4, engine=eng, nens=nens, seed=77,
                              ens_src=None if nens == 1 else np.zeros(nens, dtype=np.int32))
    st = d.run_mcmc(p0 if nens > 1 else p0[0], 0)          # zero steps: the state comes back unchanged
    assert np.array_equal(st.coords.reshape(-1, 4), p0.reshape(-1, 4)) and d.get_chain().shape[0] == 0


def test_dataflow_sampler_gives_up_instead_of_hanging(co_path, mol):
    """Every wait of the dataflow kernel is bounded: with the timeout set to zero a task whose inputs are
    not final at once raises the abort flag, the grid drains and rx_sampler_wait reports the failure."""
    from radex_emcee_amd.engine import EngineError
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    e = Engine(co_path)
    cfg = workloads.config2(256)
    _truth_source(e, mol, cfg)
    e.set_sampler_timeout_ms(0.0)
    d = DeviceEnsembleSampler(256, 4, engine=e, seed=1)
    d.fallback = False
    with pytest.raises(EngineError, match="waited longer"):
        d.run_mcmc(cfg["walkers"], 20)
    # by default the sampler repeats an abandoned run under the half-step schedule: the same chain
    d = DeviceEnsembleSampler(256, 4, engine=e, seed=1)
    with pytest.warns(UserWarning, match="abandoned"):
        d.run_mcmc(cfg["walkers"], 6)
    ref = DeviceEnsembleSampler(256, 4, engine=e, seed=1, schedule="halfsteps")
    ref.run_mcmc(cfg["walkers"], 6)
    assert np.array_equal(d.get_chain(), ref.get_chain()) and np.array_equal(d.acceptance_fraction, ref.acceptance_fraction)
    e.set_sampler_timeout_ms(2000.0)
    d = DeviceEnsembleSampler(256, 4, engine=e, seed=1)
    st = d.run_mcmc(cfg["walkers"], 3)                     # the handle is usable afterwards
    assert np.all(np.isfinite(st.coords)) and d.get_chain().shape == (3, 256, 4)
    e.close()


def test_device_sampler_samples_the_posterior(eng, mol):
    """Statistical validity on the real likelihood: the chain of the device sampler has the moments of
    the chain the host sampler draws with numpy's generator (emcee's own) -- same target."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    cfg = workloads.config2(64)
    _truth_source(eng, mol, cfg)
    p0 = cfg["truth"] + 1e-3 * np.random.RandomState(1).randn(64, 4)
    d = DeviceEnsembleSampler(64, 4, engine=eng, seed=7)
    d.run_mcmc(p0, 700)
    h = EnsembleSampler(64, 4, eng.lnprob_batch, vectorize=True, seed=3)
    h.run_mcmc(p0, 700)
    a, b = d.get_chain(flat=True, discard=300), h.get_chain(flat=True, discard=300)
    # pressure n*T is the well-constrained combination; size and column are degenerate: compare medians
    # in units of the other chain's spread
    for f in (lambda c: c[:, 0] + c[:, 1], lambda c: c[:, 2] + c[:, 3], lambda c: c[:, 1]):
        sa, sb = f(a), f(b)
        assert abs(np.median(sa) - np.median(sb)) < 0.5 * max(sa.std(), sb.std()) + 0.02
        assert 0.5 < sa.std() / sb.std() < 2.0
    assert abs(d.acceptance_fraction.mean() - h.acceptance_fraction.mean()) < 0.1


def test_device_sampler_several_sources_and_two_components(eng, mol):
    """Config 3's shape: ensembles of different sources advance in the same launches and each is the
    chain it is alone; config 4's shape: 8 parameters."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    cfg = workloads.config3(32, init="ball")
    _set_config3(eng, cfg)
    d = DeviceEnsembleSampler(32, 4, engine=eng, nens=16, ens_src=np.arange(16), seed=9)
    st = d.run_mcmc(cfg["walkers"], 12)
    ch = d.get_chain()
    assert ch.shape == (12, 16, 32, 4) and st.coords.shape == (16, 32, 4)
    for k in (0, 7, 15):
        def fn(P, k=k):
            return eng.lnprob_batch(P, src_index=np.full(len(P), k, dtype=np.int32))
        h = EnsembleSampler(32, 4, fn, vectorize=True, seed=9, rng="philox")
        # ensemble k of a multi-ensemble run draws the stream of ensemble index k: replay it
        from radex_emcee_amd import sampler as S
        coords = cfg["walkers"][k].copy()
        lnp = fn(coords)
        for step in range(12):
            for split in range(2):
                full = np.zeros((16, 32, 4)); full[k] = coords
                q, f, w = S.stretch_propose(full, 16, 32, 2.0, 9, step, split)
                sl = slice(16 * k, 16 * (k + 1))
                wl = w[sl] - 32 * k
                nl = fn(q[sl])
                r = S.philox4x32_10(np.arange(16), k, step, S.PURPOSE_ACCEPT + 16 * split, 9, 0)
                with np.errstate(divide="ignore", invalid="ignore"):
                    acc = np.log(S.u53(r[0], r[1])) < (f[sl] + nl) - lnp[wl]
                coords[wl[acc]] = q[sl][acc]
                lnp[wl[acc]] = nl[acc]
            assert np.array_equal(ch[step, k], coords), (k, step)
    # two components
    c4 = workloads.config4(64)
    _truth_source(eng, mol, c4)
    d = DeviceEnsembleSampler(64, 8, engine=eng, seed=2)
    d.run_mcmc(c4["walkers"], 6)
    h = EnsembleSampler(64, 8, eng.lnprob_batch, vectorize=True, seed=2, rng="philox")
    h.run_mcmc(c4["walkers"], 6)
    assert np.array_equal(d.get_chain(), h.get_chain())
    _truth_source(eng, mol, workloads.config2(8))


def test_reference_kats_with_a_real_co_dat_on_the_gpu():
    """/root/reference/emcee/pyradex/tests/test_radex.py:99-115 (Tex, tau, populations of CO 1-0) and
    :175-200 (the chain of Tex values as parameters change) through radex.Radex -> rx_solve_batch.
    They need the LAMDA co.dat the reference does not ship: activated by RADEX_DATAPATH."""
    dp = os.getenv("RADEX_DATAPATH")
    if not dp or not os.path.exists(os.path.join(dp, "co.dat")):
        pytest.skip("real LAMDA co.dat not available (set RADEX_DATAPATH)")
    from radex_emcee_amd.radex import Radex
    rdx = Radex(species='co', collider_densities={'H2': 1e4}, column_per_bin=1e14, deltav=1.0,
                temperature=30, tbackground=2.73, datapath=dp)
    rdx.run_radex()
    np.testing.assert_approx_equal(rdx.tex[0], 56.131, 5)
    np.testing.assert_approx_equal(rdx.tau[0], 1.786E-03, 4)
    np.testing.assert_approx_equal(rdx.upperlevelpop[0], 3.640E-01, 4)
    np.testing.assert_approx_equal(rdx.lowerlevelpop[0], 1.339E-01, 4)
    RR = Radex(datapath=dp, species='co', column=1e15, density=1e3, temperature=20)
    RR.run_radex()
    np.testing.assert_almost_equal(RR.tex[0], 8.69274406690759, decimal=2)
    RR.column = 1e14
    RR.run_radex()
    np.testing.assert_almost_equal(RR.tex[0], 8.0986662583317646, decimal=2)
    RR.density = 1e4
    RR.run_radex()
    np.testing.assert_almost_equal(RR.tex[0], 25.381267019506591, decimal=1)
    RR.temperature = 25
    RR.run_radex()
    np.testing.assert_almost_equal(RR.tex[0], 37.88, decimal=1)


def test_model_flux_on_device_buffers(eng):
    """rx_model_flux_batch_device (device pointers, a caller's stream) gives what rx_model_flux_batch gives through host buffers."""
    import torch
    cfg = workloads.config2(700, seed=3)
    eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    f0, s0, n0 = eng.model_flux_batch(cfg["walkers"], return_info=True)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        P = torch.from_numpy(cfg["walkers"]).cuda()
        f1, s1, n1 = eng.model_flux_batch_torch(P, stream=st.cuda_stream)
    st.synchronize()
    assert np.array_equal(f0, f1.cpu().numpy(), equal_nan=True) and np.array_equal(s0, s1.cpu().numpy()) and np.array_equal(n0, n1.cpu().numpy())


def test_ortho_fraction_of_h2(co_path):
    """rx_set_fortho: the share of n_H2 that goes to oH2 (default 0.75 = opr / (1 + opr), emcee_radex.py:95-96, 124-126).  The
    fluxes of model_lvg with another share must be those of the solve with the densities split by hand."""
    e = Engine(co_path)
    cfg = workloads.config2(512, seed=77)
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    W = cfg["walkers"]
    n = 10.0 ** W[:, 0]
    for f in (0.75, 0.6, 0.0):
        e.set_fortho(f)
        flux, st, nit = e.model_flux_batch(W, return_info=True)
        r = e.solve_batch(10.0 ** W[:, 1], 10.0 ** W[:, 2], np.stack([(1.0 - f) * n, f * n], 1))
        want = r["sb"][:, np.asarray(cfg["Jup"]) - 1] * (10.0 ** W[:, 3:4]) * 1e23
        assert np.array_equal(st, r["status"])
        ok = (st == RX_OK)
        assert ok.sum() > 400 and (nit[ok] == r["niter"][ok]).mean() > 0.99
        # (the split is formed on the device as n (1 - f), n f: the last bit of a density may differ from the host's product, and a
        # line much fainter than the walker's brightest is the difference of two nearly equal terms)
        tol = 1e-6 * np.abs(want) + 1e-9 * np.nanmax(np.abs(want), axis=1, keepdims=True)
        assert (np.abs(flux - want)[ok] <= tol[ok]).all(), (f, float(np.nanmax((np.abs(flux - want) / tol)[ok])))
    e.set_fortho(0.75)
    e.close()


def test_iteration_limits_other_than_the_references(co_path, mol):
    """rx_set_iteration_limits (Radex.run_radex's miniter / maxiter, core.py:460-463, 903-920).  A handle in its default state runs
    the instantiation of the solve kernel that has 10 / 200 as constants; any other limits select the general one (rx_kernel.hip.inc:
    GEN).  Explicit 10 / 200 must be the default bit for bit; with maxiter = 60 every walker that needs fewer iterations keeps its
    bits, every other stops at 60 with RX_MAXITER; and both limits are held to the checker's run_radex on single walkers."""
    e = Engine(co_path)
    e.set_source(2.73)
    rng = np.random.default_rng(5)
    N = 2000
    tkin = 10.0 ** rng.uniform(0.7, 2.9, N)
    cd = 10.0 ** rng.uniform(13.0, 18.8, N)
    n = 10.0 ** rng.uniform(1.8, 6.2, N)
    dens = np.stack([0.25 * n, 0.75 * n], 1)
    base = e.solve_batch(tkin, cd, dens)
    e.set_iteration_limits(10, 200)
    same = e.solve_batch(tkin, cd, dens)
    for k in ("status", "niter", "xpop", "tex", "tau"):
        assert np.array_equal(base[k], same[k], equal_nan=True), k
    e.set_iteration_limits(10, 60)
    cut = e.solve_batch(tkin, cd, dens)
    short = base["niter"] < 60
    assert short.sum() > N // 2 and (~short).sum() > 20
    for k in ("status", "niter", "xpop", "tex", "tau"):
        assert np.array_equal(base[k][short], cut[k][short], equal_nan=True), k
    assert (cut["niter"][~short] == 60).all()
    stopped = ~short & (base["niter"] > 60)
    assert (cut["status"][stopped] == RX_MAXITER).all()
    e.set_iteration_limits(25, 200)                      # (miniter only gates the python-side stop rule, core.py:911-920: matrix_'s own
    late = e.solve_batch(tkin, cd, dens)                 #  convergence test ends the loop from iteration 10 on whatever it is)
    assert (late["niter"] >= base["niter"]).all()
    # the checker, walker by walker, with the same limits
    for lim, got in (((10, 60), cut), ((25, 200), late)):
        for w in list(np.flatnonzero(~short)[:6]) + list(np.flatnonzero(short)[:6]):
            st = O.State(mol)
            st.backrad(2.73)
            st.set_density({2: dens[w, 0], 3: dens[w, 1]})
            st.s.tkin = tkin[w]; st.s.cdmol = cd[w]
            assert st.rates() == 0
            it, conv = st.run(False, lim[0], lim[1])
            assert int(got["niter"][w]) == it, (lim, w, got["niter"][w], it)          # (cut walkers: exactly the new limit, both sides)
            x = st.arr("xpop").copy()
            # per level 1e-6 relative + 1e-14 absolute (what a double-precision solve resolves: tests/test_gpu_refine.py); a walker
            # cut at the new maxiter is mid-flight, not chaotic yet -- held to 1e-4 relative + the same floor
            # (a walker that runs into the REFERENCE's maxiter = 200 is one of the chaotic ones: the maxiter tier of _report)
            if it >= 200:
                continue
            tol = (1e-4 if it >= lim[1] else 1e-6) * np.abs(x) + 1e-14
            assert np.all(np.abs(got["xpop"][w] - x) <= tol), (lim, w, it, np.max(np.abs(got["xpop"][w] - x) / tol))
    e.set_iteration_limits(10, 200)
    e.close()


def test_rate_setup_forms_agree_bit_for_bit(co_path, toy_path):
    """The collisional half of a walker's set-up exists in two forms: with one wavefront per SIMD lane i evaluates the detailed
    balance of all nlev partners of level i, with two every unordered pair of levels is evaluated once (rx_kernel.hip.inc:
    PAIRS_ONCE).  Same expression per pair, ctot summed in the same order: the populations, T_ex, tau and iteration counts must
    be the same bits.  Only the CO ladder instantiation has the second form (the general two-wavefront kernels keep the first:
    see the comment at PAIRS_ONCE); the general 41-level instantiation (sphere geometry) and the toy molecule (6 levels in the
    8-level instantiation) are held to the same bar between the two builds."""
    rng = np.random.default_rng(77)
    for path, geom, npart in ((co_path, "lvg", 2), (co_path, "sphere", 2), (toy_path, "lvg", 1), (toy_path, "sphere", 1)):
        e = Engine(path, escapeProbGeom=geom)
        assert e.npart == npart
        e.set_source(2.73)
        N = 3000
        tkin = 10.0 ** rng.uniform(0.6, 2.9, N)
        cd = 10.0 ** rng.uniform(12.0, 18.5, N)
        dens = 10.0 ** rng.uniform(1.5, 6.5, (N, npart))
        dens[::7, 0] = 0.0 if npart > 1 else dens[::7, 0]              # (a partner without density is skipped)
        res = []
        for occ in (1, 2):
            e.set_waves_per_simd(occ)
            res.append(e.solve_batch(tkin, cd, dens))
        e.set_waves_per_simd(0)
        e.close()
        assert (res[0]["status"] == RX_OK).sum() > N // 2, (path, geom)
        for k in ("status", "niter", "xpop", "tex", "tau", "sb"):
            assert np.array_equal(res[0][k], res[1][k], equal_nan=True), (os.path.basename(path), geom, k)


def test_one_and_two_wavefronts_per_simd_agree_bit_for_bit(eng, mol):
    """The "same chain bit for bit across schedules and ranks" claims need the two builds of the solve to give the SAME bits:
    a rank's block is nq / nranks tasks, so a rank can run the one-wavefront-per-SIMD build (its exp / log constants held
    in registers: exp_neg, rx_log_t<HELD>) where the one-GPU run of the same ensemble uses the two-wavefront build (OCML
    exp).  Prior-box walkers including ones that run into maxiter and both escape-probability branches; then a short
    dataflow chain under each."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    cfg = workloads.config2(6144, seed=4242)
    _truth_source(eng, mol, cfg)
    res = []
    for occ in (1, 2):
        eng.set_waves_per_simd(occ)
        lnp, st, nit = eng.lnprob_batch(cfg["walkers"], return_info=True)
        flux = eng.model_flux_batch(cfg["walkers"][:1024])
        d = DeviceEnsembleSampler(1024, 4, engine=eng, seed=31)
        stt = d.run_mcmc(cfg["walkers"][:1024], 6)
        res.append((lnp, st, nit, flux, stt.coords, stt.log_prob, d.get_chain()))
    eng.set_waves_per_simd(0)
    assert (res[0][1] == RX_MAXITER).sum() >= 20 and (res[0][1] == RX_OK).sum() > 4000
    assert (res[0][2] > 40).any() and (res[0][2] < 20).any()
    for a, b, what in zip(res[0], res[1], ("lnprob", "status", "niter", "flux", "chain end", "chain lnp", "chain")):
        assert np.array_equal(a, b, equal_nan=True), what
