"""The dataflow sampler's head start (rx_set_sampler_speculation): a task whose partner's last update is pending
starts on the partner's previous position (hypothesis 0), then on the partner's published proposal (hypothesis
1), watches the partner while it solves, and keeps a result only if the partner's real position is bit for bit
the one it assumed.  The chain must be THE chain -- the half-step schedule's, emcee_radex.py:483-499 -- in every
mode, at every shape, and the counters must say that the head starts really happened."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O                      # noqa: E402  (checker only)
from radex_emcee_amd import workloads               # noqa: E402
from radex_emcee_amd.engine import Engine           # noqa: E402
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State   # noqa: E402
from test_gpu_parity import _truth_source           # noqa: E402
from test_gpu_round2 import _set_config3            # noqa: E402


@pytest.fixture(scope="module")
def eng(co_path):
    return Engine(co_path)


@pytest.fixture(scope="module")
def mol(co_path):
    return O.Molecule(co_path)


def _chain(eng, mode, nw, ndim, p0, nsteps, seed, sched="dataflow", **kw):
    eng.set_sampler_speculation(mode)
    eng.sampler_stats(True)
    d = DeviceEnsembleSampler(nw, ndim, engine=eng, seed=seed, schedule=sched, **kw)
    st = d.run_mcmc(p0, nsteps)
    st2 = d.run_mcmc(State(st.coords, st.log_prob), 3, store=False)
    stats = eng.sampler_stats(False)
    eng.set_sampler_speculation(-1)
    return (d.get_chain(), d.get_log_prob(), d.acceptance_fraction, st2.coords, st2.log_prob), stats


def test_prior_box_ensemble_same_chain_in_every_mode(eng, mol):
    """1024 prior-box walkers (proposals that run into maxiter, 12 % outside the prior): off, on, automatic and
    the half-step schedule give one chain; on: about half the tasks take the head start."""
    cfg = workloads.config2(1024)
    _truth_source(eng, mol, cfg)
    ref, _ = _chain(eng, 0, 1024, 4, cfg["walkers"], 24, 11, sched="halfsteps")
    off, s_off = _chain(eng, 0, 1024, 4, cfg["walkers"], 24, 11)
    on, s_on = _chain(eng, 1, 1024, 4, cfg["walkers"], 24, 11)
    auto, s_auto = _chain(eng, -1, 1024, 4, cfg["walkers"], 24, 11)
    for got in (off, on, auto):
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
    assert s_off["head_starts"] == 0 and s_off["evaluated_twice"] == 0
    assert s_on["tasks"] == 1024 * 27 and s_on["head_starts"] > 0.2 * s_on["tasks"]
    assert 0 < s_on["evaluated_twice"] < s_on["head_starts"]
    assert s_auto["head_starts"] > 0                                  # one wavefront per SIMD: on by default
    # the counters describe the evaluation that stands: the same proposals reach the solver in every mode
    assert s_on["solved"] == s_off["solved"] and s_on["niter_sum"] == s_off["niter_sum"]
    assert s_on["maxiter_solves"] == s_off["maxiter_solves"]


@pytest.mark.parametrize("nw,nens,nsteps", [(8, 1, 30), (100, 1, 20), (30, 5, 12)])
def test_small_and_odd_ensembles(eng, mol, nw, nens, nsteps):
    """Few walkers: almost every task depends on a task in flight (the smallest ensemble: every one)."""
    cfg = workloads.config2(8)
    _truth_source(eng, mol, cfg)
    p0 = cfg["truth"] + 1e-2 * np.random.RandomState(nw).randn(nens, nw, 4)
    kw = dict(nens=nens, ens_src=None if nens == 1 else np.zeros(nens, dtype=np.int32))
    ref, _ = _chain(eng, 0, nw, 4, p0 if nens > 1 else p0[0], nsteps, 5, sched="halfsteps", **kw)
    on, s = _chain(eng, 1, nw, 4, p0 if nens > 1 else p0[0], nsteps, 5, **kw)
    for a, b in zip(on, ref):
        assert np.array_equal(a, b)
    assert s["tasks"] == nens * nw * (nsteps + 3)


def test_two_components_and_two_waves_per_simd(eng, mol):
    """Two components per task (a pass given up between the components); 16 ensembles in the two-waves-per-SIMD
    build, which has no head start whatever the mode says."""
    c4 = workloads.config4(256)
    W = c4["walkers"].copy()
    W[128:] = workloads.draw_prior_2comp(c4["bounds"], 128, 17)
    _truth_source(eng, mol, c4)
    ref, _ = _chain(eng, 0, 256, 8, W, 12, 8, sched="halfsteps")
    on, s = _chain(eng, 1, 256, 8, W, 12, 8)
    for a, b in zip(on, ref):
        assert np.array_equal(a, b)
    assert s["head_starts"] > 0
    c3 = workloads.config3(640)
    _set_config3(eng, c3)
    kw = dict(nens=16, ens_src=np.arange(16))
    ref, _ = _chain(eng, 0, 640, 4, c3["walkers"], 4, 6, sched="halfsteps", **kw)
    auto, s_auto = _chain(eng, -1, 640, 4, c3["walkers"], 4, 6, **kw)
    on, s_on = _chain(eng, 1, 640, 4, c3["walkers"], 4, 6, **kw)
    for got in (auto, on):
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
    assert s_auto["head_starts"] == 0 and s_on["head_starts"] == 0
    _truth_source(eng, mol, workloads.config2(8))


def test_argument_errors_of_the_new_entry_points(eng):
    """rx_set_sampler_speculation / rx_sampler_spec_stats / rx_lnprior_batch: bad arguments are errors, not crashes."""
    import ctypes as C
    from radex_emcee_amd.engine import EngineError
    L, h = eng._L, eng._h
    assert L.rx_set_sampler_speculation(h, 2) != 0 and L.rx_set_sampler_speculation(h, -2) != 0
    assert L.rx_set_sampler_speculation(h, -1) == 0
    assert L.rx_sampler_spec_stats(h, None) != 0
    out = (C.c_double * 1)()
    p = (C.c_double * 4)(4.0, 1.8, 17.0, -9.0)
    assert L.rx_lnprior_batch(h, 0, 0, p, out) == 0                       # no walkers: nothing to do
    assert L.rx_lnprior_batch(h, 61, 1, p, out) != 0                      # a slot that was never set
    assert L.rx_lnprior_batch(h, 0, 1, None, out) != 0
    with pytest.raises(EngineError):
        eng.lnprior_batch(np.zeros((1, 4)), src=61)
