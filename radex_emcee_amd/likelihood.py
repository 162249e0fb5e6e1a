"""Host-side mirror of the reference drivers' posterior functions, backed by the HIP engine.

Per-walker API with the reference's names and argument meaning
  [/root/reference/emcee/emcee_radex.py:98-181, emcee/emcee_radex_2comp.py:106-244]:
    init_radex(tbg), model_lvg(Jup, params, R), lnlike(p, Jup, flux, eflux, R),
    lnprior(p, bounds[, T_d]), lnprob(p, Jup, flux, eflux, bounds=[, T_d=])
and the batched form the sampler uses (one kernel launch per half-step):
    Posterior(...).lnprob_batch(P[N, ndim]) -> lnp[N]

All numerics (prior, RADEX solve, chi^2) run in libradex_emcee_amd.so; these functions only
marshal arguments.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine

opr = 3
fortho = opr / (1 + opr)                 # emcee_radex.py:95-96

R = None                                  # per-process engine handle, like the reference's global

# Source slots of the process-wide engine: slot 0 belongs to the LIVE sampler (EnsembleSampler below keeps
# reading it for its whole life: burn-in, reset, production); the per-walker functions with the reference's
# signatures are pure, like the reference's, and evaluate in scratch slots of their own.
_SLOT_SAMPLER, _SLOT_LNPROB, _SLOT_SCRATCH = 0, 62, 63        # (RX_MAX_SOURCES = 64)


def init_radex(tbg=2.7315, molfile=None, device=0):
    """emcee_radex.py:104-117: one handle per process/GPU."""
    global R
    if R is None:
        R = Engine(molfile=molfile, species='co', escapeProbGeom='lvg', deltav=1.0, device=device)
    R.set_source(tbg)
    R._tbg = float(tbg)
    return R


class Posterior:
    """Everything `lnprob` closes over for one source, resident on one GPU."""

    def __init__(self, Jup, flux, eflux, bounds, tbg, ncomp=1, T_d=None, engine=None, src=0,
                 molfile=None, device=0):
        self.engine = engine or Engine(molfile=molfile, device=device)
        self.ncomp, self.ndim, self.src = int(ncomp), 4 * int(ncomp), int(src)
        self.Jup = np.asarray(np.int_(Jup))
        self.flux = np.asarray(flux, dtype=np.float64)
        self.eflux = np.asarray(eflux, dtype=np.float64)
        self.bounds = np.asarray(bounds, dtype=np.float64).reshape(self.ndim, 2)
        self.T_d = T_d
        self.tbg = float(tbg)
        self.engine.set_source(self.tbg, self.Jup, self.flux, self.eflux, self.bounds, self.ncomp,
                               self.T_d, src=self.src)

    def model_lvg(self, params):
        """[N, ndim] -> [N, nJ] Jy km/s (NaN where the reference raises ValueError)."""
        P = np.atleast_2d(np.asarray(params, dtype=np.float64))
        return self.engine.model_flux_batch(P, src=self.src)

    def lnprob_batch(self, params, return_info=False):
        P = np.atleast_2d(np.asarray(params, dtype=np.float64))
        if self.src == 0:
            return self.engine.lnprob_batch(P, return_info=return_info)
        idx = np.full(len(P), self.src, dtype=np.int32)
        return self.engine.lnprob_batch(P, src_index=idx, return_info=return_info)

    __call__ = lnprob_batch

    def lnprob(self, p):
        return float(self.lnprob_batch(np.asarray(p)[None, :])[0])


# ---- per-walker functions with the reference's signatures ---------------------------------
def model_lvg(Jup, params, R=None):
    """emcee_radex.py:120-130 (4 params) / emcee_radex_2comp.py:122-147 (8 params).
    Raises ValueError where the reference's setters do."""
    R = R or globals()["R"]
    p = np.asarray(params, dtype=np.float64)
    ncomp = p.size // 4
    Jup = np.asarray(np.int_(Jup))
    wide = np.tile(np.array([-np.inf, np.inf]), (4 * ncomp, 1))
    R.set_source(R._tbg, Jup, np.zeros(len(Jup)), np.ones(len(Jup)), wide, ncomp, None, src=_SLOT_SCRATCH)
    flux, status, _ = R.model_flux_batch(p[None, :], src=_SLOT_SCRATCH, return_info=True)
    if status[0] == 2:
        raise ValueError("parameters outside RADEX's valid range (temperature/column/colliders)")
    return flux[0]


def lnprior(p, bounds, T_d=None, R=None):
    """emcee_radex.py:169-175; emcee_radex_2comp.py:199-234.  Evaluated by the engine's own prior
    (rx_lnprior_batch: the device function the fused lnprob uses; no solve is run): no second
    statement of the branches on the host.  Works before init_radex() -- the handle is then created lazily with
    the default background --; each call is a small GPU round trip, so batch bound checks of many points through
    Engine.lnprior_batch rather than looping over this function."""
    R = R or globals()["R"]
    if R is None:
        # the reference's lnprior needs no Radex handle (callers bound-check start points before they create one):
        # the engine is created on first use -- on the GPU, like everything else; without one this raises EngineError
        R = init_radex()
    p = np.asarray(p, dtype=np.float64)
    ncomp = p.size // 4
    one = np.ones(1)
    R.set_source(getattr(R, "_tbg", 2.7315), np.array([1]), one, one, bounds, ncomp, T_d, src=_SLOT_SCRATCH)
    return float(R.lnprior_batch(p[None, :], src=_SLOT_SCRATCH)[0])


def lnlike(p, Jup, flux, eflux, R=None, sigma_floor=1e-12):
    """emcee_radex.py:132-167 / emcee_radex_2comp.py:169-196: Gaussian log-likelihood with the
    reference's guards (ValueError of the setters, non-finite data / model / residuals -> -inf,
    |eflux| floored).  Evaluated by the engine's fused likelihood epilogue with the prior switched off
    (rx_set_source_prior): the same code path the sampler uses, no second implementation on the host."""
    R = R or globals()["R"]
    p = np.asarray(p, dtype=np.float64)
    ncomp = p.size // 4
    Jup = np.asarray(np.int_(Jup))
    e = np.asarray(eflux, dtype=np.float64)
    if sigma_floor > 1e-12:                       # (the engine's own floor is the reference's default 1e-12)
        e = np.maximum(np.abs(e), sigma_floor)
    wide = np.tile(np.array([-np.inf, np.inf]), (4 * ncomp, 1))
    R.set_source(R._tbg, Jup, np.asarray(flux, dtype=np.float64), e, wide, ncomp, None, src=_SLOT_SCRATCH)
    R.set_source_prior(_SLOT_SCRATCH, False)
    return float(R.lnprob_batch(p[None, :], src_index=np.array([_SLOT_SCRATCH], dtype=np.int32))[0])


def lnprob(p, Jup, flux, eflux, bounds=None, T_d=None):
    """emcee_radex.py:177-181 / emcee_radex_2comp.py:237-244: one walker through the engine."""
    eng = globals()["R"]
    if eng is None:
        raise RuntimeError("call init_radex(tbg) first")
    p = np.asarray(p, dtype=np.float64)
    ncomp = p.size // 4
    eng.set_source(eng._tbg, np.asarray(np.int_(Jup)), flux, eflux, bounds, ncomp, T_d, src=_SLOT_LNPROB)
    return float(eng.lnprob_batch(p[None, :], src_index=np.array([_SLOT_LNPROB], dtype=np.int32))[0])


def EnsembleSampler(nwalkers, ndim, log_prob_fn, args=None, kwargs=None, pool=None, seed=0, **options):
    """Call-site compatible with the reference's
        sampler = emcee.EnsembleSampler(nwalkers, ndim, lnprob, args=(Jup, flux, eflux),
                                        kwargs={'bounds': bounds[, 'T_d': T_d]}, pool=pool)
    (emcee_radex.py:483-488, emcee_radex_2comp.py:557-563).  When `log_prob_fn` is this module's
    `lnprob`, its closure arguments become source slot 0 of the process's engine (`init_radex`) and the
    chain runs on the GPU (sampler.DeviceEnsembleSampler: run_mcmc / reset / get_chain / get_log_prob as
    the scripts use them); `pool` is accepted and ignored -- the batch IS the parallelism.  Any other
    function gets the host sampler (one call per walker, like emcee without vectorize)."""
    from . import sampler as _s
    if log_prob_fn is lnprob:
        eng = globals()["R"]
        if eng is None:
            raise RuntimeError("call init_radex(tbg) first")
        a = tuple(args or ())
        kw = dict(kwargs or {})
        if len(a) != 3:
            raise ValueError("lnprob takes args=(Jup, flux, eflux)")
        Jup, flux, eflux = a
        ncomp = int(ndim) // 4
        eng.set_source(eng._tbg, np.asarray(np.int_(Jup)), np.asarray(flux, dtype=np.float64),
                       np.asarray(eflux, dtype=np.float64), kw.get("bounds"), ncomp, kw.get("T_d"), src=_SLOT_SAMPLER)
        return _s.DeviceEnsembleSampler(nwalkers, ndim, engine=eng, seed=seed, **options)
    return _s.EnsembleSampler(nwalkers, ndim, log_prob_fn, args=args, kwargs=kwargs, pool=pool, seed=seed, **options)
