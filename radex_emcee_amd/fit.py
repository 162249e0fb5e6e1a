"""Per-source fitting driver: the callers around the hot path.

Mirrors the parts of `main()` in /root/reference/emcee/emcee_radex.py:382-531 and
emcee/emcee_radex_2comp.py:480-611 that surround the likelihood:
  per-source set-up (tbg, bounds, p0) -> optional warm start (curve_fit, minimize; f-4)
  -> burn-in + production chains with the GPU-batched sampler (f-1)
  -> result tuple/pickle with the reference's layout (f-3) -> 16/50/84 percentile summary.
Plotting (`replot`, corner) is out of scope.
"""
from __future__ import annotations

import pickle

import numpy as np

from . import data_io, workloads
from .likelihood import Posterior
from .sampler import EnsembleSampler

P0_1COMP = [4.0, 1.4, 17.8, -9.85]                                   # emcee_radex.py:444-447
P0_2COMP = [1.9, 1.2, 16.4, -12.1, 3.9, 2.5, 17.5, -12.1]            # emcee_radex_2comp.py:513-522


def warm_start(post: Posterior, p0, use_curve_fit=True, use_minimize=True):
    """curve_fit (bounded TRF) then minimize(-lnprob) (L-BFGS-B) as in emcee_radex.py:453-468.
    Returns (popt, pcov, pmin).  Serial N=1 calls through the same engine."""
    from scipy.optimize import curve_fit, minimize
    bounds = post.bounds
    p0 = np.clip(np.asarray(p0, dtype=float), bounds[:, 0], bounds[:, 1])
    popt, pcov = p0, None
    if use_curve_fit:
        def opt_fun(_J, *p):
            m = post.model_lvg(np.asarray(p)[None, :])[0]
            return np.where(np.isfinite(m), m, 1e300)
        try:
            popt, pcov = curve_fit(opt_fun, post.Jup, post.flux, sigma=post.eflux, p0=p0,
                                   bounds=list(zip(*bounds)))
        except (RuntimeError, ValueError):
            popt, pcov = p0, None                                   # "curve_fit : failed" -> p0
    pmin = popt
    if use_minimize:
        def nll(p):
            v = -post.lnprob(p)
            return v if np.isfinite(v) else 1e300
        res = minimize(nll, popt, bounds=bounds)
        pmin = res.x
    return np.asarray(popt), pcov, np.asarray(pmin)


def summarize(flatchain, ncomp=1):
    """16/50/84 percentiles of (log n, log T, log N, log P = log n + log T) per component as
    (median, +err, -err) triples (emcee_radex.py:511-531)."""
    out = []
    for c in range(ncomp):
        ch = flatchain[:, 4 * c:4 * c + 4]
        plot = np.hstack((ch[:, [0, 1, 2]], ch[:, [0]] + ch[:, [1]]))
        pc = np.percentile(plot, [16, 50, 84], axis=0)
        out.append({name: (pc[1, i], pc[2, i] - pc[1, i], pc[1, i] - pc[0, i])
                    for i, name in enumerate(("n_H2", "T_kin", "N_CO", "P"))})
    return out


def result_tuple(source, z, bounds, Jup, flux, eflux, popt, pcov, pmin, theta_med, chain, lnprobability,
                 T_d=None):
    """(source, z, bounds[, T_d], (Jup, flux, eflux), (popt, pcov), pmin, theta_med, (chain, lnprob))
    -- emcee_radex.py:504-509; the 2-component script inserts T_d after bounds (:580-585)."""
    head = (source, z, bounds) + ((T_d,) if T_d is not None else ())
    flux, eflux = _jykms(flux), _jykms(eflux)
    return head + ((Jup, flux, eflux), (popt, pcov), pmin, theta_med, (chain, lnprobability))


def _jykms(v):
    """The reference pickles flux / eflux as astropy Quantities in Jy km/s (get_source,
    emcee_radex.py:229-240) and its replot reads `flux.value` (:315): wrap them when astropy can be
    imported, so that a pickle written here opens there; plain arrays otherwise (astropy is not a
    dependency of this package)."""
    if hasattr(v, "unit"):
        return v
    try:
        import astropy.units as u
    except Exception:
        return np.asarray(v)
    return u.Quantity(np.asarray(v, dtype=float), u.Jy * u.km / u.s)


def save_result(path, tup):
    with open(path, "wb") as f:
        pickle.dump(tup, f)


def load_result(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def fit_source(source, data, ncomp=1, nwalkers=None, n_iter_burn=100, n_iter_walk=None, seed=None,
               engine=None, warm=True, lnprob_wrapper=None, sampler=None):
    """One source end to end.  Defaults are the reference's: 100 walkers x (100 + 500) steps for
    one component, 400 x (100 + 1000) for two (emcee_radex.py:472-474, 2comp:548-550).
    sampler="device" (default): the chain runs in HBM (DeviceEnsembleSampler, dataflow schedule,
    counter-based random stream); "host" (default when an lnprob_wrapper is given): numpy stretch move
    with numpy's generator + one rx_lnprob_batch per half-step."""
    if sampler is None:
        sampler = "host" if lnprob_wrapper is not None else "device"
    if sampler not in ("device", "host"):
        raise ValueError("sampler must be 'device' or 'host'")
    if ncomp == 1:
        z, _lw, Jup, flux, eflux = data_io.get_source(source, data)
        T_d, p0 = None, P0_1COMP
        nwalkers = nwalkers or 100
        n_iter_walk = n_iter_walk or 500
    else:
        z, T_d, _lw, Jup, flux, eflux = data_io.get_source(source, data)
        p0 = P0_2COMP
        nwalkers = nwalkers or 400
        n_iter_walk = n_iter_walk or 1000
    tbg, bounds = data_io.source_setup(z, ncomp)
    post = Posterior(Jup, flux, eflux, bounds, tbg, ncomp=ncomp, T_d=T_d, engine=engine)
    if warm:
        popt, pcov, pmin = warm_start(post, p0)
    else:
        popt = np.clip(np.asarray(p0, dtype=float), bounds[:, 0], bounds[:, 1])
        pcov, pmin = None, popt
    ndim = len(popt)
    rng = np.random.RandomState(seed)
    pos = popt + 1e-3 * rng.randn(nwalkers, ndim)                    # emcee_radex.py:477
    if sampler == "device":
        from .sampler import DeviceEnsembleSampler
        dsm = DeviceEnsembleSampler(nwalkers, ndim, engine=post.engine, seed=0 if seed is None else int(seed),
                                    ens_src=[post.src] if post.src else None)
        state = dsm.run_mcmc(pos, n_iter_burn, store=False)
        dsm.reset()
        dsm.run_mcmc(state, n_iter_walk)
        chain, lnprobability, flat = dsm.get_chain(), dsm.get_log_prob(), dsm.get_chain(flat=True)
        theta_med = np.percentile(flat, 50, axis=0)
        tup = result_tuple(source, z, bounds, Jup, flux, eflux, popt, pcov, pmin, theta_med, chain,
                           lnprobability, T_d=T_d)
        return tup, summarize(flat, ncomp), dsm
    fn = post.lnprob_batch if lnprob_wrapper is None else lnprob_wrapper(post.lnprob_batch)
    # ONE advancing stream, like the reference's global numpy generator: the sampler continues where the
    # draw of the starting ball stopped instead of replaying it (same seed twice = the same variates)
    sampler = EnsembleSampler(nwalkers, ndim, fn, vectorize=True, seed=None)
    sampler._random = rng
    state = sampler.run_mcmc(pos, n_iter_burn, progress=False)
    sampler.reset()
    sampler.run_mcmc(state, n_iter_walk, progress=False)
    chain = sampler.get_chain()
    lnprobability = sampler.get_log_prob()
    flat = sampler.get_chain(flat=True)
    theta_med = np.percentile(flat, 50, axis=0)
    tup = result_tuple(source, z, bounds, Jup, flux, eflux, popt, pcov, pmin, theta_med, chain,
                       lnprobability, T_d=T_d)
    return tup, summarize(flat, ncomp), sampler


def fit_all(data, sources=None, ncomp=1, nwalkers=None, n_iter_burn=100, n_iter_walk=None, seed=0, engine=None,
            warm=True):
    """The reference's `for source in data.columns:` loop (emcee_radex.py:389-531, emcee_radex_2comp.py:486-611)
    with the chains of ALL sources advancing together: every source gets its own slot of one engine (tbg, line
    list, data, prior box[, T_d]), the warm starts run one source after the other like the reference's, and ONE
    DeviceEnsembleSampler with one ensemble per source runs burn-in and production for all of them in the same
    persistent kernel (BASELINE configs[2]: 16 sources).  Returns {source: (result tuple, summary)} and the sampler.
    The ensemble of source number 0 draws the stream a single-source `fit_source(..., sampler="device")` run draws."""
    from .engine import Engine
    from .sampler import DeviceEnsembleSampler
    names = list(data) if sources is None else list(sources)
    if not 1 <= len(names) <= 60:
        raise ValueError("between 1 and 60 sources per engine (source slots)")
    eng = engine or Engine()
    nwalkers = nwalkers or (100 if ncomp == 1 else 400)
    n_iter_walk = n_iter_walk or (500 if ncomp == 1 else 1000)
    ndim = 4 * ncomp
    rng = np.random.RandomState(seed)
    meta, pos = [], np.empty((len(names), nwalkers, ndim))
    for k, name in enumerate(names):
        if ncomp == 1:
            z, _lw, Jup, flux, eflux = data_io.get_source(name, data)
            T_d, p0 = None, P0_1COMP
        else:
            z, T_d, _lw, Jup, flux, eflux = data_io.get_source(name, data)
            p0 = P0_2COMP
        tbg, bounds = data_io.source_setup(z, ncomp)
        post = Posterior(Jup, flux, eflux, bounds, tbg, ncomp=ncomp, T_d=T_d, engine=eng, src=k)
        if warm:
            popt, pcov, pmin = warm_start(post, p0)
        else:
            popt = np.clip(np.asarray(p0, dtype=float), bounds[:, 0], bounds[:, 1])
            pcov, pmin = None, popt
        pos[k] = popt + 1e-3 * rng.randn(nwalkers, ndim)                 # emcee_radex.py:477
        meta.append((name, z, bounds, T_d, Jup, flux, eflux, popt, pcov, pmin))
    dsm = DeviceEnsembleSampler(nwalkers, ndim, engine=eng, nens=len(names), ens_src=np.arange(len(names)), seed=int(seed))
    state = dsm.run_mcmc(pos if len(names) > 1 else pos[0], n_iter_burn, store=False)
    dsm.reset()
    dsm.run_mcmc(state, n_iter_walk)
    chain, lnp = dsm.get_chain(), dsm.get_log_prob()
    if len(names) == 1:
        chain, lnp = chain[:, None], lnp[:, None]
    out = {}
    for k, (name, z, bounds, T_d, Jup, flux, eflux, popt, pcov, pmin) in enumerate(meta):
        ch, lp = chain[:, k], lnp[:, k]
        flat = ch.reshape(-1, ndim)
        theta_med = np.percentile(flat, 50, axis=0)
        out[name] = (result_tuple(name, z, bounds, Jup, flux, eflux, popt, pcov, pmin, theta_med, ch, lp, T_d=T_d),
                     summarize(flat, ncomp))
    return out, dsm
