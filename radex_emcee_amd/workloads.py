"""Synthetic workloads of BASELINE.json / SURVEY.md section 8(d).

Only shapes, bounds and seeds live here; model fluxes ("truth" SLEDs) are
produced by whichever evaluator the caller passes in; this module evaluates nothing.
"""
from __future__ import annotations

import numpy as np

from .cosmology import log10_R_angle

T_CMB0 = 2.7315          # emcee_radex.py:419


def bounds_1comp(z: float) -> np.ndarray:
    """emcee/emcee_radex.py:439-442."""
    lra = log10_R_angle(z)
    return np.array([[2.0, 7.0],
                     [np.log10(T_CMB0 * (1 + z)), 3.0],
                     [15.5, 19.5],
                     [lra - 4, lra + 4]])


def bounds_2comp(z: float) -> np.ndarray:
    """emcee/emcee_radex_2comp.py:501-510."""
    lra = log10_R_angle(z)
    one = [[1.5, 7.0], [np.log10(T_CMB0 * (1 + z)), 3.0], [14.5, 19.5], [lra - 9, lra + 9]]
    return np.array(one + one)


def draw_prior_1comp(bounds: np.ndarray, n: int, seed: int) -> np.ndarray:
    """Uniform in the prior box intersected with 10 < p2 - p0 < 17.5 so that every
    walker reaches the solver."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 4))
    k = 0
    lo, hi = bounds[:, 0], bounds[:, 1]
    while k < n:
        p = lo + (hi - lo) * rng.random(4)
        if 10.0 < p[2] - p[0] < 17.5:
            out[k] = p
            k += 1
    return out


def draw_prior_2comp(bounds: np.ndarray, n: int, seed: int) -> np.ndarray:
    """Uniform in the 8-dimensional box intersected with the hard constraints of
    emcee_radex_2comp.py:199-234 (T2 > T1, size1 >= size2, 9 < N_i - n_i < 18)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 8))
    k = 0
    lo, hi = bounds[:, 0], bounds[:, 1]
    while k < n:
        p = lo + (hi - lo) * rng.random(8)
        if p[5] > p[1] and p[3] >= p[7] and 9.0 < p[2] - p[0] < 18.0 and 9.0 < p[6] - p[4] < 18.0:
            out[k] = p
            k += 1
    return out


def config2(n_walkers: int = 1024, seed: int = 1234):
    """Synthetic CO SLED J=1..10, 1 component, z=2.5, truth [3.5, 2.0, 17.5, -9.5]."""
    z = 2.5
    b = bounds_1comp(z)
    # the truth's size is quoted as -9.5 in SURVEY 8d; the prior box is centred on R_angle(z)
    return dict(z=z, tbg=T_CMB0 * (1 + z), Jup=np.arange(1, 11, dtype=np.int32), bounds=b,
                truth=np.array([3.5, 2.0, 17.5, -9.5]), ncomp=1, T_d=None,
                walkers=draw_prior_1comp(b, n_walkers, seed))


def config3(n_walkers: int = 1024, seed: int = 3333, init: str = "prior", filename=None):
    """BASELINE configs[2]: all 16 sources of data/flux.dat (/root/reference/data/flux.dat:8-23),
    1 component, `n_walkers` walkers each, advanced together: per source z, Jup/flux/eflux (the finite
    CO columns), tbg = 2.7315 (1+z) and the prior box of emcee_radex.py:419-442.

    walkers[16][n_walkers][4]: init="prior": uniform in each source's prior box intersected with
    10 < p2 - p0 < 17.5 (every walker reaches the solver; the stress shape, like config 2);
    init="ball": the reference's start, p0 + 1e-3 N(0,1) around its p0 = [4.0, 1.4, 17.8, -9.85]
    clipped into the box (emcee_radex.py:444-451,477).  Source k lives in slot k; src_index[16*n_walkers]
    is the per-walker slot of the flattened batch."""
    from . import data_io
    data = data_io.read_data(filename or data_io.FLUX_1COMP)
    names = list(data)
    sources, walkers = [], []
    for k, name in enumerate(names):
        z, _lw, Jup, flux, eflux = data_io.get_source(name, data)
        tbg, b = data_io.source_setup(z, 1)
        sources.append(dict(name=name, z=z, tbg=tbg, Jup=Jup.astype(np.int32), flux=flux, eflux=eflux,
                            bounds=b, ncomp=1, T_d=None, slot=k))
        if init == "prior":
            walkers.append(draw_prior_1comp(b, n_walkers, seed + k))
        elif init == "ball":
            rng = np.random.default_rng(seed + k)
            p0 = np.clip(np.array([4.0, 1.4, 17.8, -9.85]), b[:, 0], b[:, 1])
            walkers.append(p0 + 1e-3 * rng.standard_normal((n_walkers, 4)))
        else:
            raise ValueError("init must be 'prior' or 'ball'")
    walkers = np.array(walkers)
    src_index = np.repeat(np.arange(len(names), dtype=np.int32), n_walkers)
    return dict(sources=sources, names=names, walkers=walkers, src_index=src_index, ncomp=1,
                n_walkers=n_walkers)


def config4(n_walkers: int = 2048, seed: int = 4321):
    """Synthetic 2-component workload (emcee_radex_2comp priors), z=2.5, T_d=40."""
    z = 2.5
    b = bounds_2comp(z)
    truth = np.array([1.9, 1.2, 16.4, -12.1, 3.9, 2.5, 17.5, -12.1])   # 2comp:513-522
    rng = np.random.default_rng(seed)
    walkers = truth + 1e-3 * rng.standard_normal((n_walkers, 8))          # 2comp:553
    return dict(z=z, tbg=T_CMB0 * (1 + z), Jup=np.arange(1, 11, dtype=np.int32), bounds=b,
                truth=truth, ncomp=2, T_d=40.0, walkers=walkers)


def config1(n_walkers: int = 400, seed: int = 0):
    """APM08279+5255 stand-in (SLED not in the reference): z=3.911."""
    z = 3.911
    b = bounds_1comp(z)
    truth = np.array([4.2, 2.4, 17.5, log10_R_angle(z)])
    rng = np.random.default_rng(seed)
    walkers = truth + 1e-3 * rng.standard_normal((n_walkers, 4))          # emcee_radex.py:477
    return dict(z=z, tbg=T_CMB0 * (1 + z), Jup=np.array([1, 2, 4, 6, 9, 10, 11], dtype=np.int32),
                bounds=b, truth=truth, ncomp=1, T_d=None, walkers=walkers)
