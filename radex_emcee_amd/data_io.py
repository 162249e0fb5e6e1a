"""Input tables and per-source set-up.

`read_data` / `get_source` follow /root/reference/emcee/emcee_radex.py:183-240 and
emcee/emcee_radex_2comp.py:247-279 (the 2-component table carries an extra T_dust column);
`source_setup` follows emcee_radex.py:419-442 and emcee_radex_2comp.py:498-510.
The tables themselves (radex_emcee_amd/data/flux.dat, flux_for2p.dat) are the published
measurements of Yang et al. 2017 shipped with the reference as data/flux*.dat.
"""
from __future__ import annotations

import os

import numpy as np

from . import workloads

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
FLUX_1COMP = os.path.join(DATA_DIR, "flux.dat")
FLUX_2COMP = os.path.join(DATA_DIR, "flux_for2p.dat")


def read_data(filename=FLUX_1COMP):
    """Return {source: {column: value}}.  Layout: SOURCE z D_L [T_d] line_width, then
    (flux, err) pairs for CO J=1..n, then CI(1-0), CI(2-1) pairs.  '#' lines are comments."""
    rows = []
    with open(filename) as f:
        for line in f:
            s = line.strip()
            if s and not s.startswith("#"):
                rows.append(s.split())
    if not rows:
        raise ValueError("no data rows in %s" % filename)
    ncol = len(rows[0])
    if any(len(r) != ncol for r in rows):
        raise ValueError("Number of columns in data rows does not match the expected number of columns")
    two = (ncol - 8) % 2 == 1                     # the extra T_d column of flux_for2p.dat
    fixed = ["SOURCE", "z", "D_L"] + (["T_d"] if two else []) + ["line_width"]
    nco = (ncol - len(fixed) - 4) // 2
    cols = list(fixed)
    for i in range(nco):
        cols += ["CO_J_%d" % (i + 1), "eCO_J_%d" % (i + 1)]
    cols += ["CI_1", "eCI_1", "CI_2", "eCI_2"]
    out = {}
    for r in rows:
        d = {}
        for name, tok in zip(cols[1:], r[1:]):
            try:
                d[name] = float(tok)
            except ValueError:
                d[name] = float("nan")            # pd.to_numeric(errors='coerce')
        out[r[0]] = d
    return out


def get_source(source, data):
    """-> z, line_width, Jup[int], flux, eflux (only finite CO columns; Jup = column index + 1);
    with a T_d column present: z, T_d, line_width, Jup, flux, eflux (2-component order)."""
    d = data[source]
    keys = [k for k in d if "CO" in k and "eCO" not in k]
    sel = [(j + 1, d[k], d["e" + k]) for j, k in enumerate(keys) if np.isfinite(d[k])]
    Jup = np.array([s[0] for s in sel], dtype=int)
    flux = np.array([s[1] for s in sel], dtype=float)
    eflux = np.array([s[2] for s in sel], dtype=float)
    if "T_d" in d:
        return d["z"], d["T_d"], d["line_width"], Jup, flux, eflux
    return d["z"], d["line_width"], Jup, flux, eflux


def source_setup(z, ncomp=1):
    """tbg = 2.7315 (1+z); prior box from the 7 kpc / mu=10 solid angle."""
    tbg = workloads.T_CMB0 * (1 + z)
    bounds = workloads.bounds_1comp(z) if ncomp == 1 else workloads.bounds_2comp(z)
    return tbg, bounds
