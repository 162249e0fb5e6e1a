"""Deterministic synthetic CO-like LAMDA molecular data file.

The reference reads ``radex_moldata/co.dat`` (LAMDA format)
[/root/reference/emcee/emcee_radex.py:108-117, README.md:58], but that file is
not shipped with the reference and is not obtainable offline.  This module
writes a stand-in with the same *shape* (41 levels, 40 lines, 2 collision
partners pH2/oH2, 820 collisional transitions x 25 temperatures) built from
physical closed forms:

* level energies: non-rigid rotor  E_J/h = B J(J+1) - D J^2 (J+1)^2 with the
  CO ground-state constants, g_J = 2J+1;
* Einstein A:  A(J->J-1) = 64 pi^4 nu^3 mu^2 / (3 h c^3) * J/(2J+1), mu = 0.11011 D
  (reproduces the published CO A-values to ~4 digits);
* collision rate coefficients: smooth analytic energy-gap law, slightly
  different for the two H2 spin species, magnitudes of order 1e-11..1e-10 cm^3/s
  as for CO-H2.  These are NOT the Yang et al. (2010) rates.

A real ``co.dat`` is picked up from ``$RADEX_DATAPATH`` when present (same
environment variable as /root/reference/emcee/pyradex/core.py:284-285).
"""
from __future__ import annotations

import math
import os

# CODATA-2018 (exact SI) in cgs
_H = 6.62607015e-27      # erg s
_C = 2.99792458e10       # cm/s
_KB = 1.380649e-16       # erg/K

_B_GHZ = 57.635968       # CO rotational constant
_D_GHZ = 1.835055e-4     # centrifugal distortion
_MU_DEBYE = 0.11011

COLL_TEMPS = [2.0, 3.0, 5.0, 7.0, 10.0, 15.0, 20.0, 30.0, 40.0, 50.0, 60.0, 70.0,
              80.0, 100.0, 150.0, 200.0, 300.0, 400.0, 500.0, 700.0, 1000.0,
              1500.0, 2000.0, 2500.0, 3000.0]

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
SYNTH_CO_PATH = os.path.join(DATA_DIR, "co_synth.dat")


def _energy_ghz(J: int) -> float:
    x = J * (J + 1.0)
    return _B_GHZ * x - _D_GHZ * x * x


def _rate(ju: int, jl: int, T: float, partner: int) -> float:
    """Analytic downward rate coefficient ju->jl [cm^3 s^-1]."""
    dj = ju - jl
    de_k = (_energy_ghz(ju) - _energy_ghz(jl)) * 1e9 * _H / _KB  # gap in K
    if partner == 0:      # para-H2
        k0, b, c, par = 3.2e-11, 0.22, 0.42, 0.20
    else:                 # ortho-H2
        k0, b, c, par = 3.9e-11, 0.27, 0.38, 0.10
    even = 1.0 + (par if dj % 2 == 0 else 0.0)
    gap = math.exp(-0.5 * de_k / (T + 120.0))
    stat = ((2.0 * jl + 1.0) / (2.0 * ju + 1.0)) ** 0.25
    return k0 * (T / 100.0) ** b * math.exp(-c * (dj - 1)) * even * gap * stat * (1.0 + 0.02 * jl)


def synth_co_text(nlev: int = 41) -> str:
    """Return the LAMDA-format text of the synthetic CO-like molecule."""
    out = []
    out.append("!MOLECULE")
    out.append("CO-synthetic (rigid-rotor closed forms; NOT LAMDA co.dat)")
    out.append("!MOLECULAR WEIGHT")
    out.append("28.0")
    out.append("!NUMBER OF ENERGY LEVELS")
    out.append("%d" % nlev)
    out.append("!LEVEL + ENERGIES(cm^-1) + WEIGHT + J")
    for J in range(nlev):
        e_cm = _energy_ghz(J) * 1e9 / _C
        out.append("%5d %17.9f %6.1f %5d" % (J + 1, e_cm, 2.0 * J + 1.0, J))
    out.append("!NUMBER OF RADIATIVE TRANSITIONS")
    out.append("%d" % (nlev - 1))
    out.append("!TRANS + UP + LOW + EINSTEINA(s^-1) + FREQ(GHz) + E_u(K)")
    mu = _MU_DEBYE * 1e-18
    for J in range(1, nlev):
        nu = (_energy_ghz(J) - _energy_ghz(J - 1)) * 1e9
        a = 64.0 * math.pi ** 4 * nu ** 3 * mu * mu / (3.0 * _H * _C ** 3) * J / (2.0 * J + 1.0)
        eup = _energy_ghz(J) * 1e9 * _H / _KB
        out.append("%5d %5d %5d %11.3e %16.7f %10.2f" % (J, J + 1, J, a, nu * 1e-9, eup))
    out.append("!NUMBER OF COLL PARTNERS")
    out.append("2")
    for partner, (pid, name) in enumerate(((2, "pH2"), (3, "oH2"))):
        out.append("!COLLISIONS BETWEEN")
        out.append("%d CO-%s synthetic analytic rates" % (pid, name))
        out.append("!NUMBER OF COLL TRANS")
        out.append("%d" % (nlev * (nlev - 1) // 2))
        out.append("!NUMBER OF COLL TEMPS")
        out.append("%d" % len(COLL_TEMPS))
        out.append("!COLL TEMPS")
        out.append(" ".join("%7.1f" % t for t in COLL_TEMPS))
        out.append("!TRANS + UP + LOW + COLLRATES(cm^3 s^-1)")
        idx = 0
        for ju in range(1, nlev):
            for jl in range(ju):
                idx += 1
                rates = " ".join("%.4e" % _rate(ju, jl, t, partner) for t in COLL_TEMPS)
                out.append("%5d %5d %5d %s" % (idx, ju + 1, jl + 1, rates))
    out.append("")
    return "\n".join(out)


def write_synth_co(path: str = SYNTH_CO_PATH) -> str:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(synth_co_text())
    return path


def default_molfile(species: str = "co") -> str:
    """Resolve the molecular data file the way the reference does
    (datapath / $RADEX_DATAPATH + species + '.dat'), falling back to the
    synthetic CO-like file shipped with this package."""
    dp = os.getenv("RADEX_DATAPATH")
    if dp:
        cand = os.path.join(os.path.expanduser(dp), species + ".dat")
        if os.path.exists(cand):
            return cand
    if species.lower() != "co":
        raise ValueError("no data file for species %r (set RADEX_DATAPATH)" % species)
    if not os.path.exists(SYNTH_CO_PATH):
        write_synth_co(SYNTH_CO_PATH)
    return SYNTH_CO_PATH


if __name__ == "__main__":
    print(write_synth_co())
