"""Flat LCDM distances used for the per-source size prior.

Replaces `FlatLambdaCDM(H0=67.8, Om0=0.308).angular_diameter_distance(z)`
[/root/reference/emcee/emcee_radex.py:93,422]; astropy is not a dependency here.
Golden values (astropy 4.3.1) are pinned in tests/test_host_logic.py.
"""
from __future__ import annotations

import math

from scipy.integrate import quad

C_KMS = 299792.458
H0 = 67.8
OM0 = 0.308


def _inv_efunc(z: float) -> float:
    zp1 = 1.0 + z
    return 1.0 / math.sqrt(zp1 ** 3 * OM0 + (1.0 - OM0))


def angular_diameter_distance(z: float) -> float:
    """Mpc.  No radiation term (astropy's default Tcmb0 = 0)."""
    dc = (C_KMS / H0) * quad(_inv_efunc, 0.0, z)[0]
    return dc / (1.0 + z)


def R_angle(z: float) -> float:
    """Solid angle [sr] of a 7 kpc-radius disk magnified by mu = 10 (emcee_radex.py:422)."""
    return ((7.0 / (angular_diameter_distance(z) * 1000.0)) ** 2 * math.pi) * 10.0


def log10_R_angle(z: float) -> float:
    return math.log10(R_angle(z))
