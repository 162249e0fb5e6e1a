"""`Radex`: host-side mirror of the slice of `pyradex.Radex` that the hot path uses.

Same constructor keywords, `set_params`, `run_radex` and state properties as
/root/reference/emcee/pyradex/core.py:195-1091 (only the members on the call stack of
`model_lvg`: SURVEY.md section 3b), with the same argument meaning and the same
`ValueError`s, but every solve runs in libradex_emcee_amd.so on the GPU.  One object
evaluates one parameter set per `run_radex()` (N = 1 batch), which is what
`scipy.optimize.curve_fit`/`minimize` and `replot` need; the sampler uses the batched
`lnprob_batch` instead.

Differences, all deliberate:
  * cold start only.  The reference's drivers pass `reuse_last=True`, which starts from
    whatever state the worker's previous walker left behind; a stateless batched engine
    has no such history (DESIGN.md "History dependence").  `reuse_last` is accepted and
    ignored.
  * values are plain floats / numpy arrays in the reference's units; when astropy is
    installed `source_line_surfbrightness` is wrapped in the same Quantity unit.
"""
from __future__ import annotations

import math
import os

import numpy as np

from .engine import Engine

_VALID = {'H2': 'H2', 'PH2': 'pH2', 'OH2': 'oH2', 'E': 'e', 'H': 'H', 'HE': 'He', 'H+': 'H+'}
_IDS = {'H2': 1, 'PH2': 2, 'OH2': 3, 'E': 4, 'H': 5, 'HE': 6, 'H+': 7}        # core.py:476-498


def _brightness_unit():
    try:
        from astropy import units as u
        return u.erg * u.s ** -1 * u.cm ** -2 * u.Hz ** -1 * u.sr ** -1
    except Exception:
        return None


class Radex:
    def __init__(self, collider_densities=None, density=None, total_density=None, temperature=None,
                 species='co', column=None, column_per_bin=None, tbackground=2.7315, deltav=1.0,
                 abundance=None, datapath=None, escapeProbGeom='lvg', outfile='radex.out',
                 logfile='radex.log', debug=False, mu=2.8, source_area=None, device=0):
        if os.getenv('RADEX_DATAPATH') and datapath is None:
            datapath = os.getenv('RADEX_DATAPATH')
        molfile = None
        if datapath is not None:
            cand = os.path.join(os.path.expanduser(datapath), species + '.dat')
            if os.path.exists(cand):
                molfile = cand
            elif species.lower() != 'co':
                raise ValueError("Must specify a valid path to a molecular data file "
                                 "else RADEX will crash.  Current path is {0}".format(cand))
        if sum(x is not None for x in (collider_densities, density, total_density)) > 1:
            raise ValueError("Can only specify one of density, total_density, and collider_densities")
        if sum(x is not None for x in (column, column_per_bin)) > 1:
            raise ValueError("Can only specify one of column, column_per_bin.")
        n_spec = sum(x is not None for x in (column, column_per_bin, collider_densities, density,
                                             total_density, abundance))
        if n_spec > 2:
            raise ValueError("Can only specify two of column, density, and abundance.")
        if n_spec < 2:
            raise ValueError("Must specify two of column, density, and abundance.")
        if abundance is not None:
            raise NotImplementedError("abundance-locked parameters are outside the hot path")
        self._eng = Engine(molfile=molfile, species=species, escapeProbGeom=escapeProbGeom,
                           deltav=float(deltav), device=device)
        self._geom = escapeProbGeom
        self.deltav = float(deltav)
        self.mu = mu
        self.source_area = source_area
        self.miniter, self.maxiter = 10, 200                     # core.py:460-463
        self._use_thermal_opr = False
        self._tkin = float(temperature)                          # "MUST happen before density is set"
        self._dens = {k: 0.0 for k in _VALID.values()}
        self.density = collider_densities or total_density or density
        self.column_per_bin = column if column is not None else column_per_bin
        self.temperature = temperature
        self.tbg = tbackground
        self._state = None

    # --- pyradex API -----------------------------------------------------------------
    def set_params(self, density=None, collider_densities=None, column=None, column_per_bin=None,
                   temperature=None, abundance=None, species=None, deltav=None, tbg=None,
                   escapeProbGeom=None):
        if species is not None or abundance is not None:
            raise NotImplementedError("changing species/abundance is outside the hot path")
        if deltav is not None and float(deltav) != self.deltav:
            raise NotImplementedError("deltav is fixed at construction (one engine per line width)")
        if escapeProbGeom is not None and escapeProbGeom != self._geom:
            raise NotImplementedError("escapeProbGeom is fixed at construction")
        if temperature is not None:
            self._tkin = float(temperature)                      # core.py:401-402
        if collider_densities is not None:
            self.density = collider_densities
        elif density is not None:
            self.density = density
        if column is not None:
            self.column = column
        elif column_per_bin is not None:
            self.column_per_bin = column_per_bin
        if temperature is not None:
            self.temperature = temperature
        if tbg is not None:
            self.tbg = tbg

    @property
    def valid_colliders(self):
        names = {v: k for k, v in _IDS.items()}
        return [_VALID[names[i]] for i in self._eng.partner_ids]

    @property
    def density(self):
        return dict(self._dens)

    @density.setter
    def density(self, collider_density):                          # core.py:489-579
        if isinstance(collider_density, (float, int, np.floating, np.integer)):
            collider_density = {'H2': collider_density}
        cd = {}
        for k, v in collider_density.items():
            if k.upper() not in _VALID:
                raise ValueError('Collider %s is not one of the valid colliders: %s' % (k, _VALID))
            cd[k.upper()] = float(v)
        d = {k: 0.0 for k in _VALID.values()}
        self._use_thermal_opr = False
        if cd.get('OH2', 0) != 0 or cd.get('PH2', 0) != 0:
            d['pH2'] = cd.get('PH2', self._dens.get('pH2', 0.0) if 'PH2' not in cd else 0.0)
            d['oH2'] = cd.get('OH2', self._dens.get('oH2', 0.0) if 'OH2' not in cd else 0.0)
        elif 'H2' in cd:
            self._use_thermal_opr = True
            T = self._tkin
            opr = min(3.0, 9.0 * math.exp(-170.6 / T)) if T > 0 else 3.0     # core.py:541-546
            fortho = opr / (1 + opr)
            d['pH2'] = cd['H2'] * (1 - fortho)
            d['oH2'] = cd['H2'] * fortho
        vc = [x.lower() for x in self.valid_colliders]
        if 'h2' in vc:                                            # core.py:551-556
            d['H2'] = d['pH2'] + d['oH2']
            d['pH2'] = d['oH2'] = 0.0
        d['e'], d['H'], d['He'], d['H+'] = (cd.get('E', 0.0), cd.get('H', 0.0), cd.get('HE', 0.0),
                                            cd.get('H+', 0.0))
        self._dens = d
        self._validate_colliders()

    def _validate_colliders(self):                                # base_class.py:224-263
        vc = self.valid_colliders
        if not any(self._dens[c] > 0 for c in vc):
            raise ValueError("The colliders in the data file have density 0.")
        lower = [c.lower() for c in vc]
        bad = [c for c, v in self._dens.items() if v > 0 and c.lower() not in lower
               and not (c.lower() in ('oh2', 'ph2') and 'h2' in lower)
               and not (c.lower() == 'h2' and ('oh2' in lower or 'ph2' in lower))]
        if bad:
            raise ValueError("There are colliders with specified densities >0 that do not have "
                             "corresponding collision rates.  The bad colliders are {0}".format(bad))

    @property
    def total_density(self):
        return sum(self._dens.values())

    @property
    def temperature(self):
        return self._tkin

    @temperature.setter
    def temperature(self, tkin):                                  # core.py:727-753
        if tkin is None:
            raise TypeError("Must specify tkin")
        tkin = float(tkin)
        if tkin <= 0 or tkin > 1e4:
            raise ValueError('Must have kinetic temperature > 0 and < 10^4 K')
        self._tkin = tkin
        if self._use_thermal_opr:
            self.density = self._dens['H2'] or (self._dens['oH2'] + self._dens['pH2'])

    @property
    def column(self):
        return self._col

    @column.setter
    def column(self, value):
        self.column_per_bin = value

    @property
    def column_per_bin(self):
        return self._col

    @column_per_bin.setter
    def column_per_bin(self, col):                                # core.py:767-787
        col = float(col)
        if col < 1e5 or col > 1e25:
            raise ValueError("Extremely low or extremely high column.")
        self._col = col

    @property
    def tbg(self):
        return self._tbg

    @tbg.setter
    def tbg(self, tbg):                                           # core.py:845-854 -> backrad_
        if tbg is None:
            return
        self._tbg = float(tbg)
        self._eng.set_source(self._tbg)

    @property
    def escapeProbGeom(self):
        return self._geom

    def run_radex(self, silent=True, reuse_last=False, reload_molfile=True,
                  abs_convergence_threshold=1e-16, rel_convergence_threshold=1e-8,
                  validate_colliders=True):                       # core.py:856-925
        if validate_colliders:
            self._validate_colliders()
        names = {v: k for k, v in _IDS.items()}
        dens = [[self._dens[_VALID[names[i]]] for i in self._eng.partner_ids]]
        self._eng.set_iteration_limits(self.miniter, self.maxiter)
        out = self._eng.solve_batch([self._tkin], [self._col], dens)
        self._state = {k: v[0] for k, v in out.items()}
        self._iter_counter = int(self._state["niter"])
        if not silent:
            if self._state["status"] == 1:
                print("Did not converge in %i iterations, stopping." % self.maxiter)
            else:
                print("Successfully converged after %i iterations" % self._iter_counter)
        return self._iter_counter

    def __call__(self, return_table=False, **kwargs):
        self.set_params(**kwargs)
        niter = self.run_radex(reload_molfile=False, validate_colliders=False)
        return self.get_table() if return_table else niter

    def get_table(self):
        s = self._need()
        return [dict(Tex=s["tex"][l], tau=s["tau"][l], frequency=self._eng.spfreq[l],
                     upperlevelpop=s["xpop"][self._eng.iupp[l] - 1],
                     lowerlevelpop=s["xpop"][self._eng.ilow[l] - 1],
                     brightness=s["sb"][l]) for l in range(self._eng.nline)]

    def _need(self):
        if self._state is None:
            raise RuntimeError("run_radex() has not been called")
        return self._state

    @property
    def level_population(self):
        return self._need()["xpop"]

    @property
    def tex(self):
        return self._need()["tex"]

    Tex = tex

    @property
    def tau(self):
        return self._need()["tau"]

    @property
    def frequency(self):
        return self._eng.spfreq

    @property
    def upperlevelpop(self):
        return self.level_population[self._eng.iupp - 1]

    @property
    def lowerlevelpop(self):
        return self.level_population[self._eng.ilow - 1]

    @property
    def source_line_surfbrightness(self):                         # base_class.py:275-277
        sb = self._need()["sb"]
        unit = _brightness_unit()
        return sb if unit is None else sb * unit
