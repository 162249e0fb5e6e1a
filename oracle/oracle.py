"""ctypes binding of the CPU oracle (oracle/radex_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
RXO_MAXPART = 9

STATUS_OK, STATUS_MAXITER, STATUS_INVALID, STATUS_PRIOR = 0, 1, 2, 3


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "radex_oracle.c")
    hdr = os.path.join(_HERE, "radex_oracle.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Mol(C.Structure):
    _fields_ = [("nlev", C.c_int), ("nline", C.c_int), ("npart", C.c_int),
                ("amass", C.c_double),
                ("eterm", C.POINTER(C.c_double)), ("gstat", C.POINTER(C.c_double)),
                ("iupp", C.POINTER(C.c_int)), ("ilow", C.POINTER(C.c_int)),
                ("aeinst", C.POINTER(C.c_double)), ("spfreq", C.POINTER(C.c_double)),
                ("eup", C.POINTER(C.c_double)), ("xnu", C.POINTER(C.c_double)),
                ("part_id", C.c_int * RXO_MAXPART),
                ("ncoll", C.c_int * RXO_MAXPART), ("ntemp", C.c_int * RXO_MAXPART),
                ("temp", C.POINTER(C.c_double) * RXO_MAXPART),
                ("lcu", C.POINTER(C.c_int) * RXO_MAXPART),
                ("lcl", C.POINTER(C.c_int) * RXO_MAXPART),
                ("coll", C.POINTER(C.c_double) * RXO_MAXPART)]


class _State(C.Structure):
    _fields_ = [("mol", C.POINTER(_Mol)), ("method", C.c_int),
                ("density", C.c_double * RXO_MAXPART),
                ("tkin", C.c_double), ("tbg", C.c_double), ("cdmol", C.c_double),
                ("deltav", C.c_double), ("totdens", C.c_double),
                ("crate", C.POINTER(C.c_double)), ("ctot", C.POINTER(C.c_double)),
                ("xpop", C.POINTER(C.c_double)), ("xpopold", C.POINTER(C.c_double)),
                ("tex", C.POINTER(C.c_double)), ("taul", C.POINTER(C.c_double)),
                ("backi", C.POINTER(C.c_double)), ("totalb", C.POINTER(C.c_double)),
                ("trj", C.POINTER(C.c_double)),
                ("yrate", C.POINTER(C.c_double)), ("rhs", C.POINTER(C.c_double)),
                ("lu", C.POINTER(C.c_double)), ("ipvt", C.POINTER(C.c_int)),
                ("lu_info", C.c_int),
                ("rf_minv", C.POINTER(C.c_double) * 2), ("rf_x", C.POINTER(C.c_double) * 2),
                ("rf_have", C.c_int * 2), ("rf_skip", C.c_int), ("rf_frun", C.c_int),
                ("rf_r", C.POINTER(C.c_double)), ("rf_d", C.POINTER(C.c_double)),
                ("rf_a", C.POINTER(C.c_double)),
                ("rf_full", C.c_long), ("rf_refined", C.c_long), ("rf_steps", C.c_long),
                ("rf_failed", C.c_long), ("rf_kept", C.c_long)]


class _Source(C.Structure):
    _fields_ = [("tbg", C.c_double), ("nJ", C.c_int),
                ("Jup", C.POINTER(C.c_int32)),
                ("flux", C.POINTER(C.c_double)), ("eflux", C.POINTER(C.c_double)),
                ("bounds", C.POINTER(C.c_double)),
                ("ncomp", C.c_int), ("T_d", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.rxo_mol_load.restype = C.POINTER(_Mol)
        L.rxo_mol_load.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.rxo_mol_free.argtypes = [C.POINTER(_Mol)]
        L.rxo_state_new.restype = C.POINTER(_State)
        L.rxo_state_new.argtypes = [C.POINTER(_Mol), C.c_int, C.c_double]
        L.rxo_state_free.argtypes = [C.POINTER(_State)]
        L.rxo_rates.argtypes = [C.POINTER(_State)]
        L.rxo_backrad.argtypes = [C.POINTER(_State), C.c_double]
        L.rxo_escprob.restype = C.c_double
        L.rxo_escprob.argtypes = [C.c_double, C.c_int]
        L.rxo_matrix.argtypes = [C.POINTER(_State), C.c_int, C.POINTER(C.c_int)]
        L.rxo_lubksb.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double),
                                 C.POINTER(C.c_int)]
        L.rxo_run.argtypes = [C.POINTER(_State), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.rxo_source_line_surfbrightness.argtypes = [C.POINTER(_State), C.POINTER(C.c_double)]
        L.rxo_lnprior.restype = C.c_double
        L.rxo_lnprior.argtypes = [C.POINTER(_Source), C.POINTER(C.c_double)]
        for fn in (L.rxo_lnprob_batch, L.rxo_model_flux_batch):
            fn.argtypes = [C.POINTER(_Mol), C.c_int, C.c_double, C.POINTER(_Source), C.c_int,
                           C.POINTER(C.c_double), C.POINTER(C.c_double),
                           C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
        L.rxo_solve_state.argtypes = [C.POINTER(_Mol), C.c_int, C.c_double, C.c_double,
                                      C.POINTER(C.c_double), C.c_double, C.c_double,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.rxo_set_refine.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.rxo_set_refine_componentwise.argtypes = [C.c_double, C.c_double, C.c_double]
        L.rxo_refine_counters.argtypes = [C.POINTER(C.c_long)] * 5 + [C.c_int]
        _lib = L
    return _lib


def set_refine(first_iter=0, tol=1e-10, max_steps=4, lag=2, crit=0, d1max=0.0, loose=0.0, backoff=0):
    """Switch the iterative-refinement variant of matrix_'s step 4 on (first_iter > 0) or off (0) for every
    state created afterwards -- the scheme of the device kernels, restated on the CPU so that it can be
    measured against the reference's arithmetic (radex_oracle.h: rxo_set_refine)."""
    lib().rxo_set_refine(int(first_iter), float(tol), int(max_steps), int(lag), int(crit), float(d1max), float(loose), int(backoff))


def set_refine_componentwise(rel, floor, loose_rel):
    """Thresholds of set_refine(crit=2): per level, relative to its population in the start vector."""
    lib().rxo_set_refine_componentwise(float(rel), float(floor), float(loose_rel))


def refine_counters(reset=True):
    v = [C.c_long(0) for _ in range(5)]
    lib().rxo_refine_counters(*[C.byref(x) for x in v], int(bool(reset)))
    return dict(full=v[0].value, refined=v[1].value, steps=v[2].value, failed=v[3].value, kept=v[4].value)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Source:
    """Per-source observational data + prior box (rxo_source)."""

    def __init__(self, tbg, Jup, flux, eflux, bounds, ncomp=1, T_d=None):
        self.Jup = np.ascontiguousarray(Jup, dtype=np.int32)
        self.flux = np.ascontiguousarray(flux, dtype=np.float64)
        self.eflux = np.ascontiguousarray(eflux, dtype=np.float64)
        self.bounds = np.ascontiguousarray(bounds, dtype=np.float64).reshape(4 * ncomp, 2)
        self.ncomp = int(ncomp)
        self.tbg = float(tbg)
        self.T_d = float("nan") if T_d is None else float(T_d)
        self.c = _Source(self.tbg, len(self.Jup), _ip(self.Jup), _dp(self.flux),
                         _dp(self.eflux), _dp(self.bounds), self.ncomp, self.T_d)


class Molecule:
    def __init__(self, path: str):
        err = C.create_string_buffer(256)
        self.ptr = lib().rxo_mol_load(path.encode(), err, 256)
        if not self.ptr:
            raise ValueError("oracle: %s (%s)" % (err.value.decode(), path))
        m = self.ptr.contents
        self.nlev, self.nline, self.npart = m.nlev, m.nline, m.npart
        self.eterm = np.ctypeslib.as_array(m.eterm, (m.nlev,)).copy()
        self.gstat = np.ctypeslib.as_array(m.gstat, (m.nlev,)).copy()
        self.iupp = np.ctypeslib.as_array(m.iupp, (m.nline,)).copy()
        self.ilow = np.ctypeslib.as_array(m.ilow, (m.nline,)).copy()
        self.aeinst = np.ctypeslib.as_array(m.aeinst, (m.nline,)).copy()
        self.spfreq = np.ctypeslib.as_array(m.spfreq, (m.nline,)).copy()
        self.xnu = np.ctypeslib.as_array(m.xnu, (m.nline,)).copy()
        self.part_id = [m.part_id[i] for i in range(m.npart)]
        self.eup = np.ctypeslib.as_array(m.eup, (m.nline,)).copy()
        self.amass = m.amass

    def partner_tables(self):
        """[(id, temps[ntemp], lcu[ncoll], lcl[ncoll], coll[ncoll, ntemp])] as parsed (rows naming levels above nlev dropped)"""
        m, out = self.ptr.contents, []
        for i in range(m.npart):
            nc, nt = m.ncoll[i], m.ntemp[i]
            as_a = lambda p, n: np.ctypeslib.as_array(p, (n,)).copy() if n else np.zeros(0)
            out.append((m.part_id[i], as_a(m.temp[i], nt), as_a(m.lcu[i], nc), as_a(m.lcl[i], nc),
                        as_a(m.coll[i], nc * nt).reshape(nc, nt)))
        return out

    def __del__(self):
        try:
            if self.ptr:
                lib().rxo_mol_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class State:
    """One private solver state (= one reference pool worker's COMMON blocks)."""

    def __init__(self, mol: Molecule, method: int = 2, deltav_kms: float = 1.0):
        self.mol = mol
        self.ptr = lib().rxo_state_new(mol.ptr, method, deltav_kms)
        self.s = self.ptr.contents

    def __del__(self):
        try:
            if self.ptr:
                lib().rxo_state_free(self.ptr)
                self.ptr = None
        except Exception:
            pass

    def arr(self, name, n=None):
        m = self.mol
        sizes = dict(crate=m.nlev * m.nlev, ctot=m.nlev, xpop=m.nlev, xpopold=m.nlev,
                     tex=m.nline, taul=m.nline, backi=m.nline, totalb=m.nline, trj=m.nline,
                     yrate=m.nlev * m.nlev, rhs=m.nlev)
        return np.ctypeslib.as_array(getattr(self.s, name), (n or sizes[name],))

    def set_density(self, dens_by_id: dict):
        for k in range(RXO_MAXPART):
            self.s.density[k] = 0.0
        for pid, v in dens_by_id.items():
            self.s.density[pid - 1] = float(v)

    def rates(self):
        return lib().rxo_rates(self.ptr)

    def backrad(self, tbg):
        lib().rxo_backrad(self.ptr, float(tbg))

    def matrix(self, niter):
        conv = C.c_int(0)
        lib().rxo_matrix(self.ptr, int(niter), C.byref(conv))
        return conv.value

    def run(self, reuse_last=False, miniter=10, maxiter=200):
        conv = C.c_int(0)
        it = lib().rxo_run(self.ptr, int(reuse_last), miniter, maxiter, C.byref(conv))
        return it, bool(conv.value)

    def surfbrightness(self):
        out = np.empty(self.mol.nline)
        lib().rxo_source_line_surfbrightness(self.ptr, _dp(out))
        return out


def escprob(tau, method=2):
    return lib().rxo_escprob(float(tau), int(method))


def lubksb(a_colmajor: np.ndarray, return_ipvt: bool = False):
    """a: (n,n) array interpreted as Fortran A(i,j) = a[i,j]; returns (x, info[, ipvt 0-based])."""
    n = a_colmajor.shape[0]
    buf = np.asfortranarray(a_colmajor, dtype=np.float64).copy(order="F")
    flat = buf.reshape(-1, order="F").copy()
    x = np.empty(n)
    ipvt = np.empty(n, dtype=np.int32)
    info = lib().rxo_lubksb(_dp(flat), n, _dp(x), ipvt.ctypes.data_as(C.POINTER(C.c_int)))
    if return_ipvt:
        return x, info, ipvt
    return x, info


def lnprior(src: Source, p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    return lib().rxo_lnprior(C.byref(src.c), _dp(p))


def lnprob_batch(mol: Molecule, src: Source, params, method=2, deltav_kms=1.0, nthreads=1):
    params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4 * src.ncomp)
    N = params.shape[0]
    lnp = np.empty(N)
    status = np.empty(N, dtype=np.int32)
    niter = np.empty(N, dtype=np.int32)
    rc = lib().rxo_lnprob_batch(mol.ptr, method, deltav_kms, C.byref(src.c), N, _dp(params),
                                _dp(lnp), _ip(status), _ip(niter), nthreads)
    if rc:
        raise RuntimeError("rxo_lnprob_batch rc=%d" % rc)
    return lnp, status, niter


def model_flux_batch(mol: Molecule, src: Source, params, method=2, deltav_kms=1.0, nthreads=1):
    params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4 * src.ncomp)
    N = params.shape[0]
    flux = np.empty((N, len(src.Jup)))
    status = np.empty(N, dtype=np.int32)
    niter = np.empty(N, dtype=np.int32)
    lib().rxo_model_flux_batch(mol.ptr, method, deltav_kms, C.byref(src.c), N, _dp(params),
                               _dp(flux), _ip(status), _ip(niter), nthreads)
    return flux, status, niter


def solve_state(mol: Molecule, tbg, dens_by_id, tkin, cdmol, method=2, deltav_kms=1.0):
    dens = np.zeros(RXO_MAXPART)
    for pid, v in dens_by_id.items():
        dens[pid - 1] = v
    xpop = np.empty(mol.nlev)
    tex = np.empty(mol.nline)
    tau = np.empty(mol.nline)
    conv = C.c_int(0)
    it = lib().rxo_solve_state(mol.ptr, method, deltav_kms, float(tbg), _dp(dens), float(tkin),
                               float(cdmol), _dp(xpop), _dp(tex), _dp(tau), C.byref(conv))
    return dict(niter=it, converged=bool(conv.value), xpop=xpop, tex=tex, tau=tau)
