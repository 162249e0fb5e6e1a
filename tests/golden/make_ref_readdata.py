#!/usr/bin/env python3
"""Golden vectors for readdata_ -- row a6 of SURVEY.md section 8 -- computed by EXECUTING the reference's own
readdata_ (radex.so@0x1cf90, called by the reference at emcee/pyradex/core.py:570,744) on the committed
LAMDA-format files (container only).

    python tests/golden/make_ref_readdata.py        -> tests/golden/ref_readdata.json

oracle/macho_ref.py maps the Mach-O image and serves the libgfortran OPEN / READ / CLOSE calls readdata_ makes
(list-directed reads and two formats) on the real file; everything else -- the positional parsing, xnu = E_up - E_low,
the temperature bracket, the linear interpolation, crate += density * rate, detailed balance, ctot -- is the
reference's machine code.  A case is rejected if any trap fired (a STOP, an I/O form the shim does not implement).

Per case: molecule file, tkin, density by partner id -> sha256 over the little-endian bytes of crate[nlev][nlev]
(row i = rates out of level i) followed by ctot[nlev]; the full arrays are stored for a subset so that a
failure can be looked at.  Per molecule: the parsed tables (eterm, gstat, iupp, ilow, aeinst, spfreq, xnu).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.macho_ref import RefRadex, MAXLEV   # noqa: E402
from radex_emcee_amd.molecule import SYNTH_CO_PATH   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = {"co_synth": SYNTH_CO_PATH, "toy6": os.path.join(HERE, "toy6.dat")}


def fl(a):
    return [float(x) for x in np.asarray(a, dtype=np.float64).ravel()]


def digest(crate, ctot):
    return hashlib.sha256(np.ascontiguousarray(crate, dtype="<f8").tobytes()
                          + np.ascontiguousarray(ctot, dtype="<f8").tobytes()).hexdigest()


def main():
    # the (molecule, tkin, density) points of the matrix_ histories (ref_matrix.json) ...
    pts = [(c["mol"], c["tkin"], {int(k): x for k, x in c["density"].items()})
           for c in json.load(open(os.path.join(HERE, "ref_matrix.json")))["cases"]]
    # ... and the corners of the interpolation: on a grid point, on the first / last one, just inside, far outside,
    # one partner without density, an exponent written with `d` (toy6.dat, second rate row)
    co_T = [2.0, 5.0, 10.0, 20.0, 30.0, 50.0, 70.0, 100.0, 150.0, 200.0, 300.0, 500.0, 700.0, 1000.0, 2000.0, 3000.0]
    for T in (co_T[0], co_T[-1], 20.0, 100.0, np.nextafter(co_T[0], 9.0), np.nextafter(co_T[-1], 0.0), 0.5, 9999.0, 47.11):
        pts.append(("co_synth", float(T), {2: 2.5e3, 3: 7.5e3}))
    pts.append(("co_synth", 77.7, {2: 0.0, 3: 1e5}))
    pts.append(("co_synth", 77.7, {2: 1e5, 3: 0.0}))
    for T in (10.0, 30.0, 300.0, 29.999, 5.0, 301.0, 64.0):
        pts.append(("toy6", T, {1: 1e4}))
    images, mols, cases = {}, {}, []
    for k, (name, tkin, dens) in enumerate(pts):
        R = images.setdefault(name, RefRadex())
        v = R.views()
        R.readdata(FILES[name], tkin, dens)
        assert not R.trap_log, (name, tkin, R.trap_log)
        n, nl = int(v["imolec_hdr"][0]), int(v["imolec_hdr"][1])
        cr = v["crate"]
        crate = np.array([[cr[j * MAXLEV + i] for j in range(n)] for i in range(n)])
        ctot = v["ctot"][:n].copy()
        if name not in mols:
            mols[name] = dict(file=os.path.relpath(FILES[name], ROOT), nlev=n, nline=nl,
                              ncoll_last_partner=int(v["imolec_hdr"][2]), npart=int(v["imolec_hdr"][3]),
                              eterm=fl(v["eterm"][:n]), gstat=fl(v["gstat"][:n]),
                              iupp=[int(x) for x in v["iupp"][:nl]], ilow=[int(x) for x in v["ilow"][:nl]],
                              aeinst=fl(v["aeinst"][:nl]), spfreq=fl(v["spfreq"][:nl]), xnu=fl(v["xnu"][:nl]))
        c = dict(mol=name, tkin=float(tkin), density={str(i): float(x) for i, x in dens.items()},
                 sha256=digest(crate, ctot), ctot=fl(ctot),
                 warnings=sorted(set(m.strip() for m in R.io.messages)))
        R.io.messages.clear()
        if name == "toy6" or k % 6 == 0:
            c["crate"] = fl(crate)
        cases.append(c)
        print(name, tkin, dens, c["sha256"][:12], c["warnings"])
    json.dump(dict(source="radex.so:_readdata_ executed on the committed LAMDA-format files (oracle/macho_ref.py)",
                   layout="crate row-major [nlev][nlev], crate[i][j] = rate i -> j; sha256 over crate then ctot, little-endian f64",
                   molecules=mols, cases=cases),
              open(os.path.join(HERE, "ref_readdata.json"), "w"), indent=0)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
