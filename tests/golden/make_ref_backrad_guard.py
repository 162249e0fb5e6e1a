#!/usr/bin/env python3
"""Golden vectors for backrad_'s exp-guard branch, by EXECUTING the reference's radex.so (container only).

    python tests/golden/make_ref_backrad_guard.py   ->  tests/golden/ref_backrad_guard.json

ref_backrad.json (make_ref_vectors.py) holds the backgrounds of the matrix_ histories, T_bg >= 2.73 K: none of them
reaches `fk xnu / tbg >= 160 -> backi = 1e-30f` [radex.so@0x1be30, SURVEY A.1].  These vectors do: T_bg from 0.3 K to
the edge of the guard for the highest CO line, for both committed molecules.  The molecule tables are the binary's own
(its readdata_ on the committed LAMDA files); a vector is rejected if any trapped import fired.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.macho_ref import RefRadex                # noqa: E402
from radex_emcee_amd.molecule import SYNTH_CO_PATH   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TOY_PATH = os.path.join(HERE, "toy6.dat")
FK = 1.4387809925261357                              # [BIN 0x26bc8]


def main():
    out = []
    for name, path, dens in (("co_synth", SYNTH_CO_PATH, {2: 2.5e3, 3: 7.5e3}), ("toy6", TOY_PATH, {1: 1e4})):
        R = RefRadex()
        v = R.views()
        R.readdata(path, 30.0, dens)
        nline = int(v["imolec_hdr"][1])
        xmax = float(np.max(v["xnu"][:nline]))
        edge = FK * xmax / 160.0                      # the highest line sits exactly on the guard near this T_bg
        for tbg in (0.3, 0.7, 1.0, float(np.nextafter(edge, 0.0)), edge, float(np.nextafter(edge, 10.0)), 2.0):
            v["tbg"][0] = tbg
            R.backrad()
            assert not R.trap_log, R.trap_log
            out.append(dict(mol=name, tbg=tbg, backi=[float(x) for x in v["backi"][:nline]],
                            totalb=[float(x) for x in v["totalb"][:nline]], trj=[float(x) for x in v["trj"][:nline]]))
            print(name, tbg, "lines on the floor:", int(np.sum(np.array(out[-1]["backi"]) == 1.0000000031710769e-30)))
    json.dump(dict(source="radex.so:_backrad_, exp-guard branch", cases=out),
              open(os.path.join(HERE, "ref_backrad_guard.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
