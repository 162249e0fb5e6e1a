#!/usr/bin/env python3
"""BASELINE's own workloads evaluated by the REFERENCE BINARY (container only).

    python tests/golden/make_ref_configs.py [--procs 4]

Writes tests/golden/ref_configs.npz: for every walker of
  * config 2  -- the 1024 prior-box walkers of workloads.config2(1024, 1234), the bench.py headline batch,
  * config 4  -- both components of the 2048 two-component walkers of workloads.config4(2048),
  * config 3  -- the first 64 prior-box walkers of each of the 16 sources of data/flux.dat,
the cold-start answer of /root/reference/emcee/pyradex/radex/radex.so itself: its readdata_ parses
co_synth.dat and forms the rates, backrad_ the background, matrix_ is driven exactly as
emcee/pyradex/core.py:896-925 drives it with reuse_last=False (niter = 0 first, maxiter 200, the
population test with iter > 10).  Stored per solve: niter, conv, and for the first NKEEP lines
(J_up = 1..11, everything flux.dat observes) tex, taul, backi as the binary left them in COMMON /radi/
plus the line surface brightness core.py:986-1003 / base_class.py:275-277 forms from them (astropy's
constants, SURVEY A.6 -- numpy arithmetic on the binary's numbers, not the oracle).

Nothing the oracle computes enters: oracle/ is imported only for the Mach-O loader.  A solve is
rejected (the script aborts) if any trapped import (Fortran I/O outside readdata_'s forms, STOP) fired.
One mapped image per worker process, like one reference pool worker (emcee_radex.py:480-482).
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.macho_ref import RefRadex                     # noqa: E402
from radex_emcee_amd import workloads                     # noqa: E402
from radex_emcee_amd.molecule import SYNTH_CO_PATH        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NKEEP = 11
NLEV, NLINE = 41, 40
THC_PY, FK_PY = 3.9728917142978573e-16, 1.4387768775039338   # core.py:981-984 (astropy CODATA-2018)

_R = None


def _image():
    global _R
    if _R is None:
        _R = RefRadex()
    return _R


def solve(job):
    """job = (log10 n, log10 T, log10 N, tbg) -> one row of numbers from the binary."""
    p0, p1, p2, tbg = job
    R = _image()
    v = R.views()
    n_h2 = 10.0 ** p0
    dens = {2: 0.25 * n_h2, 3: 0.75 * n_h2}                # emcee_radex.py:95-96,124-126: opr = 3
    R.readdata(SYNTH_CO_PATH, 10.0 ** p1, dens)
    v["method"][0] = 2
    v["cdmol"][0] = 10.0 ** p2
    v["deltav"][0] = 1e5                                   # core.py:451-454
    v["tbg"][0] = tbg
    R.backrad()
    v["xpop"][:NLEV] = 0.0
    v["tex"][:NLINE] = 0.0
    v["taul"][:NLINE] = 0.0
    it, conv = 0, 0
    last = v["xpop"][:NLEV].copy()
    while not conv:                                        # core.py:903-920
        if it >= 200:
            break
        conv = R.matrix(it, conv)
        x = v["xpop"][:NLEV]
        if np.abs(last - x).sum() < 1e-16 and it > 10:
            break
        last = x.copy()
        it += 1
    if R.trap_log:
        raise RuntimeError("trap fired: %r" % (R.trap_log,))
    tex = v["tex"][:NKEEP].copy()
    taul = v["taul"][:NKEEP].copy()
    backi = v["backi"][:NKEEP].copy()
    xnu = v["xnu"][:NKEEP].copy()
    with np.errstate(all="ignore"):                        # core.py:986-1003 - background
        ftau = np.exp(-taul)
        bnutex = THC_PY * xnu ** 3 / (np.exp(FK_PY * xnu / tex) - 1.0)
        sb = backi * ftau + bnutex * (1.0 - ftau) - backi
    return np.concatenate(([float(it), float(conv)], tex, taul, backi, sb))


def run(jobs, procs):
    t0 = time.time()
    with mp.Pool(procs) as pool:
        rows = []
        for k, r in enumerate(pool.imap(solve, jobs, chunksize=8)):
            rows.append(r)
            if (k + 1) % 256 == 0:
                print("  %d / %d solves, %.0f s" % (k + 1, len(jobs), time.time() - t0), flush=True)
    a = np.array(rows)
    return dict(niter=a[:, 0].astype(np.int32), conv=a[:, 1].astype(np.int32),
                tex=a[:, 2:2 + NKEEP], taul=a[:, 2 + NKEEP:2 + 2 * NKEEP],
                backi=a[:, 2 + 2 * NKEEP:2 + 3 * NKEEP], sb=a[:, 2 + 3 * NKEEP:2 + 4 * NKEEP])


def save_deterministic(path, arrays):
    """An .npz whose bytes depend on the arrays alone (np.savez stamps every member with the current time): members in sorted
    order, the zip epoch as their date, deflate at a fixed level -- so that regenerating the fixture reproduces it byte for byte."""
    import io
    import zipfile
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED, compresslevel=6) as z:
        for name in sorted(arrays):
            buf = io.BytesIO()
            np.lib.format.write_array(buf, np.ascontiguousarray(arrays[name]), allow_pickle=False)
            info = zipfile.ZipInfo(name + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            info.external_attr = 0o644 << 16
            z.writestr(info, buf.getvalue(), compresslevel=6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(HERE, "ref_configs.npz"))
    ap.add_argument("--repack", action="store_true", help="rewrite an existing --out deterministically (no solves)")
    a = ap.parse_args()
    if a.repack:
        old = np.load(a.out)
        save_deterministic(a.out, {k: old[k] for k in old.files})
        print("repacked", a.out, os.path.getsize(a.out), "bytes")
        return
    out = {}

    c2 = workloads.config2(1024, 1234)
    print("config 2: 1024 walkers", flush=True)
    r = run([(p[0], p[1], p[2], c2["tbg"]) for p in c2["walkers"]], a.procs)
    out["c2_params"] = c2["walkers"]
    out.update({"c2_" + k: x for k, x in r.items()})

    c4 = workloads.config4(2048)
    print("config 4: 2 x 2048 components", flush=True)
    comps = c4["walkers"].reshape(-1, 4)                   # row 2 w + c = component c of walker w
    r = run([(p[0], p[1], p[2], c4["tbg"]) for p in comps], a.procs)
    out["c4_params"] = c4["walkers"]
    out.update({"c4_" + k: x for k, x in r.items()})

    c3 = workloads.config3(64, 3333)
    print("config 3: 16 sources x 64 walkers", flush=True)
    jobs = [(p[0], p[1], p[2], s["tbg"]) for s, W in zip(c3["sources"], c3["walkers"]) for p in W]
    r = run(jobs, a.procs)
    out["c3_params"] = c3["walkers"].reshape(-1, 4)
    out["c3_src"] = c3["src_index"]
    out["c3_tbg"] = np.array([s["tbg"] for s in c3["sources"]])
    out.update({"c3_" + k: x for k, x in r.items()})

    save_deterministic(a.out, out)
    print("wrote", a.out, os.path.getsize(a.out), "bytes")


if __name__ == "__main__":
    main()
