"""Golden pivot sequences from the reference's own sgefa_ (emcee/pyradex/radex/radex.so, called
through oracle/macho_ref.py; needs /root/reference, runs only in the build container).

Writes tests/golden/ref_sgefa.json: matrices (row-major) with the LAST ROW ALREADY SET TO ONES (what
lubksb_ hands to sgeir_/sgefa_), the binary's ipvt (converted to 0-based) and info.  The cases
target isamax's first-maximum rule: exact ties at step 0, an exact tie at step 1 that the
interchange of step 0 turns into a position-versus-row-index question, small-integer matrices,
and plain gaussian ones.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import macho_ref as M      # noqa: E402


def main():
    R = M.RefRadex()
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int)
    sgefa = R._fn("_sgefa_", None, pd, pi, pi, pi, pi)
    rng = np.random.RandomState(424242)
    cases = []

    def add(kind, a):
        n = a.shape[0]
        a = a.copy()
        a[n - 1] = 1.0
        A = np.asfortranarray(a).copy(order="F")
        ipvt = np.zeros(n, dtype=np.int32)
        info, lda, nn = C.c_int(0), C.c_int(n), C.c_int(n)
        sgefa(A.ctypes.data_as(pd), C.byref(lda), C.byref(nn), ipvt.ctypes.data_as(pi), C.byref(info))
        cases.append(dict(kind=kind, n=n, A=[float(v) for v in a.reshape(-1)],
                          ipvt=[int(v) - 1 for v in ipvt], info=int(info.value)))

    for n in (8, 20, 41):
        for _ in range(2):
            add("gauss", rng.randn(n, n))
        for _ in range(3):
            a = rng.randn(n, n) * 0.3
            a[rng.choice(n - 1, 3, replace=False), 0] = 5.0 * rng.choice([-1.0, 1.0], 3)
            add("tie0", a)
        for _ in range(3):
            a = rng.randn(n, n) * 0.3
            r0 = int(rng.randint(3, n - 1))
            a[r0, 0] = 9.0
            a[0, :2] = a[2, :2] = (0.3, 7.0)
            add("tie1", a)
        for _ in range(2):
            add("int", rng.randint(-2, 3, size=(n, n)).astype(float))
    assert not R.trap_log, R.trap_log
    json.dump(dict(source="radex.so:_sgefa_ (last row = 1.0, as lubksb_ passes it)", layout="A row-major; ipvt 0-based",
                   cases=cases), open(os.path.join(HERE, "ref_sgefa.json"), "w"), indent=0)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
