"""GPU: what an iteration costs a lone wavefront with the refinement on / off -- launches of walkers that never converge
(one per SIMD at most), limited to different iteration counts; slopes in microseconds per iteration."""
import sys, time; sys.path.insert(0, ".")
import numpy as np
from oracle import oracle as O
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine

cfg = workloads.config2(4096, seed=1234)
eng = Engine(); mol = O.Molecule(eng.molfile)
tf = np.ones(10)
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
lnp, st, nit = eng.lnprob_batch(cfg["walkers"], return_info=True)
slow = cfg["walkers"][st == 1]
print("never-converging walkers:", len(slow))
W = np.tile(slow, (max(1, 512 // len(slow)), 1))[:512]


def t_launch(maxiter, on):
    eng.set_iteration_limits(10, maxiter)
    eng.set_refinement(on)
    eng.lnprob_batch(W)
    ts = []
    for _ in range(15):
        t = time.perf_counter(); eng.lnprob_batch(W); ts.append(time.perf_counter() - t)
    return 1e6 * np.median(ts)


for on in (False, True):
    t = {m: t_launch(m, on) for m in (10, 12, 20, 40, 100, 200)}
    eng.set_iteration_limits(10, 200); eng.set_refinement_counting(True); eng.refinement_counters(reset=True); eng.lnprob_batch(W); c = eng.refinement_counters(); eng.set_refinement_counting(False)
    print("refinement %-3s: launch (us) %s" % ("on" if on else "off", {k: round(v, 1) for k, v in t.items()}))
    print("   per iteration: 10-12 %.2f us | 12-20 %.2f | 20-40 %.2f | 40-100 %.2f | 100-200 %.2f ; counters per walker %s" % (
        (t[12] - t[10]) / 2, (t[20] - t[12]) / 8, (t[40] - t[20]) / 20, (t[100] - t[40]) / 60, (t[200] - t[100]) / 100,
        {k: round(v / len(W), 1) for k, v in c.items()}))
