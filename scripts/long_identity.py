"""Rare-event check of the dataflow schedule (stale reads across XCDs would show up as a different chain):
long runs against the half-step schedule, final state and every stored step compared bit for bit."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler
eng = Engine()
cfg = workloads.config2(1024)
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
for nw, nens, nsteps, p0 in ((1024, 1, 3000, cfg["walkers"]),
                             (256, 8, 1500, cfg["truth"] + 1e-2 * np.random.RandomState(2).randn(8, 256, 4)),
                             (4096, 1, 300, workloads.config2(4096, seed=9)["walkers"])):
    res = {}
    for sched in ("dataflow", "halfsteps"):
        d = DeviceEnsembleSampler(nw, 4, engine=eng, nens=nens, seed=31, schedule=sched,
                                  ens_src=None if nens == 1 else np.zeros(nens, dtype=np.int32))
        t0 = time.perf_counter(); d.run_mcmc(p0, nsteps); dt = time.perf_counter() - t0
        res[sched] = (d.get_chain(), d.get_log_prob(), dt)
    same = np.array_equal(res["dataflow"][0], res["halfsteps"][0]) and np.array_equal(res["dataflow"][1], res["halfsteps"][1])
    print("%d x %d walkers, %d steps (%.1e proposals): identical %s   dataflow %.2f s, half-steps %.2f s"
          % (nens, nw, nsteps, nens * nw * nsteps, same, res["dataflow"][2], res["halfsteps"][2]), flush=True)
    assert same
