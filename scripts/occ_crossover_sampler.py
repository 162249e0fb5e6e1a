"""Dataflow sampler step time with one and with two wavefronts per SIMD, by ensemble size."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
e = Engine()
cfg = workloads.config2(8)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = e.model_flux_batch(cfg["truth"][None, :])[0]
e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
for n in (2048, 4096, 6144, 8192, 16384):
    p0 = cfg["truth"] + 1e-3 * np.random.RandomState(1).randn(n, 4)
    r = []
    for occ in (1, 2):
        e.set_waves_per_simd(occ)
        d = DeviceEnsembleSampler(n, 4, engine=e, seed=3)
        st = d.run_mcmc(p0, 30, store=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d.run_mcmc(State(st.coords, st.log_prob), 30, store=False)
        torch.cuda.synchronize(); r.append((time.perf_counter() - t0) / 30 * 1e3)
    e.set_waves_per_simd(0)
    print("N=%6d walkers: 1 wave/SIMD %.3f ms/step (%.2f M/s)   2 waves/SIMD %.3f ms/step (%.2f M/s)"
          % (n, r[0], n / r[0] / 1e3, r[1], n / r[1] / 1e3), flush=True)
