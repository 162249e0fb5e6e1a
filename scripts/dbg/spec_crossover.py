"""One wavefront per SIMD with head starts against two without, by ensemble size (the rule in rx_sampler_run_async_device)."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
e = Engine()
cfg = workloads.config2(2048)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = e.model_flux_batch(cfg["truth"][None, :])[0]; e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
for nw in (3072, 4096, 6144, 8192, 16384):
    c = workloads.config2(nw)
    for occ in (1, 2, 1, 2):
        e.set_waves_per_simd(occ)
        d = DeviceEnsembleSampler(nw, 4, engine=e, seed=3)
        st = d.run_mcmc(c["walkers"], 5, store=False); torch.cuda.synchronize()
        t = time.perf_counter(); d.run_mcmc(State(st.coords, st.log_prob), 20, store=False); torch.cuda.synchronize()
        print("%d walkers, %d wavefront(s) per SIMD: %.3f ms/step" % (nw, occ, (time.perf_counter() - t) / 20 * 1e3), flush=True)
e.set_waves_per_simd(0)
