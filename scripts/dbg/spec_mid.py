"""Head starts where the tasks of a half-step outnumber the wavefronts of the one-wavefront-per-SIMD build
(1024 < tasks <= 1536): off / on."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
e = Engine()
cfg = workloads.config2(2048)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = e.model_flux_batch(cfg["truth"][None, :])[0]; e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
for nw in (1536, 2048, 2560, 3072):
    c = workloads.config2(nw)
    for mode in (0, 1, 0, 1):
        e.set_sampler_speculation(mode)
        d = DeviceEnsembleSampler(nw, 4, engine=e, seed=3)
        st = d.run_mcmc(c["walkers"], 5, store=False); torch.cuda.synchronize()
        t = time.perf_counter(); d.run_mcmc(State(st.coords, st.log_prob), 30, store=False); torch.cuda.synchronize()
        print("%d prior-box walkers (%d tasks per half-step) head starts %d: %.3f ms/step" % (nw, nw // 2, mode, (time.perf_counter() - t) / 30 * 1e3), flush=True)
