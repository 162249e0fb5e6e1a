"""Head starts where two wavefronts share a SIMD: the 16-source sampler of config 3 and the 65536-walker
ensemble of config 5, speculation off / on."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
e = Engine()
c3 = workloads.config3(1024, init="ball")
for s in c3["sources"]: e.set_source(s["tbg"], s["Jup"], s["flux"], s["eflux"], s["bounds"], src=s["slot"])
for mode in (0, 1, 0, 1):
    e.set_sampler_speculation(mode)
    d = DeviceEnsembleSampler(1024, 4, engine=e, nens=16, ens_src=np.arange(16), seed=11)
    st = d.run_mcmc(c3["walkers"], 2, store=False); torch.cuda.synchronize()
    t = time.perf_counter(); d.run_mcmc(State(st.coords, st.log_prob), 10, store=False); torch.cuda.synchronize()
    print("config3 16x1024 speculation %d: %.3f ms/step" % (mode, (time.perf_counter() - t) * 100), flush=True)
cfg = workloads.config2(2048)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = e.model_flux_batch(cfg["truth"][None, :])[0]; e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
for nw in (2048, 4096, 65536):
    c = workloads.config2(nw)
    for mode in (0, 1):
        e.set_sampler_speculation(mode)
        d = DeviceEnsembleSampler(nw, 4, engine=e, seed=3)
        st = d.run_mcmc(c["walkers"], 3, store=False); torch.cuda.synchronize()
        t = time.perf_counter(); d.run_mcmc(State(st.coords, st.log_prob), 10, store=False); torch.cuda.synchronize()
        print("%d prior-box walkers speculation %d: %.3f ms/step" % (nw, mode, (time.perf_counter() - t) * 100), flush=True)
