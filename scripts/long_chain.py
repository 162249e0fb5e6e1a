"""Stability: long dataflow chains (no abort, finite state, steady rate).  usage: python scripts/long_chain.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
eng = Engine()
cfg = workloads.config2(1024)
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
d = DeviceEnsembleSampler(1024, 4, engine=eng, seed=1)
st = d.run_mcmc(cfg["truth"] + 1e-3 * np.random.RandomState(0).randn(1024, 4), 1, store=False)
t0 = time.perf_counter(); n = 0
for chunk in range(20):
    st = d.run_mcmc(State(st.coords, st.log_prob), 1000, store=False); n += 1000
    print("1024 walkers: %6d steps  %.2f M walker-steps/s  acceptance %.3f  lnp median %.3f min %.3f  finite %s"
          % (n, 1024 * n / (time.perf_counter() - t0) / 1e6, float(d.acceptance_fraction.mean()),
             float(np.median(st.log_prob)), float(st.log_prob.min()), bool(np.all(np.isfinite(st.coords)))), flush=True)
c3 = workloads.config3(1024, init="ball")
e3 = Engine()
for s in c3["sources"]:
    e3.set_source(s["tbg"], s["Jup"], s["flux"], s["eflux"], s["bounds"], src=s["slot"])
d3 = DeviceEnsembleSampler(1024, 4, engine=e3, nens=16, ens_src=np.arange(16), seed=2)
st = d3.run_mcmc(c3["walkers"], 1, store=False)
t0 = time.perf_counter(); n = 0
for chunk in range(4):
    st = d3.run_mcmc(State(st.coords, st.log_prob), 250, store=False); n += 250
    print("16 x 1024 walkers: %5d steps  %.2f M walker-steps/s  acceptance per source %.2f..%.2f  finite %s"
          % (n, 16384 * n / (time.perf_counter() - t0) / 1e6, float(d3.acceptance_fraction.mean(1).min()),
             float(d3.acceptance_fraction.mean(1).max()), bool(np.all(np.isfinite(st.log_prob)))), flush=True)
