import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
e = Engine()
cfg = workloads.config2(2048)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = e.model_flux_batch(cfg["truth"][None, :])[0]; e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
for nw in (8192, 65536):
    c = workloads.config2(nw, seed=5678)
    d = DeviceEnsembleSampler(nw, 4, engine=e, seed=2024)
    st = d.run_mcmc(c["walkers"], 2, store=False); torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t = time.perf_counter(); st = d.run_mcmc(State(st.coords, st.log_prob), 8, store=False); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) / 8 * 1e3)
    print("%s: %d walkers %s ms/step" % (os.environ.get("TAG"), nw, " ".join("%.3f" % x for x in ts)), flush=True)
