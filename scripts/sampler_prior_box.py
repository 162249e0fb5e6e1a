"""The workload behind bench.py's `sampler_config2_prior_box`, alone (for rocprofv3 passes over rx_sampler_kernel):
the 1024 prior-box walkers of BASELINE configs[1] as ONE ensemble under the dataflow schedule -- 20 steps of burn-in
(one launch), then 120 steps (one launch).  Prints the timing and the kernel's own counters.
usage: python3 scripts/sampler_prior_box.py [nwalkers] [nsteps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 120
cfg = workloads.config2(nw)
eng = Engine()
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
if os.environ.get("RX_SPECULATE"): eng.set_sampler_speculation(int(os.environ["RX_SPECULATE"]))
d = DeviceEnsembleSampler(nw, 4, engine=eng, seed=7)
s = d.run_mcmc(cfg["walkers"], 20, store=False)
torch.cuda.synchronize()
eng.sampler_stats(True)
t = time.perf_counter()
d.run_mcmc(State(s.coords, s.log_prob), nst, store=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t
st = eng.sampler_stats(False)
print(json.dumps({"walkers": nw, "steps": nst, "ms_per_step": dt / nst * 1e3, "walker_steps_per_s": nw * nst / dt,
                  "niter_mean": st["niter_sum"] / max(1, st["solved"]), "maxiter_fraction": st["maxiter_solves"] / max(1, st["solved"]),
                  "outside_prior": 1 - st["solved"] / max(1, st["tasks"]), "mean_task_us": st["busy_ticks"] / max(1, st["tasks"]) / 100.0,
                  "mean_wait_us": st["wait_ticks"] / max(1, st["tasks"]) / 100.0,
                  "head_starts": st["head_starts"] / max(1, st["tasks"]), "evaluated_twice": st["evaluated_twice"] / max(1, st["tasks"])}))
