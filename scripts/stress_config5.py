"""BASELINE configs[4] at full length on ONE GPU: 65536 walkers x 10000 steps (6.5536e8 lnlike evaluations),
dataflow sampler, chunks of 250 steps, nothing stored.  Prints the sustained rate.
usage: python scripts/stress_config5.py [nsteps=10000] [nwalkers=65536]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cfg = workloads.config2(nw, seed=5678)
eng = Engine()
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
d = DeviceEnsembleSampler(nw, 4, engine=eng, seed=2024)
st = d.run_mcmc(cfg["walkers"], 1, store=False)
torch.cuda.synchronize()
t0 = time.perf_counter(); done = 1
while done < nsteps:
    n = min(250, nsteps - done)
    st = d.run_mcmc(State(st.coords, st.log_prob), n, store=False)
    done += n
    dt = time.perf_counter() - t0
    print("steps %6d  %.1f s  sustained %.3f M walker-steps/s  acceptance %.3f  median lnp %.3f"
          % (done, dt, nw * (done - 1) / dt / 1e6, float(d.acceptance_fraction.mean()), float(np.median(st.log_prob))), flush=True)
print("TOTAL %d evaluations in %.1f s = %.3f M evals/s" % (nw * (done - 1), dt, nw * (done - 1) / dt / 1e6))
