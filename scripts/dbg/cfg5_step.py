"""ms per step of the config-5 shape (65 536 walkers, dataflow sampler, one GPU) for the library in RADEX_EMCEE_AMD_LIB: as bench.py's
sharded.config5.one_gpu_dataflow (1 step, then 6 timed), repeated.  usage: python scripts/dbg/cfg5_step.py a.so b.so ..."""
import os, subprocess, sys
CHILD = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
e = Engine(); c = workloads.config2(65536, seed=5678)
e.set_source(c["tbg"], c["Jup"], np.ones(10), np.ones(10), c["bounds"])
tf = e.model_flux_batch(c["truth"][None, :])[0]; e.set_source(c["tbg"], c["Jup"], tf, 0.1 * tf, c["bounds"])
out = []
for rep in range(5):
    d = DeviceEnsembleSampler(65536, 4, engine=e, seed=2024)
    st = d.run_mcmc(c["walkers"], 1, store=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.run_mcmc(State(st.coords, st.log_prob), 6, store=False)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 6 * 1e3)
print("RESULT " + " ".join("%.3f" % x for x in out) + "  median %.3f ms per step" % sorted(out)[2])
'''
for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RADEX_EMCEE_AMD_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print("%-12s %s" % (os.path.basename(lib)[:-3], line[0][7:] if line else r.stderr[-400:]), flush=True)
