"""Kernel time and result hashes of alternative builds on the bench workloads: 1024 config-2 walkers (latency
regime, one wavefront per SIMD), 32768 (throughput regime, two per SIMD), 4096 two-component walkers, the 16
sources of config 3, and a 1024-walker dataflow sampler run.  Equal hashes = bit-identical results.
Usage: python scripts/ablate.py lib.so ..."""
import os, subprocess, sys
CHILD = r'''
import sys, hashlib, time, numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
def H(*ts):
    h = hashlib.sha1()
    for t in ts: h.update(np.ascontiguousarray(t.cpu().numpy() if hasattr(t, "cpu") else t).tobytes())
    return h.hexdigest()[:10]
e = Engine()
out = []
for n, seed in ((1024, 1234), (32768, 5678)):
    cfg = workloads.config2(n, seed=seed)
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    P = torch.from_numpy(cfg["walkers"]).cuda()
    o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
    e.time_lnprob_torch(P, *o, reps=2)
    ms = e.time_lnprob_torch(P, *o, reps=10)
    out.append("%d: %.4f ms (%.0f k/s) %s" % (n, ms, n / ms, H(*o)))
c4 = workloads.config4(4096); W = c4["walkers"].copy(); W[2048:] = workloads.draw_prior_2comp(c4["bounds"], 2048, 91)
e.set_source(c4["tbg"], c4["Jup"], np.ones(10), 0.1 * np.ones(10), c4["bounds"], 2, 40.0)
P = torch.from_numpy(W).cuda(); o = [torch.empty(4096, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
ms = e.time_lnprob_torch(P, *o, reps=5); out.append("2comp4096: %.3f ms %s" % (ms, H(*o)))
c3 = workloads.config3(512)
for s in c3["sources"]: e.set_source(s["tbg"], s["Jup"], s["flux"], s["eflux"], s["bounds"], src=s["slot"])
P = torch.from_numpy(c3["walkers"].reshape(-1, 4)).cuda(); idx = torch.from_numpy(c3["src_index"]).cuda()
o = [torch.empty(8192, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
ms = e.time_lnprob_torch(P, *o, reps=5, src_index=idx); out.append("cfg3: %.3f ms %s" % (ms, H(*o)))
cfg = workloads.config2(1024); e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = e.model_flux_batch(cfg["truth"][None, :])[0]; e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
d = DeviceEnsembleSampler(1024, 4, engine=e, seed=7)
st = d.run_mcmc(cfg["truth"] + 1e-3 * np.random.RandomState(99).randn(1024, 4), 20, store=False)
torch.cuda.synchronize(); t0 = time.perf_counter(); st = d.run_mcmc(State(st.coords, st.log_prob), 100, store=False); torch.cuda.synchronize()
out.append("sampler1024: %.4f ms/step %s" % ((time.perf_counter() - t0) * 10, H(st.coords, st.log_prob)))
print("RESULT " + " | ".join(out))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, RADEX_EMCEE_AMD_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print("%-14s %s" % (os.path.basename(lib)[:-3], line[0][7:] if line else r.stderr[-400:]))
