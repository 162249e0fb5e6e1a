"""Diagnostic: iteration counts near the truth, host-call vs kernel time, and a profile of the sampler loop."""
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
cfg = workloads.config2(1024); eng = Engine()
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = eng.model_flux_batch(cfg["truth"][None,:])[0]
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1*tf, cfg["bounds"])
rs = np.random.RandomState(99)
p0 = cfg["truth"] + 1e-3*rs.randn(512,4)
print("truth", cfg["truth"])
lnp, st, nit = eng.lnprob_batch(p0, return_info=True)
print("niter near truth: mean %.1f max %d" % (nit.mean(), nit.max()))
for n in (1, 64, 512):
    t=time.perf_counter()
    for _ in range(20): eng.lnprob_batch(p0[:n])
    print("host-buffer lnprob_batch N=%d: %.3f ms per call" % (n, (time.perf_counter()-t)/20*1e3))
P = torch.from_numpy(p0).cuda(); o=[torch.empty(512,dtype=t,device='cuda') for t in (torch.float64,torch.int32,torch.int32)]
print("kernel ms (512 near truth):", eng.time_lnprob_torch(P,*o,reps=10))
from radex_emcee_amd.sampler import EnsembleSampler
p0 = cfg["truth"] + 1e-3*rs.randn(1024,4)
smp = EnsembleSampler(1024, 4, eng.lnprob_batch, vectorize=True, seed=7)
state = smp.run_mcmc(p0, 5, progress=False)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); smp.run_mcmc(state, 20, progress=False); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
lnp, st, nit = eng.lnprob_batch(state.coords, return_info=True)
print("niter after 25 steps: mean %.1f max %d" % (nit.mean(), nit.max()))
