"""GPU: throughput regime -- one wavefront per SIMD WITH the refinement against two wavefronts per SIMD without it (the
two-wavefront build has no room for the kept inverses).  32768 walkers per launch; the 65536-walker ensemble in the sampler."""
import sys, time; sys.path.insert(0, ".")
import numpy as np
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler

eng = Engine()
cfg = workloads.config2(32768, seed=5678)
tf = np.ones(10)
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
for occ, on in ((2, False), (2, True), (1, False), (1, True)):
    eng.set_waves_per_simd(occ); eng.set_refinement(on)
    eng.lnprob_batch(cfg["walkers"])
    ts = []
    for _ in range(7):
        t = time.perf_counter(); eng.lnprob_batch(cfg["walkers"]); ts.append(time.perf_counter() - t)
    eng.set_refinement_counting(True); eng.refinement_counters(reset=True); lnp, st, nit = eng.lnprob_batch(cfg["walkers"], return_info=True); c = eng.refinement_counters(); eng.set_refinement_counting(False)
    import hashlib
    print("32768 walkers, %d wavefront(s) per SIMD, refinement %-3s: %.2f ms per launch (host-timed median); refined %.1f %% of %d iterations; sha1 of lnprob %s"
          % (occ, "on" if on else "off", 1e3 * np.median(ts), 100.0 * c["refined"] / max(c["iterations"], 1), c["iterations"], hashlib.sha1(lnp.tobytes()).hexdigest()[:12]))
c5 = workloads.config2(65536, seed=5678)
for occ, on in ((2, False), (2, True), (1, True)):
    eng.set_waves_per_simd(occ); eng.set_refinement(on)
    d = DeviceEnsembleSampler(65536, 4, engine=eng, seed=1)
    st = d.run_mcmc(c5["walkers"], 4, store=False)
    t = time.perf_counter(); st = d.run_mcmc(st, 6, store=False); dt = time.perf_counter() - t
    print("65536-walker ensemble, dataflow sampler, %d wavefront(s) per SIMD, refinement %-3s: %.2f ms per step" % (occ, "on" if on else "off", 1e3 * dt / 6))
eng.set_waves_per_simd(0)
