import sys, json, time
sys.path.insert(0, '.')
import numpy as np
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
from oracle import oracle as O
cfg = workloads.config2(1024)
e = Engine()
e.set_source(cfg['tbg'], cfg['Jup'], np.ones(10), np.ones(10), cfg['bounds'])
flux, st, nit = e.model_flux_batch(cfg['walkers'], return_info=True)
t=time.time(); flux, st, nit = e.model_flux_batch(cfg['walkers'], return_info=True); dt=time.time()-t
print("gpu flux batch 1024 (2nd): %.3f ms -> %.3g evals/s" % (dt*1e3, 1024/dt))
mol = O.Molecule(e.molfile)
src = O.Source(cfg['tbg'], cfg['Jup'], np.ones(10), np.ones(10), cfg['bounds'])
rf, rst, rnit = O.model_flux_batch(mol, src, cfg['walkers'], nthreads=8)
print("status equal", (st==rst).mean(), "niter equal", (nit==rnit).mean())
# background flux scale: max backi * size * 1e23
st0 = O.State(mol); st0.backrad(cfg['tbg']); bmax = st0.arr('backi').max()
size = 10**cfg['walkers'][:,3:4]
bscale = bmax*size*1e23
smax = np.max(np.abs(rf), axis=1, keepdims=True)
d = np.abs(flux-rf)
for floor_rel in (1e-6, 1e-8, 1e-9, 1e-10, 1e-12):
    for bf in (0.0, 1e-10, 1e-12):
        tol = 1e-4*np.abs(rf) + floor_rel*smax + bf*bscale
        print("floor %.0e bgfloor %.0e : violations %d" % (floor_rel, bf, (d>tol).sum()))
sig = np.abs(rf) > 1e-6*smax
rel = d[sig]/np.abs(rf[sig])
print("lines above 1e-6*max: n=%d, rel dev median %.2e p99 %.2e max %.2e" % (sig.sum(), np.median(rel), np.percentile(rel,99), rel.max()))
w,j = np.unravel_index(np.argmax(np.where(sig, d/np.maximum(np.abs(rf),1e-300), 0)), d.shape)
print("worst significant:", w, j, cfg['walkers'][w], flux[w], rf[w], nit[w])
