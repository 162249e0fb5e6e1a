import sys, os, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import oracle as O
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.molecule import default_molfile
from test_gpu_parity import _flux_ok, _truth_source
mol = O.Molecule(default_molfile()); eng = Engine()
cfg = workloads.config2(65536, seed=5678)
src = _truth_source(eng, mol, cfg)
flux, fst, fn = eng.model_flux_batch(cfg["walkers"], return_info=True)
rflux, rfst, rn = O.model_flux_batch(mol, src, cfg["walkers"], nthreads=16)
ok, d = _flux_ok(flux, rflux, cfg["walkers"], cfg["tbg"], mol)
conv = rfst == 0
bad = np.argwhere(~ok & conv[:, None])
for w, j in bad[:10]:
    print("walker", w, "line J=", j + 1, "params", cfg["walkers"][w], "niter", fn[w], rn[w], "gpu", flux[w], "\nref", rflux[w], "\nrel", np.abs(flux[w] - rflux[w]) / np.abs(rflux[w]))
st = O.State(mol); st.backrad(cfg["tbg"]); print("backi max", st.arr("backi").max(), "floor per size:", 1e23 * st.arr("backi").max())
