"""GPU: the 1024-walker headline launch with the refinement switched off at run time (every solve the pivoted elimination): what
the refinement code costs the pivoted path of the same library.  RADEX_EMCEE_AMD_LIB selects the build (scripts/mk.sh)."""
import sys, os; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
eng = Engine(); eng.set_refinement(False)
cfg = workloads.config2(1024, 1234)
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
P = torch.from_numpy(cfg["walkers"]).cuda()
o = [torch.empty(1024, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
eng.time_lnprob_torch(P, *o, reps=5)
print(os.environ.get("RADEX_EMCEE_AMD_LIB"), "refinement off: %.4f ms" % np.median([eng.time_lnprob_torch(P, *o, reps=1) for _ in range(30)]))
