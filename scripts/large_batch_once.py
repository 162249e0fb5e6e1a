"""One warm-up and three timed launches over 32768 config-2 walkers (2 waves per SIMD): for rocprofv3 --pmc."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from radex_emcee_amd.engine import Engine
from radex_emcee_amd import workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cfg = workloads.config2(n, seed=5678)
e = Engine()
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
P = torch.from_numpy(cfg["walkers"]).cuda()
o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
for _ in range(4):
    e.lnprob_batch_torch(P, *o)
torch.cuda.synchronize()
print("niter sum", int(o[2].sum()))
