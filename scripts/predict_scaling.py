"""What `bench.py --gpus N` should print on real xGMI, from one-GPU measurements (no multi-GPU node was ever available to the
builder): the launch time of every rank's block of the strong-scaled headline and of the half-step schedules' per-rank launches,
measured on ONE MI355X with HIP events, plus a stated allowance for the collective.  DESIGN.md section 6 quotes the table.

    python scripts/predict_scaling.py
"""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine

T_AG = (15e-3, 40e-3)        # ms: one small all_gather_into_tensor over RCCL / xGMI inside the step (latency bound; assumed range)
T_FIX = 12e-3                # ms: propose + accept kernels of a half-step (measured: ~4-5 us each) and launch gaps

eng = Engine()
cfg = workloads.config2(1024, 1234)
eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = eng.model_flux_batch(cfg["truth"][None, :])[0]
eng.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])


def launch_ms(W, ncomp=1, reps=12):
    P = torch.from_numpy(np.ascontiguousarray(W)).cuda()
    n = len(W)
    o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
    eng.time_lnprob_torch(P, *o, reps=3)
    return eng.time_lnprob_torch(P, *o, reps=reps)


print("headline: the SAME 1024 walkers in contiguous blocks of 1024/N, one all_gather inside the step")
print("| N | slowest block's launch (ms) | blocks (ms) | predicted value (evals/s) for a %d / %d us collective |" % (T_AG[0] * 1e3, T_AG[1] * 1e3))
for N in (1, 2, 4, 8):
    per = 1024 // N
    t = [launch_ms(cfg["walkers"][b * per:(b + 1) * per]) for b in range(N)]
    lo, hi = (1024 / (max(t) + (a if N > 1 else 0.0)) * 1e3 for a in T_AG)
    print("| %d | %.3f | %s | %.0f / %.0f |" % (N, max(t), " ".join("%.3f" % x for x in t), lo, hi))

print()
print("half-steps + all-gather, per rank: launches of nq / N proposals (prior-box draws), two per step")
c5 = workloads.config2(32768, seed=5678)["walkers"]
c4cfg = workloads.config4(2048)
for name, W, nq, ncomp in (("config 5 (65536 walkers, 1 component)", c5, 32768, 1),):
    print("| %s | N | proposals per rank | launch (ms) | ms per step | walker-steps/s |" % name)
    for N in (1, 2, 4, 8):
        per = nq // N
        t = launch_ms(W[:per])
        step = 2 * (t + T_FIX + (T_AG[1] if N > 1 else 0.0))
        print("| | %d | %d | %.3f | %.3f | %.2f M |" % (N, per, t, step, 2 * nq / step / 1e3))
