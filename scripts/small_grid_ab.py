"""(Round 5, kept for the record: the rule it measured was removed afterwards -- no effect.)  A/B of launch()'s small-batch rule on the SAME binary: RX_SMALL_GRID=1 (one workgroup per walker up to the number of compute
units, so that the kernel deals one walker per compute unit first) against RX_SMALL_GRID=0 (ceil(items / 4) workgroups, four walkers
per workgroup).  Kernel time (HIP events, median of 5 x 12 launches) for 128 / 256 / 512 walkers: a rank's block of a strong-scaled
1024-walker ensemble.   python scripts/small_grid_ab.py"""
import os, subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, ".")
    import numpy as np, torch
    from radex_emcee_amd import workloads
    from radex_emcee_amd.engine import Engine
    eng = Engine()
    cfg = workloads.config2(1024, 1234)
    eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    out = []
    for n in (128, 256, 512):
        P = torch.from_numpy(np.ascontiguousarray(cfg["walkers"][:n])).cuda()
        o = [torch.empty(n, dtype=t, device="cuda") for t in (torch.float64, torch.int32, torch.int32)]
        eng.time_lnprob_torch(P, *o, reps=3)
        out.append(float(np.median([eng.time_lnprob_torch(P, *o, reps=12) for _ in range(5)])))
    print(" ".join("%.4f" % x for x in out))
    sys.exit(0)
for rep in range(2):
    for sg in ("1", "0"):
        r = subprocess.run([sys.executable, __file__, "worker"], env=dict(os.environ, RX_SMALL_GRID=sg), capture_output=True, text=True)
        print("RX_SMALL_GRID=%s: 128 / 256 / 512 walkers %s ms" % (sg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
