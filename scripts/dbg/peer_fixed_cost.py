"""Fixed and per-step cost of a peer-write run_mcmc call (ranks sharing GPU 0, gloo): config-5 shape, calls of 3 / 6 / 12 / 24 steps.
usage: python scripts/dbg/peer_fixed_cost.py [nranks=2] [nwalkers=65536]"""
import os, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

def run(d, p0, label):
    from radex_emcee_amd.sampler import State
    import torch
    st = d.run_mcmc(p0, 1, store=False)
    out = []
    for nst in (3, 6, 12, 24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = d.run_mcmc(State(st.coords, st.log_prob), nst, store=False)
        torch.cuda.synchronize(); out.append((nst, (time.perf_counter() - t0) * 1e3))
    (n1, t1), (n2, t2) = out[1], out[3]
    slope = (t2 - t1) / (n2 - n1)
    print("%s: %s  -> %.2f ms per step + %.1f ms per call (%s)" % (label, "  ".join("%d steps %.1f ms" % o for o in out), slope, t1 - n1 * slope, d.last_schedule), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "worker":
    rank, port, world, nw = (int(x) for x in sys.argv[2:6])
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from radex_emcee_amd.engine import Engine
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    from radex_emcee_amd import workloads
    torch.cuda.set_device(0)
    e = Engine(device=0)
    cfg = workloads.config2(nw, seed=5678)
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    tf = e.model_flux_batch(cfg["truth"][None, :])[0]; e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
    d = DeviceEnsembleSampler(nw, 4, engine=e, seed=2024, group=dist.group.WORLD, schedule="dataflow")
    run(d, cfg["walkers"], "rank %d of %d, peer-write" % (rank, world))
    dist.barrier(); dist.destroy_process_group(); e.close(); sys.exit(0)

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(port), str(world), str(nw)], env=env) for r in range(world)]
rc = [p.wait(timeout=600) for p in ps]
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler
from radex_emcee_amd import workloads
e = Engine(device=0)
cfg = workloads.config2(nw, seed=5678)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = e.model_flux_batch(cfg["truth"][None, :])[0]; e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
d = DeviceEnsembleSampler(nw, 4, engine=e, seed=2024)
run(d, cfg["walkers"], "one GPU, dataflow")
print("worker exit codes", rc)
