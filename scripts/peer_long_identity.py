"""Long runs of the peer-write dataflow schedule (one process per rank, replicas mapped through IPC handles) against the
one-GPU dataflow kernel, compared bit for bit.  On a box with ONE GPU the ranks share it (the rehearsal every record in
profiles/ comes from); on a node with more, rank r takes GPU r -- THE run that is still missing: stores, atomics and
release/acquire across xGMI (`python3 scripts/peer_long_identity.py 1200 400 <nranks>`; equal hashes = verified). 1024 config-2 prior-box walkers over
`nsteps` steps in calls of `chunk` (the ring of versions and its per-step counters turn over 100 times per call),
then 3 ensembles x 256 walkers with per-ensemble sources.
usage: python3 scripts/peer_long_identity.py [nsteps=1200] [chunk=400] [nranks=2]"""
import hashlib, os, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def setup(e, shape, nw):
    from radex_emcee_amd import workloads
    if shape == "config2":
        cfg = workloads.config2(nw, seed=31)
        e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
        tf = e.model_flux_batch(cfg["truth"][None, :])[0]
        e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
        return cfg["walkers"], 1, None
    c3 = workloads.config3(nw, init="prior")
    for k in range(3):
        s3 = c3["sources"][k]
        e.set_source(s3["tbg"], s3["Jup"], s3["flux"], s3["eflux"], s3["bounds"], src=k)
    return c3["walkers"][:3], 3, np.arange(3)

def run(d, p0, nsteps, chunk):
    h = hashlib.sha1()
    st, done, t0 = p0, 0, time.perf_counter()
    while done < nsteps:
        n = min(chunk, nsteps - done)
        st = d.run_mcmc(st, n)
        done += n
    for a in (d.get_chain(), d.get_log_prob(), st.coords, st.log_prob, d.acceptance_fraction):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest(), time.perf_counter() - t0

if len(sys.argv) > 1 and sys.argv[1] == "worker":
    rank, port, nsteps, chunk, world = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from radex_emcee_amd.engine import Engine
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    ndev = torch.cuda.device_count()
    dev = rank if ndev >= world else 0                    # one GPU per rank where there are enough, else all on GPU 0
    torch.cuda.set_device(dev)
    e = Engine(device=dev)
    for shape, nw in (("config2", 1024), ("config3", 256)):
        p0, nens, ens_src = setup(e, shape, nw)
        d = DeviceEnsembleSampler(nw, 4, engine=e, seed=17, group=dist.group.WORLD, nens=nens, ens_src=ens_src, schedule="dataflow")
        d.fallback = False
        hx, dt = run(d, p0, nsteps, chunk)
        print("rank %d (GPU %d) %s: %s  %s  %.2f s (%.0f walker-steps/s)  verified against half-steps: %s"
              % (rank, dev, shape, d.last_schedule, hx, dt, nens * nw * nsteps / dt, d.peer_verified), flush=True)
    dist.barrier(); dist.destroy_process_group(); e.close()
    sys.exit(0)

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 400
world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(port), str(nsteps), str(chunk), str(world)], env=env)
      for r in range(world)]
rc = [p.wait(timeout=1100) for p in ps]
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler
e = Engine(device=0)
for shape, nw in (("config2", 1024), ("config3", 256)):
    p0, nens, ens_src = setup(e, shape, nw)
    d = DeviceEnsembleSampler(nw, 4, engine=e, seed=17, nens=nens, ens_src=ens_src)
    hx, dt = run(d, p0, nsteps, chunk)
    print("one GPU %s: %s  %s  %.2f s (%.0f walker-steps/s)" % (shape, d.last_schedule, hx, dt, nens * nw * nsteps / dt), flush=True)
print("worker exit codes", rc, "-- equal hashes = the same chain, positions, log-probabilities and acceptance counts, bit for bit")
