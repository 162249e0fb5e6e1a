"""Diagnostic: two processes on GPU 0, peer dataflow through IPC handles; where do final lnp and the chain disagree?"""
import os, sys, socket, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    rank, port, nw, nsteps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=2)
    from radex_emcee_amd import workloads
    from radex_emcee_amd.engine import Engine
    from radex_emcee_amd.sampler import DeviceEnsembleSampler, walker_permutation
    e = Engine(device=0)
    cfg = workloads.config2(nw, seed=77)
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    tf = e.model_flux_batch(cfg["truth"][None, :])[0]
    e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
    d = DeviceEnsembleSampler(nw, 4, engine=e, seed=5, group=dist.group.WORLD)
    d.fallback = False
    st = d.run_mcmc(cfg["walkers"], nsteps)
    cl = d.get_log_prob()
    bad = np.nonzero(st.log_prob != cl[-1])[0]
    # which rank ran the last-step task of each walker
    h = nw // 2; per = -(-h // 2)
    perm = walker_permutation(nw, 5, nsteps - 1, 0)
    owner = np.empty(nw, dtype=int); split_of = np.empty(nw, dtype=int)
    for split in range(2):
        for j in range(h):
            w = perm[split * h + j]; owner[w] = j // per; split_of[w] = split
    acc_last = cl[-1] != (cl[-2] if nsteps > 1 else d.compute_log_prob(cfg["walkers"]).cpu().numpy())
    print("rank %d schedule %s: final lnp != last chain row for %d walkers; of these last updated by rank: %s, accepted in last step: %d; "
          "total accepted in last step by rank0 %d rank1 %d" % (rank, d.last_schedule, len(bad), np.bincount(owner[bad], minlength=2).tolist(),
          int(acc_last[bad].sum()), int((acc_last & (owner == 0)).sum()), int((acc_last & (owner == 1)).sum())), flush=True)
    if len(bad):
        k = bad[:5]
        print("rank %d examples: walker %s final %s chain[-1] %s chain[-2] %s" % (rank, k.tolist(), st.log_prob[k], cl[-1][k], cl[-2][k] if nsteps > 1 else None), flush=True)
    st2 = d.run_mcmc(st, 3)
    cl2 = d.get_log_prob()
    bad2 = np.nonzero(st2.log_prob != cl2[-1])[0]
    print("rank %d second call: final lnp != last chain row for %d walkers %s" % (rank, len(bad2), bad2[:8].tolist()), flush=True)
    np.savez("/tmp/peer_dbg_%d.npz" % rank, lnp1=st.log_prob, lnp2=st2.log_prob, cl=cl2, coords2=st2.coords, acc=d.acceptance_fraction)
    np.save("/tmp/peer_dbg_%d.npy" % rank, st.log_prob)
    dist.barrier(); dist.destroy_process_group(); e.close()
    sys.exit(0)
nw, nsteps = int(sys.argv[1]), int(sys.argv[2])
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(r), str(port), str(nw), str(nsteps)], env=env) for r in range(2)]
for p in ps:
    p.wait(timeout=300)
a, b = np.load("/tmp/peer_dbg_0.npy"), np.load("/tmp/peer_dbg_1.npy")
print("ranks agree on final lnp:", np.array_equal(a, b), "differ at", int((a != b).sum()))
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler
e = Engine(device=0)
cfg = workloads.config2(nw, seed=77)
e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
tf = e.model_flux_batch(cfg["truth"][None, :])[0]
e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
ref = DeviceEnsembleSampler(nw, 4, engine=e, seed=5)
s1 = ref.run_mcmc(cfg["walkers"], nsteps)
s2 = ref.run_mcmc(s1, 3)
rcl = ref.get_log_prob()
print("ref: final lnp == last chain row:", np.array_equal(s2.log_prob, rcl[-1]), "after call 1:", np.array_equal(s1.log_prob, rcl[nsteps - 1]))
for r in range(2):
    z = np.load("/tmp/peer_dbg_%d.npz" % r)
    print("rank %d vs ref: lnp1 %s lnp2 %s chain_lnp %s coords2 %s acc %s" % (r, np.array_equal(z["lnp1"], s1.log_prob), np.array_equal(z["lnp2"], s2.log_prob),
          np.array_equal(z["cl"], rcl), np.array_equal(z["coords2"], s2.coords), np.array_equal(z["acc"], ref.acceptance_fraction)))
    bad = np.nonzero(z["lnp2"] != s2.log_prob)[0]
    print("   lnp2 differs at", bad[:10].tolist(), z["lnp2"][bad[:4]], s2.log_prob[bad[:4]], "ref chain last rows", rcl[-2:, bad[:4]])
    bad1 = np.nonzero(z["lnp1"] != s1.log_prob)[0]
    print("   lnp1 differs at", bad1[:10].tolist(), z["lnp1"][bad1[:4]], s1.log_prob[bad1[:4]])
