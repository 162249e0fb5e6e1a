"""The dataflow sampler across ranks (rx_sampler_peer_*: replicas of the sampler's shared state written by
peers): the multi-GPU form of the reference's Pool.map over walkers [/root/reference/emcee/emcee_radex.py:
480-488], rehearsed on ONE GPU -- (i) two handles in one process, their replicas exchanged as device
pointers, two kernels on two streams; (ii) two processes that share GPU 0 and map each other's replica
through hipIpcGetMemHandle / hipIpcOpenMemHandle, exactly the code path of one process per GPU.  The bar is
the one the schedule promises: the chain of the one-GPU dataflow run, bit for bit (positions,
log-probabilities, acceptance counts, stored chain)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O                      # noqa: E402  (checker only)
from radex_emcee_amd import workloads               # noqa: E402
from radex_emcee_amd.engine import Engine           # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mol(co_path):
    return O.Molecule(co_path)


def _setup(e, mol, shape, nw):
    """source slot 0 of engine e for the config-2 (1 component) or config-4 (2 components) shape; start positions"""
    if shape == "config2":
        cfg = workloads.config2(nw, seed=77)
        e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
        tf = e.model_flux_batch(cfg["truth"][None, :])[0]               # (the workers below make the same SLED)
        e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
        return cfg["walkers"], 1
    cfg = workloads.config4(nw)
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"], 2, cfg["T_d"])
    tf = e.model_flux_batch(cfg["truth"][None, :])[0]
    e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"], 2, cfg["T_d"])
    return cfg["walkers"], 2


INPROC = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine, EngineError
from radex_emcee_amd.sampler import DeviceEnsembleSampler
dev = torch.device("cuda", 0)
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

def setup(e, shape, nw):
    if shape == "config2":
        cfg = workloads.config2(nw, seed=77); nc = 1
        e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
        tf = e.model_flux_batch(cfg["truth"][None, :])[0]
        e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
    else:
        cfg = workloads.config4(nw); nc = 2
        e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"], 2, cfg["T_d"])
        tf = e.model_flux_batch(cfg["truth"][None, :])[0]
        e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"], 2, cfg["T_d"])
    return cfg["walkers"], nc

for shape, nw, nsteps in (("config2", 1024, 14), ("config4", 256, 5)):
    engs = [Engine(), Engine()]
    for e in engs:
        p0, ncomp = setup(e, shape, nw)
        e.set_sampler_grid_limit(ncu // 2)
    ndim = 4 * ncomp
    ref = DeviceEnsembleSampler(nw, ndim, engine=engs[0], seed=5)            # one GPU, one kernel: the reference chain
    st_ref = ref.run_mcmc(p0, nsteps)
    assert ref.last_schedule == "dataflow"
    lnp0 = ref.compute_log_prob(p0)
    for r, e in enumerate(engs):
        e.sampler_peer_setup(2, r, 1, nw, ncomp, export=False)
    bases = [e.sampler_peer_base() for e in engs]
    assert all(bases)
    state = []
    import re
    ids = [e.bus_id() for e in engs]                      # (PCI bus id: the GPU's name in every process of the node)
    assert ids[0] == ids[1] and re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F]", ids[0]), ids
    for r, e in enumerate(engs):
        # the first shape by bus ids (both replicas on this GPU), the second by what the pointer attributes say
        e.sampler_peer_connect(bases=bases, bus_ids=ids if shape == "config2" else None)
        assert e.sampler_peer_same_device() == 2
        state.append((torch.from_numpy(np.ascontiguousarray(p0)).to(dev), lnp0.clone(), torch.zeros(nw, dtype=torch.int32, device=dev),
                      torch.zeros(nsteps, nw, ndim, dtype=torch.float64, device=dev),
                      torch.zeros(nsteps, nw, dtype=torch.float64, device=dev)))
    torch.cuda.synchronize()
    for r, e in enumerate(engs):
        e.sampler_peer_begin(*state[r][:3], stream=streams[r].cuda_stream)
    # (both replicas are seeded: the barrier of the multi-process form)
    for r, e in enumerate(engs):
        e.sampler_peer_run(2.0, 5, 0, nsteps, dev, state[r][3], state[r][4], stream=streams[r].cuda_stream)
    for r, e in enumerate(engs):
        e.sampler_wait(dev, stream=streams[r].cuda_stream)
    for r, e in enumerate(engs):
        e.sampler_peer_finish(*state[r][:3], stream=streams[r].cuda_stream)
    torch.cuda.synchronize()
    for r in range(2):
        assert np.array_equal(state[r][0].cpu().numpy(), st_ref.coords), "rank %d: final positions differ" % r
        assert np.array_equal(state[r][1].cpu().numpy(), st_ref.log_prob)
        assert np.array_equal(state[r][2].cpu().numpy() / float(nsteps), ref.acceptance_fraction)
    chain = (state[0][3] + state[1][3]).cpu().numpy()
    chain_lnp = (state[0][4] + state[1][4]).cpu().numpy()
    assert np.array_equal(chain, ref.get_chain()) and np.array_equal(chain_lnp, ref.get_log_prob())
    # every row of the chain was written by exactly one rank
    w0, w1 = (state[0][4] != 0).cpu().numpy(), (state[1][4] != 0).cpu().numpy()
    assert not (w0 & w1).any() and (w0 | w1).all() and w0.sum() == w1.sum()
    # teardown in the order the multi-process form needs: every rank unmaps its peers' blocks, THEN the blocks are freed
    # (and a handle can be set up again afterwards)
    for e in engs:
        e.sampler_peer_disconnect()
    try:
        engs[0].sampler_peer_begin(*state[0][:3], stream=streams[0].cuda_stream)
        raise AssertionError("begin after disconnect must be refused")
    except EngineError:
        pass
    for e in engs:
        e.sampler_peer_close()
    engs[0].sampler_peer_setup(2, 0, 1, nw, ncomp, export=False)
    assert engs[0].sampler_peer_base()
    for e in engs:
        e.sampler_peer_close()
        e.close()
    print("OK", shape, nw, nsteps, flush=True)
'''


def test_peer_dataflow_two_handles_one_process(tmp_path):
    """Two ranks = two handles on GPU 0, each running its block of every half-step in its own persistent
    kernel (half the CUs each, two streams) and publishing into both replicas; config-2 and config-4 shapes.
    More steps than the ring of versions holds (config 2: 14 > 12), so the per-step counters are exercised
    across ranks too.  In its own process with GPU_MAX_HW_QUEUES=8: two kernels that wait for each other must
    sit in different hardware queues, and HIP deals a process's streams onto 4 queues by default (a pair of
    streams that shares one never runs concurrently -- an artefact of ranks inside ONE process; the
    one-process-per-rank form below needs nothing of the kind)."""
    script = tmp_path / "inproc.py"
    script.write_text(INPROC)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=420, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("OK ") == 2


def test_peer_dataflow_one_rank_group(co_path, mol):
    """A group of ONE rank runs the same kernel in its one-GPU form on the replica block."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    e = Engine(co_path)
    p0, _ = _setup(e, mol, "config2", 256)
    ref = DeviceEnsembleSampler(256, 4, engine=e, seed=9)
    st_ref = ref.run_mcmc(p0, 6)
    import torch.distributed as dist
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        da = DeviceEnsembleSampler(256, 4, engine=e, seed=9, group=dist.group.WORLD)
        assert da.schedule == "auto"                                # the default with a group
        st_a = da.run_mcmc(p0, 6)                                   # 128 proposals per half-step: rank 0 alone, by rule
        assert da.last_schedule == "rank0" and da.schedule_choice == "rank0" and "by rule" in da.schedule_reason
        assert np.array_equal(st_a.coords, st_ref.coords) and np.array_equal(da.get_chain(), ref.get_chain())
        d = DeviceEnsembleSampler(256, 4, engine=e, seed=9, group=dist.group.WORLD, schedule="dataflow")
        st = d.run_mcmc(p0, 6)
        assert d.last_schedule == "dataflow-peer" and d.peer_state is True
        assert np.array_equal(st.coords, st_ref.coords) and np.array_equal(d.get_chain(), ref.get_chain())
        assert np.array_equal(d.get_log_prob(), ref.get_log_prob())
        st2 = d.run_mcmc(st, 3)                                     # a second call re-seeds the replica
        st2_ref = ref.run_mcmc(st_ref, 3)
        assert np.array_equal(st2.coords, st2_ref.coords)
    finally:
        dist.destroy_process_group()
    e.close()


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, port, out, shape, nw, nsteps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6], int(sys.argv[7]), int(sys.argv[8])
import json
import torch
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine
from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
e = Engine(device=0)                                   # every rank on GPU 0: the replicas travel as IPC handles
nens, ens_src = 1, None
if shape == "config3":                                 # three ensembles advanced together, one source slot each
    c3 = workloads.config3(nw, init="ball"); ncomp = 1; nens = 3; ens_src = np.arange(3)
    for k in range(3):
        s3 = c3["sources"][k]
        e.set_source(s3["tbg"], s3["Jup"], s3["flux"], s3["eflux"], s3["bounds"], src=k)
    cfg = dict(walkers=c3["walkers"][:3])
elif shape == "config2":
    cfg = workloads.config2(nw, seed=77); ncomp = 1
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    tf = e.model_flux_batch(cfg["truth"][None, :])[0]
    e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"])
else:
    cfg = workloads.config4(nw); ncomp = 2
    e.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"], 2, cfg["T_d"])
    tf = e.model_flux_batch(cfg["truth"][None, :])[0]
    e.set_source(cfg["tbg"], cfg["Jup"], tf, 0.1 * tf, cfg["bounds"], 2, cfg["T_d"])
d = DeviceEnsembleSampler(nw, 4 * ncomp, engine=e, seed=5, group=dist.group.WORLD, nens=nens, ens_src=ens_src, schedule="dataflow")
d.fallback = False                                     # a timeout is a failure here, not a silent half-step run
if os.environ.get("RX_TEST_INJECT_PEER_ERROR"):        # (this worker's own switches: the sampler itself reads no environment variable)
    d._inject = ("error", int(os.environ["RX_TEST_INJECT_PEER_ERROR"]))
if os.environ.get("RX_TEST_INJECT_PEER_STALL"):        # (the stall test: the watchdog path IS the expected one, and quick)
    d._inject = ("stall", int(os.environ["RX_TEST_INJECT_PEER_STALL"]))
    d.fallback = True
    d.verify_peer_steps = 0
    e.set_sampler_timeout_ms(1500.0)
from radex_emcee_amd.engine import EngineError
try:
    st = d.run_mcmc(cfg["walkers"], nsteps)
    st = d.run_mcmc(st, 3)                                 # a second call: replica re-seeded, step counter continues
except EngineError as exc:                              # (the injected-failure test: every rank must get here, none may hang)
    open(out + "/peer_%d.err" % rank, "w").write(str(exc))
    dist.destroy_process_group()
    sys.exit(3)
np.savez(out + "/peer_%d.npz" % rank, coords=st.coords, lnp=st.log_prob, chain=d.get_chain(), chain_lnp=d.get_log_prob(),
         acc=d.acceptance_fraction, schedule=np.array(d.last_schedule), peer=np.array(str(d.peer_state)),
         verified=np.array(str(d.peer_verified)), detail=np.array(json.dumps(d.peer_verify_detail)),
         reason=np.array(str(d.schedule_reason)), same_device=np.array(e.sampler_peer_same_device() if d.peer_state is True else -1))
dist.barrier(); dist.destroy_process_group()
e.close()
'''


@pytest.mark.parametrize("shape,nw,nsteps", [("config2", 1024, 13), ("config4", 256, 4), ("config3", 250, 5)])
def test_peer_dataflow_two_processes_ipc(co_path, mol, tmp_path, shape, nw, nsteps):
    """One process per rank, both on GPU 0, DeviceEnsembleSampler(group=...): the replicas are exported with
    hipIpcGetMemHandle and mapped with hipIpcOpenMemHandle -- the code path of one process per GPU -- and the
    chain equals the one-GPU dataflow chain bit for bit on every rank."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "peer_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", str(port), str(tmp_path), shape, str(nw),
                               str(nsteps)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    e = Engine(co_path)
    if shape == "config3":                                   # 3 ensembles x 250 walkers: 375 proposals per half-step,
        c3 = workloads.config3(nw, init="ball")              # blocks of 188 + 187 that cut through the second ensemble
        for k in range(3):
            s3 = c3["sources"][k]
            e.set_source(s3["tbg"], s3["Jup"], s3["flux"], s3["eflux"], s3["bounds"], src=k)
        p0, ncomp = c3["walkers"][:3], 1
        ref = DeviceEnsembleSampler(nw, 4, engine=e, seed=5, nens=3, ens_src=np.arange(3))
    else:
        p0, ncomp = _setup(e, mol, shape, nw)
        ref = DeviceEnsembleSampler(nw, 4 * ncomp, engine=e, seed=5)
    st = ref.run_mcmc(p0, nsteps)
    st = ref.run_mcmc(st, 3)
    for r in range(2):
        z = np.load(tmp_path / ("peer_%d.npz" % r))
        assert str(z["schedule"]) == "dataflow-peer", (str(z["schedule"]), str(z["peer"]))
        # the sampler checked its first steps against the half-step schedule on both ranks before it relied on the peer path,
        # and the library saw that both replicas live on ONE device (it splits the compute units by itself then)
        assert str(z["verified"]) == "True" and "identical" in str(z["reason"]), (str(z["verified"]), str(z["detail"]))
        assert int(z["same_device"]) == 2
        assert np.array_equal(z["coords"], st.coords) and np.array_equal(z["lnp"], st.log_prob)
        assert np.array_equal(z["chain"], ref.get_chain()) and np.array_equal(z["chain_lnp"], ref.get_log_prob())
        assert np.array_equal(z["acc"], ref.acceptance_fraction)
    e.close()


@pytest.mark.parametrize("failing", [(0, 1), (1,)])
def test_peer_unavailable_falls_back_to_halfsteps_on_every_rank(co_path, mol, tmp_path, failing):
    """When the replicas cannot be set up (here: switched off with RX_NO_PEER=1 on both ranks or on ONE of them, as a
    failing hipExtMallocWithFlags / hipIpcGetMemHandle would on a real node) the ranks agree on it and run the
    half-step schedule with the all-gather: the same chain, no hang."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "peer_worker.py"
    script.write_text(WORKER)
    envs = [dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RX_NO_PEER="1" if r in failing else "0") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, "-W", "ignore", str(script), ROOT, str(r), "2", str(port), str(tmp_path), "config2",
                               "256", "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=envs[r]) for r in range(2)]
    outs = [p.communicate(timeout=420)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    e = Engine(co_path)
    p0, ncomp = _setup(e, mol, "config2", 256)
    ref = DeviceEnsembleSampler(256, 4, engine=e, seed=5)
    st = ref.run_mcmc(p0, 4)
    st = ref.run_mcmc(st, 3)
    for r in range(2):
        z = np.load(tmp_path / ("peer_%d.npz" % r))
        assert str(z["schedule"]) == "halfsteps" and "RX_NO_PEER" in str(z["peer"])
        assert np.array_equal(z["coords"], st.coords) and np.array_equal(z["chain"], ref.get_chain())
    e.close()


def test_peer_error_on_one_rank_raises_on_every_rank(co_path, mol, tmp_path):
    """A launch failure on rank 1 (injected: RX_TEST_INJECT_PEER_ERROR) must not leave rank 0 in a barrier until the
    process-group timeout: rank 1 raises the abort word in every replica from the host (rx_sampler_peer_abort), rank 0's
    kernel drains, the ranks exchange what happened and BOTH raise EngineError -- within seconds."""
    import time
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "peer_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RX_TEST_INJECT_PEER_ERROR="1")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-W", "ignore", str(script), ROOT, str(r), "2", str(port), str(tmp_path), "config2",
                               "256", "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung behind the failing one")
    assert [p.returncode for p in procs] == [3, 3], "\n".join(outs)
    for r in range(2):
        msg = open(tmp_path / ("peer_%d.err" % r)).read()
        assert "rank 1" in msg and "injected launch failure" in msg, msg
    assert time.time() - t0 < 200


def test_two_engines_sample_concurrently_on_one_gpu(co_path, mol):
    """Two independent fits on ONE GPU (two handles, two streams, both persistent dataflow kernels in flight together):
    a task only waits for tasks of its own launch, so neither can starve the other -- both chains are the chains the
    engines produce alone, and nobody runs into the watchdog (a run that did would have been repeated per half-step)."""
    import torch
    from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
    engs = [Engine(co_path), Engine(co_path)]
    p0s, refs = [], []
    for k, e in enumerate(engs):
        p0, _ = _setup(e, mol, "config2", 1024)
        p0s.append(p0 + 1e-3 * k)
        r = DeviceEnsembleSampler(1024, 4, engine=e, seed=21 + k)
        refs.append((r.run_mcmc(p0s[k], 30), r))
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in engs]
    smp = [DeviceEnsembleSampler(1024, 4, engine=e, seed=21 + k) for k, e in enumerate(engs)]
    chains, lnps = [], []
    for k, (e, d) in enumerate(zip(engs, smp)):                # both launches are enqueued before either is waited for
        d.fallback = False
        d.coords.copy_(torch.from_numpy(p0s[k]))
        d.lnp.copy_(d.compute_log_prob(p0s[k]))
        chains.append(torch.empty(30, 1024, 4, dtype=torch.float64, device=dev))
        lnps.append(torch.empty(30, 1024, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    for k, (e, d) in enumerate(zip(engs, smp)):
        e.sampler_run_async_torch(1, 1024, 1, 2.0, 21 + k, 0, 30, d.coords, d.lnp, d.naccept, chains[k], lnps[k],
                                  stream=streams[k].cuda_stream)
    for k, e in enumerate(engs):
        e.sampler_wait(dev, stream=streams[k].cuda_stream)     # raises RX_E_TIMEOUT if a task ran into the watchdog
    for k, d in enumerate(smp):
        st_ref, r = refs[k]
        assert np.array_equal(d.coords.cpu().numpy(), st_ref.coords) and np.array_equal(d.lnp.cpu().numpy(), st_ref.log_prob)
        assert np.array_equal(chains[k].cpu().numpy(), r.get_chain())
    for e in engs:
        e.close()


def test_peer_whose_kernel_never_starts_is_given_up_on_and_the_run_repeated(co_path, mol, tmp_path):
    """Rank 1's kernel never starts (injected: RX_TEST_INJECT_PEER_STALL).  Rank 0's tasks wait for results that never come; the
    no-progress watchdog is not armed (not every rank's grid is running: that could be a slow code-object load), so the flat
    bound of a single wait -- 1.5 s here -- ends the run: abort word in every replica, both ranks see RX_E_TIMEOUT, both repeat
    the call under the half-step schedule, and the chain is the chain."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "peer_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RX_TEST_INJECT_PEER_STALL="1")
    procs = [subprocess.Popen([sys.executable, "-W", "ignore", str(script), ROOT, str(r), "2", str(port), str(tmp_path), "config2",
                               "256", "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    e = Engine(co_path)
    p0, ncomp = _setup(e, mol, "config2", 256)
    ref = DeviceEnsembleSampler(256, 4, engine=e, seed=5)
    st = ref.run_mcmc(p0, 4)
    st = ref.run_mcmc(st, 3)
    for r in range(2):
        z = np.load(tmp_path / ("peer_%d.npz" % r))
        assert str(z["schedule"]) == "halfsteps" and "abandoned" in str(z["reason"]), (str(z["schedule"]), str(z["reason"]))
        assert np.array_equal(z["coords"], st.coords) and np.array_equal(z["chain"], ref.get_chain())
    e.close()
