"""CPU tests of the ensemble sampler (stretch move) and of the multi-rank log-prob sharding
(world_size-2 gloo).  The likelihood here is an analytic stand-in: the HIP engine needs a GPU."""
import os
import socket
import sys

import numpy as np
import pytest

from radex_emcee_amd.sampler import EnsembleSampler, ShardedLogProb, State

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gauss_batch(P, mu, isig):
    d = (P - mu) * isig
    return -0.5 * np.sum(d * d, axis=1)


def gauss_one(p, mu, isig):
    d = (p - mu) * isig
    return -0.5 * float(np.dot(d, d))


def test_api_shapes_and_reset():
    mu, isig = np.array([1.0, -2.0, 0.5, 3.0]), 1.0 / np.array([0.5, 1.0, 2.0, 0.1])
    s = EnsembleSampler(32, 4, gauss_batch, args=(mu, isig), vectorize=True, seed=1)
    p0 = mu + 1e-3 * np.random.RandomState(0).randn(32, 4)
    st = s.run_mcmc(p0, 20, progress=False)
    assert isinstance(st, State) and st.coords.shape == (32, 4) and st.log_prob.shape == (32,)
    assert s.get_chain().shape == (20, 32, 4) and s.get_log_prob().shape == (20, 32)
    assert s.get_chain(flat=True).shape == (640, 4) and s.get_log_prob(flat=True).shape == (640,)
    assert s.nevals == 32 + 20 * 32
    s.reset()
    assert s.get_chain().shape == (0, 32, 4) and s.iteration == 0
    st2 = s.run_mcmc(st, 5)
    assert s.get_chain().shape == (5, 32, 4) and np.all(np.isfinite(st2.log_prob))
    with pytest.raises(ValueError):
        EnsembleSampler(6, 4, gauss_batch)


def test_vectorized_and_per_walker_paths_agree():
    mu, isig = np.zeros(3), np.ones(3)
    p0 = np.random.RandomState(3).randn(16, 3)
    a = EnsembleSampler(16, 3, gauss_batch, args=(mu, isig), vectorize=True, seed=7)
    b = EnsembleSampler(16, 3, gauss_one, args=(mu, isig), seed=7)
    sa, sb = a.run_mcmc(p0, 30), b.run_mcmc(p0, 30)
    assert np.allclose(sa.coords, sb.coords, rtol=0, atol=1e-12)
    assert np.array_equal(a.acceptance_fraction, b.acceptance_fraction)


def test_stretch_move_samples_the_target():
    rng = np.random.RandomState(11)
    mu, sig = np.array([0.3, -1.0, 2.0, 0.0]), np.array([0.2, 1.5, 0.7, 1.0])
    s = EnsembleSampler(64, 4, gauss_batch, args=(mu, 1.0 / sig), vectorize=True, seed=5)
    st = s.run_mcmc(mu + 1e-3 * rng.randn(64, 4), 400)
    s.reset()
    s.run_mcmc(st, 1500)
    flat = s.get_chain(flat=True)
    assert np.all(np.abs(flat.mean(0) - mu) < 0.08 * sig + 0.02)
    assert np.all(np.abs(flat.std(0) / sig - 1.0) < 0.08)
    assert 0.3 < s.acceptance_fraction.mean() < 0.8


def test_nan_and_minus_inf_handling():
    def bad(P):
        out = np.zeros(len(P)); out[0] = np.nan
        return out
    s = EnsembleSampler(8, 2, bad, vectorize=True, seed=0)
    with pytest.raises(ValueError):
        s.run_mcmc(np.random.RandomState(0).randn(8, 2), 1)

    def box(P):          # -inf outside the unit box: proposals there are always rejected
        return np.where(np.all(np.abs(P) < 1.0, axis=1), 0.0, -np.inf)
    s = EnsembleSampler(16, 2, box, vectorize=True, seed=2)
    st = s.run_mcmc(0.5 * (np.random.RandomState(1).rand(16, 2) - 0.5), 200)
    assert np.all(np.abs(st.coords) < 1.0)


WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from radex_emcee_amd.sampler import EnsembleSampler, ShardedLogProb
rank, world, port = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
mu, isig = np.array([1.0, -2.0, 0.5, 3.0]), 1.0 / np.array([0.5, 1.0, 2.0, 0.1])
calls = []
def local(P):
    calls.append(len(P))
    d = (P - mu) * isig
    return -0.5 * np.sum(d * d, axis=1)
f = ShardedLogProb(local)
s = EnsembleSampler(26, 4, f, vectorize=True, seed=123)        # 13 per half: ragged over 2 ranks
p0 = mu + 1e-3 * np.random.RandomState(0).randn(26, 4)
st = s.run_mcmc(p0, 25)
np.save(sys.argv[5] + "/coords_%d.npy" % rank, st.coords)
np.save(sys.argv[5] + "/calls_%d.npy" % rank, np.array(calls))
# empty shard: fewer rows than ranks
out = f(p0[:1])
assert out.shape == (1,) and np.isfinite(out[0])
dist.barrier(); dist.destroy_process_group()
'''


def test_sharded_logprob_two_ranks_gloo(tmp_path):
    import subprocess
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", str(port), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    c0, c1 = np.load(tmp_path / "coords_0.npy"), np.load(tmp_path / "coords_1.npy")
    assert np.array_equal(c0, c1)                          # every rank holds the same ensemble
    # single-process reference run: identical chain
    mu, isig = np.array([1.0, -2.0, 0.5, 3.0]), 1.0 / np.array([0.5, 1.0, 2.0, 0.1])
    s = EnsembleSampler(26, 4, gauss_batch, args=(mu, isig), vectorize=True, seed=123)
    st = s.run_mcmc(mu + 1e-3 * np.random.RandomState(0).randn(26, 4), 25)
    assert np.array_equal(st.coords, c0)
    k0, k1 = np.load(tmp_path / "calls_0.npy"), np.load(tmp_path / "calls_1.npy")
    assert k0[0] == 13 and k1[0] == 13                     # initial 26 walkers split 13/13
    assert set(k0[1:51]) == {7} and set(k1[1:51]) == {6}   # 13 proposals per half-step -> 7 + 6


# ---- counter-based stream shared with the device kernels (csrc/rx_sampler.hip.inc) ----------------
def test_philox_known_answers():
    """Random123's published known-answer vectors for philox4x32-10."""
    from radex_emcee_amd.sampler import philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


@pytest.mark.parametrize("n", [2, 8, 26, 100, 1024, 1500])
def test_walker_permutation_is_a_bijection_and_changes_with_the_step(n):
    from radex_emcee_amd.sampler import walker_permutation
    p0, p1 = walker_permutation(n, 99, 0, 0), walker_permutation(n, 99, 1, 0)
    assert sorted(p0) == list(range(n)) and sorted(p1) == list(range(n))
    if n >= 26:
        assert not np.array_equal(p0, p1)
        assert not np.array_equal(p0, walker_permutation(n, 99, 0, 1))     # per ensemble
        # a balanced, well-mixed split: about half of the first half's walkers are even-numbered
        assert abs(np.mean(p0[:n // 2] % 2) - 0.5) < 0.25


def test_philox_mode_samples_the_target_and_matches_the_device_mirror():
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    mu, sig = np.array([1.0, -2.0, 0.5]), np.array([0.5, 1.0, 2.0])
    s = EnsembleSampler(40, 3, gauss_batch, args=(mu, 1.0 / sig), vectorize=True, seed=11, rng="philox")
    p0 = mu + 1e-2 * np.random.RandomState(3).randn(40, 3)
    s.run_mcmc(p0, 1500)
    flat = s.get_chain(flat=True, discard=300)
    assert np.all(np.abs(flat.mean(0) - mu) < 0.1 * sig + 0.02)
    assert np.all(np.abs(flat.std(0) / sig - 1.0) < 0.1)
    assert 0.3 < s.acceptance_fraction.mean() < 0.8
    # the device sampler's host backend (the numpy restatement of the kernels) is the same chain, bit for bit
    d = DeviceEnsembleSampler(40, 3, log_prob_fn=lambda P: gauss_batch(P, mu, 1.0 / sig), seed=11)
    d.run_mcmc(p0, 200)
    assert np.array_equal(d.get_chain(), s.get_chain()[:200])
    assert np.array_equal(d.get_log_prob(), s.get_log_prob()[:200])
    # resuming continues the counter: 2 x 100 steps == 200 steps
    d2 = DeviceEnsembleSampler(40, 3, log_prob_fn=lambda P: gauss_batch(P, mu, 1.0 / sig), seed=11)
    st = d2.run_mcmc(p0, 100)
    d2.run_mcmc(st, 100)
    assert np.array_equal(d2.get_chain(), d.get_chain())


def test_several_ensembles_advance_independently():
    """BASELINE config 3's shape: one ensemble per source, one batch per half-step."""
    from radex_emcee_amd.sampler import DeviceEnsembleSampler
    mus = np.array([[0.0, 0.0], [5.0, -5.0], [-3.0, 2.0]])

    def lp(P, src):
        return -0.5 * np.sum((P - mus[src]) ** 2, axis=1)
    d = DeviceEnsembleSampler(16, 2, log_prob_fn=lp, nens=3, ens_src=[0, 1, 2], seed=4)
    p0 = mus[:, None, :] + 0.1 * np.random.RandomState(0).randn(3, 16, 2)
    st = d.run_mcmc(p0, 400)
    ch = d.get_chain(discard=100)
    assert ch.shape == (300, 3, 16, 2) and st.coords.shape == (3, 16, 2)
    for e in range(3):
        assert np.all(np.abs(ch[:, e].reshape(-1, 2).mean(0) - mus[e]) < 0.25)
        # ensemble e alone, same seed: the same chain (ensembles never mix)
    solo = DeviceEnsembleSampler(16, 2, log_prob_fn=lambda P: lp(P, 0), seed=4)
    solo.run_mcmc(p0[0], 50)
    assert np.array_equal(solo.get_chain(), d.get_chain()[:50, 0])


WORKER_DEV = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from radex_emcee_amd.sampler import DeviceEnsembleSampler, ShardedLogProb
rank, world, port = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
mu = np.array([1.0, -2.0, 0.5, 3.0])
calls = []
def lp(P):
    calls.append(len(P))
    return -0.5 * np.sum((P - mu) ** 2, axis=1)
# the code path bench.py --gpus N times: replicated state, block-sharded proposals, one all_gather per
# half-step, identical accept on every rank -- here with the numpy restatement of the kernels
d = DeviceEnsembleSampler(26, 4, log_prob_fn=lp, seed=123, group=dist.group.WORLD, schedule="halfsteps")   # 13 proposals: ragged 7 + 6
p0 = mu + 1e-3 * np.random.RandomState(0).randn(26, 4)
lnp0 = -0.5 * np.sum((p0 - mu) ** 2, axis=1)
from radex_emcee_amd.sampler import State
st = d.run_mcmc(State(p0, lnp0), 25)
np.save(sys.argv[5] + "/dcoords_%d.npy" % rank, st.coords)
np.save(sys.argv[5] + "/dcalls_%d.npy" % rank, np.array(calls))
assert d.schedule == "halfsteps" and d.last_schedule == "halfsteps" and d.schedule_reason == "requested"
ncalls_half = len(calls)
# with a group the DEFAULT is "auto": 13 proposals per half-step are far below one GPU's latency regime, so by rule rank 0
# advances the ensemble alone and ONE broadcast per run_mcmc call hands the result to the others
da = DeviceEnsembleSampler(26, 4, log_prob_fn=lp, seed=123, group=dist.group.WORLD)
assert da.schedule == "auto"
sta = da.run_mcmc(State(p0, lnp0), 25)
assert da.last_schedule == "rank0" and da.schedule_choice == "rank0" and "by rule" in da.schedule_reason, da.schedule_reason
assert np.array_equal(sta.coords, st.coords) and np.array_equal(da.get_chain(), d.get_chain())      # the same chain, every rank
assert np.array_equal(da.get_log_prob(), d.get_log_prob()) and np.array_equal(da.acceptance_fraction, d.acceptance_fraction)
assert len(calls) - ncalls_half == (50 if rank == 0 else 0)       # only rank 0 evaluated anything: 50 half-steps of 13
assert set(calls[ncalls_half:]) <= {13}
stb = da.run_mcmc(sta, 5); stc = d.run_mcmc(st, 5)                  # resuming: still the same chain
assert np.array_equal(stb.coords, stc.coords) and np.array_equal(stb.log_prob, stc.log_prob)
# above the rule's threshold the candidates are TIMED and every rank takes the same decision
db = DeviceEnsembleSampler(26, 4, log_prob_fn=lp, seed=123, group=dist.group.WORLD, verify_peer_steps=2)
db.AUTO_RANK0_TASKS = 4
std = db.run_mcmc(State(p0, lnp0), 25)
assert db.schedule_choice in ("rank0", "halfsteps") and "by probe" in db.schedule_reason, db.schedule_reason
assert set(db.auto_probe["seconds"]) == {"rank0", "halfsteps"} and db.auto_probe["peer_candidate"] is False
choices = [None, None]
dist.all_gather_object(choices, (db.schedule_choice, db.last_schedule))
assert choices[0] == choices[1], choices
assert np.array_equal(std.coords, st.coords) and np.array_equal(db.get_chain(), d.get_chain()[:25])
np.save(sys.argv[5] + "/dauto_%d.npy" % rank, sta.coords)
del calls[ncalls_half:]
# the collective that replaces the bare barriers of the peer protocol: one rank's failure is seen by EVERY rank
assert d._agree(None) == []
bad = d._agree("boom on rank 1" if rank == 1 else None)
assert bad == [(1, "boom on rank 1")], bad
# tensor form of ShardedLogProb: block in, block out, collective on tensors
def lp_t(P, out):
    out.copy_(-0.5 * ((P - torch.from_numpy(mu)) ** 2).sum(1))
f = ShardedLogProb(lp_t, tensors=True)
got = f(torch.from_numpy(p0))
assert torch.is_tensor(got) and np.allclose(got.numpy(), lnp0, rtol=0, atol=0)
assert f(torch.from_numpy(p0[:1])).shape == (1,)                  # fewer rows than ranks: an empty shard
dist.barrier(); dist.destroy_process_group()
'''


def test_device_sampler_sharded_two_ranks_gloo(tmp_path):
    import subprocess
    from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "worker_dev.py"
    script.write_text(WORKER_DEV)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", str(port), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    c0, c1 = np.load(tmp_path / "dcoords_0.npy"), np.load(tmp_path / "dcoords_1.npy")
    assert np.array_equal(c0, c1)
    mu = np.array([1.0, -2.0, 0.5, 3.0])
    lp = lambda P: -0.5 * np.sum((P - mu) ** 2, axis=1)
    p0 = mu + 1e-3 * np.random.RandomState(0).randn(26, 4)
    d = DeviceEnsembleSampler(26, 4, log_prob_fn=lp, seed=123)
    st = d.run_mcmc(State(p0, lp(p0)), 25)
    assert np.array_equal(st.coords, c0)                    # sharded == unsharded, bit for bit
    k0, k1 = np.load(tmp_path / "dcalls_0.npy"), np.load(tmp_path / "dcalls_1.npy")
    assert set(k0) == {7} and set(k1) == {6} and len(k0) == 50
    # schedule="auto" (rank 0 alone + one broadcast): the unsharded chain on BOTH ranks, bit for bit
    assert np.array_equal(np.load(tmp_path / "dauto_0.npy"), c0) and np.array_equal(np.load(tmp_path / "dauto_1.npy"), c0)
