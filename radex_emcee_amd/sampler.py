"""Ensemble sampler (affine-invariant stretch move) driving the batched GPU likelihood.

API subset of emcee v3 actually used by the reference
[/root/reference/emcee/emcee_radex.py:483-499, emcee/emcee_radex_2comp.py:557-574]:
    EnsembleSampler(nwalkers, ndim, log_prob_fn, args=, kwargs=, pool=, vectorize=)
    .run_mcmc(pos | state, nsteps, progress=False) -> State
    .reset(); .get_chain(flat=, discard=, thin=); .get_log_prob(flat=, ...); .acceptance_fraction

Algorithm = emcee's `StretchMove` (a = 2) inside `RedBlueMove` with two splits: walkers
are dealt to two halves through a shuffled `arange(n) % 2`; each half is updated against
the other: z = ((a-1)u+1)^2/a, partner c uniform from the other half, q = c-(c-s)z,
accept iff (ndim-1) ln z + lnp(q) - lnp(s) > ln u'.  A NaN log-probability is an error.

Where the reference evaluates `lnprob` once per walker through `multiprocessing.Pool.map`
(emcee_radex.py:480-488), this sampler hands each half-ensemble to `log_prob_fn` as ONE
[N/2, ndim] batch (vectorize=True) -> one kernel launch on the GPU.

`DeviceEnsembleSampler` is the same move with positions, log-probabilities, proposals, accept/reject
and a counter-based random stream resident on the GPU (csrc/rx_sampler.hip.inc); the host sampler in
rng="philox" mode replays its stream and is its checker.

Multi-GPU: `ShardedLogProb` splits that batch into contiguous blocks, one per rank
(one process per GPU), evaluates its block locally and all-gathers the log-probabilities
(torch.distributed: "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests).  Positions and
the random stream are replicated (same seed on every rank), so proposals need no exchange
and every rank performs the identical accept/reject.
"""
from __future__ import annotations

import numpy as np


# ---- counter-based random stream shared with the device kernels ---------------------------------
# numpy restatement of radex_emcee_amd/csrc/rx_sampler.hip.inc (Philox4x32-10, the 53-bit uniforms,
# the keyed permutation of the walkers): the host sampler in rng="philox" mode replays exactly the
# variates the device kernels draw, which makes it the checker of the on-device stretch move.
PURPOSE_PROPOSE, PURPOSE_ACCEPT, PURPOSE_PERM = 0, 1, 2
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011) on arrays of 32-bit counters; returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _U32 for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0), p1 & _U32,
                          (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1), p0 & _U32)
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u53(hi, lo):
    v = ((hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)) >> np.uint64(11)
    return v.astype(np.float64) * (1.0 / 9007199254740992.0)


def walker_permutation(n, seed, step, ens):
    """perm[position] = walker: positions [0, n/2) are the first half of this step's split."""
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    b = 1
    while (1 << b) < n:
        b += 1
    mask = np.uint64((1 << b) - 1)
    s1, s2 = np.uint64(max(b // 2, 1)), np.uint64(max(b // 3, 1))
    ka = philox4x32_10(0, ens, step, PURPOSE_PERM, lo, hi)
    kc = philox4x32_10(1, ens, step, PURPOSE_PERM, lo, hi)
    mul = [np.uint64(int(ka[i]) | 1) for i in range(3)]
    add = [np.uint64(int(kc[i])) for i in range(3)]

    def rounds(v):
        v = (v * mul[0] + add[0]) & mask; v ^= v >> s1
        v = (v * mul[1] + add[1]) & mask; v ^= v >> s2
        v = (v * mul[2] + add[2]) & mask; v ^= v >> s1
        return v
    v = rounds(np.arange(n, dtype=np.uint64))
    while True:
        out = v >= np.uint64(n)
        if not out.any():
            break
        v[out] = rounds(v[out])                          # cycle walking
    return v.astype(np.int64)


def stretch_propose(coords, nens, nwalkers, a, seed, step, split):
    """numpy mirror of rx_stretch_propose_kernel: returns (q, factor, widx)."""
    ndim = coords.shape[-1]
    h = nwalkers // 2
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    X = coords.reshape(nens * nwalkers, ndim)
    q = np.empty((nens * h, ndim))
    factor = np.empty(nens * h)
    widx = np.empty(nens * h, dtype=np.int32)
    j = np.arange(h)
    for e in range(nens):
        perm = walker_permutation(nwalkers, seed, step, e)
        r = philox4x32_10(j, e, step, PURPOSE_PROPOSE + 16 * split, lo, hi)
        u, u2 = u53(r[0], r[1]), u53(r[2], r[3])
        t1 = (a - 1.0) * u + 1.0
        z = t1 * t1 / a
        ri = np.minimum((u2 * float(h)).astype(np.int64), h - 1)
        ws = e * nwalkers + perm[split * h + j]
        wc = e * nwalkers + perm[(1 - split) * h + ri]
        sl = slice(e * h, (e + 1) * h)
        q[sl] = X[wc] - (X[wc] - X[ws]) * z[:, None]
        factor[sl] = (ndim - 1.0) * np.log(z)
        widx[sl] = ws
    return q, factor, widx


def stretch_accept(coords, lnp, naccept, q, lnp_q, factor, widx, nens, nwalkers, seed, step, split):
    """numpy mirror of rx_stretch_accept_kernel (in place)."""
    h = nwalkers // 2
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    X = coords.reshape(nens * nwalkers, -1)
    j = np.arange(h)
    for e in range(nens):
        r = philox4x32_10(j, e, step, PURPOSE_ACCEPT + 16 * split, lo, hi)
        with np.errstate(divide="ignore", invalid="ignore"):
            lu = np.log(u53(r[0], r[1]))
            sl = slice(e * h, (e + 1) * h)
            w = widx[sl]
            acc = lu < (factor[sl] + lnp_q[sl]) - lnp[w]
        X[w[acc]] = q[sl][acc]
        lnp[w[acc]] = lnp_q[sl][acc]
        if naccept is not None:
            naccept[w[acc]] += 1


class State:
    def __init__(self, coords, log_prob=None, random_state=None):
        self.coords = np.array(coords, dtype=np.float64, copy=True)
        self.log_prob = None if log_prob is None else np.array(log_prob, dtype=np.float64, copy=True)
        self.random_state = random_state

    def __iter__(self):
        return iter((self.coords, self.log_prob, self.random_state))


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, log_prob_fn, args=None, kwargs=None, pool=None,
                 vectorize=False, a=2.0, seed=None, rng="numpy"):
        if nwalkers < 2 * ndim:
            raise ValueError("The number of walkers needs to be at least twice the dimension "
                             "of your parameter space")
        if nwalkers % 2:
            raise ValueError("The number of walkers must be even")
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self.log_prob_fn = log_prob_fn
        self.args = tuple(args or ())
        self.kwargs = dict(kwargs or {})
        self.pool = pool
        self.vectorize = bool(vectorize)
        self.a = float(a)
        self._random = np.random.RandomState(seed)
        # rng="philox": draw the counter-based stream of the device kernels (checker of
        # DeviceEnsembleSampler) instead of numpy's Mersenne Twister (emcee's own generator)
        if rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")
        self.rng = rng
        self.seed = 0 if seed is None else int(seed)
        self.step_counter = 0
        self.reset()

    # --- bookkeeping ---------------------------------------------------------------------
    def reset(self):
        self._chain = []
        self._log_prob = []
        self._accepted = np.zeros(self.nwalkers)
        self.iteration = 0
        self.nevals = 0

    @property
    def random_state(self):
        return self._random.get_state()

    @property
    def acceptance_fraction(self):
        return self._accepted / float(max(self.iteration, 1))

    def get_chain(self, flat=False, thin=1, discard=0):
        v = np.array(self._chain)[discard::thin]
        if len(v) == 0:
            v = np.empty((0, self.nwalkers, self.ndim))
        return v.reshape(-1, self.ndim) if flat else v

    def get_log_prob(self, flat=False, thin=1, discard=0):
        v = np.array(self._log_prob)[discard::thin]
        if len(v) == 0:
            v = np.empty((0, self.nwalkers))
        return v.reshape(-1) if flat else v

    # --- log-probability ---------------------------------------------------------------------
    def compute_log_prob(self, coords):
        p = np.asarray(coords, dtype=np.float64)
        if np.any(np.isinf(p)):
            raise ValueError("At least one parameter value was infinite")
        if np.any(np.isnan(p)):
            raise ValueError("At least one parameter value was NaN")
        if self.vectorize:
            lp = np.asarray(self.log_prob_fn(p, *self.args, **self.kwargs), dtype=np.float64)
        else:
            mapper = self.pool.map if self.pool is not None else map
            fn = _Wrapper(self.log_prob_fn, self.args, self.kwargs)
            lp = np.array([float(x) for x in mapper(fn, (p[i] for i in range(len(p))))])
        self.nevals += len(p)
        if lp.shape != (len(p),):
            raise ValueError("log_prob_fn returned the wrong shape")
        if np.any(np.isnan(lp)):
            raise ValueError("Probability function returned NaN")
        return lp

    # --- sampling -----------------------------------------------------------------------------
    def run_mcmc(self, initial_state, nsteps, progress=False, store=True):
        state = initial_state if isinstance(initial_state, State) else State(np.atleast_2d(initial_state))
        if state.coords.shape != (self.nwalkers, self.ndim):
            raise ValueError("incompatible input dimensions")
        if state.random_state is not None:
            try:
                self._random.set_state(state.random_state)
            except Exception:
                pass
        if state.log_prob is None:
            state.log_prob = self.compute_log_prob(state.coords)
        if np.any(np.isnan(state.log_prob)):
            raise ValueError("The initial log_prob was NaN")
        for _ in range(int(nsteps)):
            self._step(state)
            self.iteration += 1
            if store:
                self._chain.append(state.coords.copy())
                self._log_prob.append(state.log_prob.copy())
        state.random_state = self.random_state
        return state

    def _step_philox(self, state):
        n, a, seed, step = self.nwalkers, self.a, self.seed, self.step_counter
        for split in range(2):
            q, factor, widx = stretch_propose(state.coords, 1, n, a, seed, step, split)
            new_lp = self.compute_log_prob(q)
            stretch_accept(state.coords, state.log_prob, self._accepted, q, new_lp, factor, widx, 1, n, seed,
                           step, split)
        self.step_counter += 1

    def _step(self, state):
        if self.rng == "philox":
            return self._step_philox(state)
        n, ndim, a, rng = self.nwalkers, self.ndim, self.a, self._random
        all_inds = np.arange(n)
        inds = all_inds % 2
        rng.shuffle(inds)
        for split in range(2):
            S1 = inds == split
            s = state.coords[S1]
            c = state.coords[~S1]
            Ns, Nc = len(s), len(c)
            zz = ((a - 1.0) * rng.rand(Ns) + 1) ** 2.0 / a
            factors = (ndim - 1.0) * np.log(zz)
            rint = rng.randint(Nc, size=(Ns,))
            q = c[rint] - (c[rint] - s) * zz[:, None]
            new_lp = self.compute_log_prob(q)
            with np.errstate(invalid="ignore"):          # (-inf) - (-inf) = NaN -> rejected, as in emcee
                lnpdiff = factors + new_lp - state.log_prob[all_inds[S1]]
                accepted = np.log(rng.rand(Ns)) < lnpdiff
            idx = all_inds[S1][accepted]
            state.coords[idx] = q[accepted]
            state.log_prob[idx] = new_lp[accepted]
            self._accepted[idx] += 1


class _Wrapper:
    def __init__(self, f, args, kwargs):
        self.f, self.args, self.kwargs = f, args, kwargs

    def __call__(self, x):
        return self.f(x, *self.args, **self.kwargs)


def block_partition(N, world, rank):
    """Contiguous blocks of ceil(N/world) rows: (lo, hi, per).  Every rank but the last non-empty one
    holds a full block, so the gathered [world*per] vector holds the N results in order."""
    per = -(-N // world)
    lo = min(rank * per, N)
    return lo, min(lo + per, N), per


class ShardedLogProb:
    """Evaluate a [N, ndim] batch across the ranks of a torch.distributed group.

    Rank r evaluates rows [r*ceil(N/G), ...) with `local_fn` (its own GPU), then ONE
    all_gather of float64 log-probabilities makes the full vector available everywhere
    (N/(2G) doubles per rank per half-step; latency-bound).

    tensors=True: tensor in, tensor out -- with a CUDA batch the block, the evaluator's output and
    the collective (RCCL over xGMI) all stay in HBM on the current stream; `local_fn(P[lo:hi],
    out[:hi-lo])` writes its block of log-probabilities into `out`.  tensors=False: `local_fn` maps
    a numpy block to numpy log-probabilities (host evaluators, gloo tests).  A numpy batch is wrapped
    without a copy and the result returned as numpy."""

    def __init__(self, local_fn, group=None, device=None, tensors=False):
        import torch.distributed as dist
        self.local_fn = local_fn
        self.tensors = bool(tensors)      # local_fn(block tensor, out tensor) instead of numpy -> numpy
        self.group = group
        self.dist = dist
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self._buf = {}

    def partition(self, N):
        return block_partition(N, self.world, self.rank)

    def _buffers(self, per, dev):
        import torch
        key = (per, str(dev))
        if key not in self._buf:
            self._buf[key] = (torch.empty(per, dtype=torch.float64, device=dev),
                              torch.empty(per * self.world, dtype=torch.float64, device=dev))
        return self._buf[key]

    def gather(self, mine, out):
        if self.world == 1:
            out.copy_(mine)
            return out
        try:
            self.dist.all_gather_into_tensor(out, mine, group=self.group)
        except (RuntimeError, NotImplementedError):      # backend without the flat form
            parts = list(out.view(self.world, -1).unbind(0))
            self.dist.all_gather(parts, mine, group=self.group)
        return out

    def __call__(self, P):
        import torch
        as_numpy = not torch.is_tensor(P)
        if as_numpy:
            P = torch.from_numpy(np.ascontiguousarray(P, dtype=np.float64))
            if self.device is not None and str(self.device) != "cpu":
                P = P.to(self.device)
        N = P.shape[0]
        lo, hi, per = self.partition(N)
        mine, out = self._buffers(per, P.device)
        mine.fill_(float("-inf"))
        if hi > lo:
            if self.tensors:
                self.local_fn(P[lo:hi], mine[:hi - lo])
            else:
                r = self.local_fn(P[lo:hi].cpu().numpy())
                mine[:hi - lo].copy_(torch.as_tensor(np.asarray(r, dtype=np.float64)))
        res = self.gather(mine, out)[:N]
        return res.cpu().numpy() if as_numpy else res


# ---- the stretch move with the state resident on the device -----------------------------------------
class _HostStretchBackend:
    """numpy restatement of the device kernels on CPU tensors: the checker of the device sampler and
    the backend of the multi-process (gloo) tests.  `log_prob_fn(q[n, ndim] numpy) -> lnp[n]`."""

    def __init__(self, log_prob_fn):
        self.log_prob_fn = log_prob_fn
        self.device = "cpu"

    def propose(self, S, step, split):
        q, f, w = stretch_propose(S.coords.numpy(), S.nens, S.nwalkers, S.a, S.seed, step, split)
        S.q.numpy()[:] = q
        S.factor.numpy()[:] = f
        S.widx.numpy()[:] = w
        if S.qsrc is not None:
            S.qsrc.numpy()[:] = np.repeat(S.ens_src.numpy(), S.nwalkers // 2)

    def evaluate(self, S, q, out, qsrc):
        out.numpy()[:] = self.log_prob_fn(q.numpy()) if qsrc is None else self.log_prob_fn(q.numpy(), qsrc.numpy())

    def accept(self, S, step, split, lnp_q):
        stretch_accept(S.coords.numpy(), S.lnp.numpy(), S.naccept.numpy(), S.q.numpy(), lnp_q.numpy(),
                       S.factor.numpy(), S.widx.numpy(), S.nens, S.nwalkers, S.seed, step, split)


class _EngineStretchBackend:
    """rx_stretch_propose_device / rx_lnprob_batch_device / rx_stretch_accept_device on the current stream."""

    def __init__(self, engine):
        import torch
        self.engine = engine
        self.device = torch.device("cuda", engine.device)

    def propose(self, S, step, split):
        self.engine.stretch_propose_torch(S.nens, S.nwalkers, S.a, S.seed, step, split, S.coords, S.q, S.factor,
                                          S.widx, ens_src=S.ens_src, qsrc=S.qsrc)

    def evaluate(self, S, q, out, qsrc):
        n = q.shape[0]
        self.engine.lnprob_batch_torch(q, out, S.qstatus[:n], S.qniter[:n], src_index=qsrc)

    def accept(self, S, step, split, lnp_q):
        self.engine.stretch_accept_torch(S.nens, S.nwalkers, S.seed, step, split, S.q, lnp_q, S.factor, S.widx,
                                         S.coords, S.lnp, S.naccept)


class DeviceEnsembleSampler:
    """emcee's EnsembleSampler API subset (see the module docstring) with walker positions,
    log-probabilities, proposals, accept/reject and the random stream on the device: one half-step =
    propose kernel -> solve kernel over the proposals -> accept kernel, nothing crosses PCIe
    [/root/reference/emcee/emcee_radex.py:483-499].

    engine    radex_emcee_amd.engine.Engine whose source slot(s) are set (GPU path), or
    log_prob_fn  a host function (numpy in, numpy out): CPU restatement of the same kernels (checker)
    nens      independent ensembles advancing together (BASELINE config 3: one per source);
              ens_src[nens] = source slot of each (default: slot 0 for all)
    schedule  "dataflow" (the default on ONE GPU): the chain as one persistent kernel whose tasks start when
              their two input walkers are final -- or earlier, on the newest final position of a walker that is
              still being updated (Engine.set_sampler_speculation; the same chain either way);
              "halfsteps": one propose / solve / accept round per half-step
    group     torch.distributed group, one process per GPU: the proposals of every half-step are dealt out
              in contiguous blocks, one per rank (the reference's Pool.map over walkers,
              emcee_radex.py:480-488).  The DEFAULT with a group is "auto" (below).  "halfsteps": block
              evaluation per rank, ONE all_gather of log-probabilities (RCCL on GPUs) before the accept step --
              north_star's and SURVEY 8e's literal form.  schedule="dataflow" with a group: every rank runs its
              block's tasks in its own persistent kernel and publishes each result into the replicas of ALL
              ranks with peer writes over xGMI (rx_sampler_peer_*: IPC-mapped fine-grained memory, no
              collective, no barrier per half-step, no host in the loop).  That path has only ever run with
              its ranks on ONE physical GPU, so a sampler does not trust it blindly: its first run_mcmc call
              first advances `verify_peer_steps` steps (default 8) under BOTH schedules from the same state
              and compares positions, log-probabilities and acceptance counts bit for bit on every rank
              (`peer_verified`, `peer_verify_detail`); any difference, on any rank, and all ranks use
              "halfsteps" from then on.  It also falls back when the replicas cannot be shared (IPC
              unavailable) or a run is abandoned (watchdog).  Positions and the counter-based random stream
              are replicated either way, so the proposals need no exchange and every rank accepts
              identically.  The chain is the one-GPU chain, bit for bit, under every schedule.
              "rank0": the ensemble is too small to shard -- rank 0 advances it alone with the one-GPU schedule
              (the dataflow kernel on a GPU) and ONE broadcast per run_mcmc call hands positions,
              log-probabilities, acceptance counts and the stored chain to the other ranks.
              "auto" (the default with a group) never lets the group lose to one GPU: with at most
              AUTO_RANK0_TASKS proposals per half-step (one GPU's latency regime: a half-step lasts as long as its
              slowest walker wherever it runs, and sharding only adds a collective to it) it is "rank0" by rule;
              above that the first run_mcmc call times `verify_peer_steps` steps of each candidate -- "rank0",
              "halfsteps" and, when the replicas connect and its chain verifies, the peer-write dataflow -- from the
              same state (which is restored: the random stream is counter based) and keeps the fastest by the
              slowest rank's clock.  Every rank takes the same decision (the timings are gathered; the choice is
              a function of the gathered list).  `schedule_choice`, `schedule_reason` and `auto_probe` say what was
              chosen and why [/root/reference/emcee/emcee_radex.py:480-488: the pool never makes a run slower
              than processes=1 by more than its IPC]."""

    AUTO_RANK0_TASKS = 1536     # proposals per half-step up to which one GPU runs one wavefront per SIMD (rx_api.hip: 6 x 256 CUs)

    def __init__(self, nwalkers, ndim, engine=None, log_prob_fn=None, nens=1, ens_src=None, a=2.0, seed=0,
                 group=None, sharded=None, schedule=None, verify_peer_steps=8):
        import torch
        if nwalkers < 2 * ndim:
            raise ValueError("The number of walkers needs to be at least twice the dimension "
                             "of your parameter space")
        if nwalkers % 2:
            raise ValueError("The number of walkers must be even")
        if (engine is None) == (log_prob_fn is None):
            raise ValueError("give exactly one of engine / log_prob_fn")
        if engine is not None and int(ndim) not in (4, 8):
            raise ValueError("the engine evaluates 1- or 2-component models: ndim must be 4 or 8")
        self.nwalkers, self.ndim, self.nens = int(nwalkers), int(ndim), int(nens)
        self.a, self.seed = float(a), int(seed)
        self.engine = engine
        self.backend = _EngineStretchBackend(engine) if engine is not None else _HostStretchBackend(log_prob_fn)
        dev = self.backend.device
        N, nq = self.nens * self.nwalkers, self.nens * (self.nwalkers // 2)
        self.N, self.nq = N, nq
        self.coords = torch.zeros(N, ndim, dtype=torch.float64, device=dev)
        self.lnp = torch.zeros(N, dtype=torch.float64, device=dev)
        self.naccept = torch.zeros(N, dtype=torch.int32, device=dev)
        self.q = torch.empty(nq, ndim, dtype=torch.float64, device=dev)
        self.factor = torch.empty(nq, dtype=torch.float64, device=dev)
        self.widx = torch.empty(nq, dtype=torch.int32, device=dev)
        self.qstatus = torch.empty(nq, dtype=torch.int32, device=dev)
        self.qniter = torch.empty(nq, dtype=torch.int32, device=dev)
        self.ens_src = self.qsrc = None
        if ens_src is not None:
            self.ens_src = torch.as_tensor(np.asarray(ens_src, dtype=np.int32)).to(dev)
            if self.ens_src.numel() != self.nens:
                raise ValueError("ens_src needs one source slot per ensemble")
            self.qsrc = torch.empty(nq, dtype=torch.int32, device=dev)
        self.group = group
        self.world = 1
        self._sharded = bool(group is not None or sharded)   # the collective path, even on a group of one rank
        if group is not None or sharded:
            import torch.distributed as dist
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
            self._dist = dist
        lo, hi, per = block_partition(nq, self.world, getattr(self, "rank", 0))
        self._lo, self._hi, self._per = lo, hi, per
        self.lnp_q = torch.empty(per * self.world, dtype=torch.float64, device=dev)
        self._mine = torch.empty(per, dtype=torch.float64, device=dev)
        # schedule (single GPU, engine backend): "dataflow" = one persistent kernel, every proposal starts
        # as soon as the two walkers it reads are final (rx_sampler_run_async_device); "halfsteps" = propose /
        # solve / accept launches per half-step (rx_sampler_run_device).  The chains are bit-identical.
        if schedule is None:              # with a group: never slower than one GPU (see the class docstring)
            schedule = "auto" if self._sharded else "dataflow"
        if schedule not in ("dataflow", "halfsteps", "auto", "rank0"):
            raise ValueError("schedule must be 'dataflow', 'halfsteps', 'auto' or 'rank0'")
        if schedule in ("auto", "rank0") and not self._sharded:
            schedule = "dataflow"         # one process: nothing to choose
        self.requested_schedule = schedule
        self.schedule_choice = None       # "auto": what the rule or the probe chose ("rank0", "halfsteps", "dataflow-peer")
        self.auto_probe = None            # "auto": seconds per candidate over verify_peer_steps steps (the slowest rank's)
        self.broadcast_chain = True       # "rank0": the stored chain goes to every rank (emcee's API on every rank)
        self.schedule = schedule
        self.verify_peer_steps = int(verify_peer_steps)
        self.peer_verified = None         # None: not checked (yet); True / False: the first peer run against the half-step schedule
        self.peer_verify_detail = None
        self.schedule_reason = None       # why the last run_mcmc used the schedule it used
        self.peer_state = None            # None: not tried yet; True: replicas connected; str: why not (halfsteps then)
        self.last_schedule = None         # what the last run_mcmc actually used
        self.fallback = True              # dataflow run abandoned (timeout) -> repeat it per half-step
        self.step_counter = 0
        self.time_solves = False          # benchmarks: sum the solve-kernel time of every half-step (HIP events)
        self.last_solve_ms = None
        self.reset()

    # --- bookkeeping (emcee API) ------------------------------------------------------------------
    def reset(self):
        self._chain, self._chain_lnp = [], []
        self.naccept.zero_()
        self.iteration = 0

    @property
    def acceptance_fraction(self):
        v = self.naccept.cpu().numpy().astype(np.float64) / float(max(self.iteration, 1))
        return v if self.nens == 1 else v.reshape(self.nens, self.nwalkers)

    def _cat(self, parts, tail):
        if not parts:
            return np.empty((0,) + tail)
        return parts[0] if len(parts) == 1 else np.concatenate(parts)

    def get_chain(self, flat=False, thin=1, discard=0):
        v = self._cat(self._chain, (self.N, self.ndim))[discard::thin]
        if self.nens > 1:
            v = v.reshape(len(v), self.nens, self.nwalkers, self.ndim)
            return v.transpose(1, 0, 2, 3).reshape(self.nens, -1, self.ndim) if flat else v
        return v.reshape(-1, self.ndim) if flat else v

    def get_log_prob(self, flat=False, thin=1, discard=0):
        v = self._cat(self._chain_lnp, (self.N,))[discard::thin]
        if self.nens > 1:
            v = v.reshape(len(v), self.nens, self.nwalkers)
            return v.transpose(1, 0, 2).reshape(self.nens, -1) if flat else v
        return v.reshape(-1) if flat else v

    # --- evaluation of one batch of proposals (sharded over the group when there is one) --------------
    def _evaluate(self, q, qsrc):
        if not self._sharded:
            out = self.lnp_q[:q.shape[0]]
            self.backend.evaluate(self, q, out, qsrc)
            return out
        lo, hi = self._lo, self._hi
        self._mine.fill_(float("-inf"))
        if hi > lo:
            self.backend.evaluate(self, q[lo:hi], self._mine[:hi - lo], None if qsrc is None else qsrc[lo:hi])
        if self._mine.is_cuda and self._dist.get_backend(self.group) == "gloo":
            # rehearsal of the control flow on a box whose ranks share one GPU: gloo moves host copies
            import torch
            out = torch.empty(self.lnp_q.shape, dtype=torch.float64)
            self._dist.all_gather_into_tensor(out, self._mine.cpu(), group=self.group)
            self.lnp_q.copy_(out)
            return self.lnp_q[:q.shape[0]]
        try:
            self._dist.all_gather_into_tensor(self.lnp_q, self._mine, group=self.group)
        except (RuntimeError, NotImplementedError):
            self._dist.all_gather(list(self.lnp_q.view(self.world, -1).unbind(0)), self._mine, group=self.group)
        return self.lnp_q[:q.shape[0]]

    def compute_log_prob(self, coords):
        """log-probabilities of [N, ndim] positions (ensemble-major), on the device."""
        import torch
        X = torch.as_tensor(np.ascontiguousarray(coords, dtype=np.float64).reshape(self.N, self.ndim)).to(self.coords.device)
        src = None if self.ens_src is None else self.ens_src.repeat_interleave(self.nwalkers).contiguous()
        out = torch.empty(self.N, dtype=torch.float64, device=X.device)
        st = torch.empty(self.N, dtype=torch.int32, device=X.device)
        if self.engine is not None:
            self.engine.lnprob_batch_torch(X, out, st, torch.empty_like(st), src_index=src)
        else:
            self.backend.evaluate(self, X, out, src)
        return out

    # --- the dataflow schedule across the ranks of the group (peer writes into IPC-mapped replicas) --------
    def _gather_objects(self, obj):
        out = [None] * self.world
        if self.world == 1:
            return [obj]
        self._dist.all_gather_object(out, obj, group=self.group)
        return out

    def _barrier(self):
        if self.world > 1:
            self._dist.barrier(group=self.group)

    def _peer_teardown(self):
        """Collective: every rank unmaps its peers' replicas, THEN (behind a barrier) frees its own block -- a block must not be
        freed, nor its successor exported, while a peer still has it mapped."""
        from .engine import EngineError
        for step in (self.engine.sampler_peer_disconnect, self.engine.sampler_peer_close):
            try:
                step()
            except EngineError:
                pass
            self._gather_objects(None)                 # (= barrier, on whatever transport the group has)

    def _peer_setup(self):
        """Allocates this rank's replica, exchanges the IPC handles, maps the peers'.  Collective: every rank
        learns whether ALL ranks succeeded; otherwise all fall back to the half-step schedule together."""
        import os
        import socket
        from .engine import EngineError
        eng, err, handle = self.engine, None, None
        self._peer_teardown()                          # what an earlier sampler left on this handle (collective like the rest)
        try:
            handle = eng.sampler_peer_setup(self.world, self.rank, self.nens, self.nwalkers, self.ndim // 4)
        except EngineError as exc:
            err = str(exc)
        try:
            bus = eng.bus_id()                         # the GPU's name that means the same in every process of the node
        except EngineError:
            bus = None
        me = (socket.gethostname(), os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"),
              os.environ.get("CUDA_VISIBLE_DEVICES"), eng.device, bus)
        infos = self._gather_objects((handle, err, me))
        bad = [i for i, (_, e, _) in enumerate(infos) if e]
        if not bad and err is None:
            # A kernel that stores into memory of a GPU it has no peer access to FAULTS (and takes the process down): where this
            # rank can name the peers' devices -- same host, same *_VISIBLE_DEVICES, so that ordinals mean the same thing --
            # hipDeviceCanAccessPeer must say yes for every one of them, or the peer-write path is not taken at all
            from .engine import peer_topology
            for r, (_, _, m) in enumerate(infos):
                if r != self.rank and m[:4] == me[:4] and m[4] != me[4]:
                    t = peer_topology(me[4], m[4])
                    if t is None or t[0] != 1:
                        err = "no peer access from device %d to device %d (rank %d)" % (me[4], m[4], r)
                        break
            errs = self._gather_objects(err)
            bad = [i for i, e in enumerate(errs) if e]
            if bad:
                infos = [(h, errs[i], m) for i, (h, _, m) in enumerate(infos)]
        if not bad:
            # ranks that share one GPU (rehearsals on a one-GPU box) must all be resident at once: split the CUs
            # (the library counts the replicas that live on its own device itself -- rx_sampler_peer_same_device -- and
            # splits the compute units; an explicit limit of an earlier set-up must not outlive it)
            eng.set_sampler_grid_limit(0)
            try:
                ids = [m[5] for (_, _, m) in infos]
                eng.sampler_peer_connect(ipc_handles=[h for (h, _, _) in infos], bus_ids=ids if all(ids) else None)
            except EngineError as exc:
                err = str(exc)
            errs = self._gather_objects(err)
            bad = [i for i, e in enumerate(errs) if e]
            if bad:
                err = "rank %d: %s" % (bad[0], errs[bad[0]])
        else:
            err = "rank %d: %s" % (bad[0], infos[bad[0]][1])
        if bad:
            self._peer_teardown()
            self.peer_state = err
            if self.rank == 0:
                import warnings
                warnings.warn("multi-GPU dataflow sampler unavailable (%s): half-steps + all_gather instead" % err)
            return False
        self.peer_state = True
        eng._peer_owner = self            # one replica block per handle: another sampler's set-up replaces it
        return True

    def _agree(self, err):
        """Collective in place of a barrier: every rank contributes None or what went wrong on it and learns about all
        of them, so that one rank's failure raises (or falls back) on ALL ranks instead of leaving the others in a
        barrier until the process-group timeout.  Returns [(rank, message)] of the ranks that failed."""
        errs = self._gather_objects(err)
        return [(r, e) for r, e in enumerate(errs) if e]

    def _run_peer(self, nsteps, chain, chain_lnp):
        """One run_mcmc call under the peer-write dataflow schedule.  Returns False when a task on some rank gave
        up waiting (every rank sees it): the caller repeats the run per half-step.  An error on one rank (HIP,
        state) is raised on EVERY rank."""
        import os
        from .engine import EngineError, RX_E_TIMEOUT
        eng, dev = self.engine, self.coords.device
        err = None
        try:
            eng.sampler_peer_begin(self.coords, self.lnp, self.naccept)
        except EngineError as exc:
            err = "sampler_peer_begin: %s" % exc
        bad = self._agree(err)                        # (= barrier) every replica is seeded: peers may write into it
        if bad:
            raise EngineError("multi-GPU dataflow sampler, rank %d: %s" % bad[0])
        err, timed_out = None, False
        try:
            inject = getattr(self, "_inject", None)              # (set by tests/test_gpu_peer.py's worker only: never read from the environment)
            if inject == ("error", self.rank):                   # a launch failure on one rank
                raise EngineError("injected launch failure (test)", -5)
            if inject != ("stall", self.rank):                   # ("stall": a rank whose kernel never starts)
                eng.sampler_peer_run(self.a, self.seed, self.step_counter, nsteps, dev, chain, chain_lnp, ens_src=self.ens_src)
        except EngineError as exc:
            err = "sampler_peer_run: %s" % exc
            try:
                eng.sampler_peer_abort()              # the peers' kernels wait for this rank's tasks: end them now
            except EngineError:
                pass
        if err is None:
            try:
                eng.sampler_wait(dev)
            except EngineError as exc:                # (the one-GPU flag; the replicas' abort word is read by finish)
                if exc.rc == RX_E_TIMEOUT:
                    timed_out = True
                else:
                    err = "sampler_wait: %s" % exc
        bad = self._agree(err)                        # (= barrier) every peer has finished: this rank's replica is complete
        if bad:
            raise EngineError("multi-GPU dataflow sampler, rank %d: %s" % bad[0])
        try:
            eng.sampler_peer_finish(self.coords, self.lnp, self.naccept)
        except EngineError as exc:
            if exc.rc != RX_E_TIMEOUT:
                err = "sampler_peer_finish: %s" % exc
            timed_out = True
        flags = self._gather_objects((err, timed_out))
        bad = [(r, e) for r, (e, _) in enumerate(flags) if e]
        if bad:
            raise EngineError("multi-GPU dataflow sampler, rank %d: %s" % bad[0])
        if any(t for (_, t) in flags):
            return False
        if chain is not None and self.world > 1:      # each rank wrote the rows of the walkers it updated (others zero)
            for t in (chain, chain_lnp):
                if self._dist.get_backend(self.group) == "gloo":
                    h = t.cpu()
                    self._dist.all_reduce(h, group=self.group)
                    t.copy_(h)
                else:
                    self._dist.all_reduce(t, group=self.group)
        return True

    def _halfsteps(self, nsteps, chain, chain_lnp):
        """nsteps steps, one propose / evaluate (sharded when there is a group) / accept round per half-step."""
        for s in range(nsteps):
            step = self.step_counter + s
            for split in range(2):
                self.backend.propose(self, step, split)
                lnp_q = self._evaluate(self.q, self.qsrc)
                self.backend.accept(self, step, split, lnp_q)
            if chain is not None:
                chain[s].copy_(self.coords)
                chain_lnp[s].copy_(self.lnp)

    def _verify_peer(self):
        """The peer-write path has never crossed a device boundary in a test: before a sampler
        relies on it, the first steps are advanced under BOTH schedules from the same state and compared bit for bit,
        on every rank.  Collective; returns True when every rank saw identical states.  The sampler's own state and
        step counter are left as they were (the random stream is counter based: nothing is consumed)."""
        import hashlib
        import torch
        k = self.verify_peer_steps
        start = (self.coords.clone(), self.lnp.clone(), self.naccept.clone())
        ok_run = self._run_peer(k, None, None)
        a = tuple(t.clone() for t in (self.coords, self.lnp, self.naccept))
        for dst, src in zip((self.coords, self.lnp, self.naccept), start):
            dst.copy_(src)
        self._halfsteps(k, None, None)
        b = (self.coords, self.lnp, self.naccept)
        bits = lambda t: t.view(torch.int64) if t.dtype == torch.float64 else t          # bit for bit, NaN or not
        same = bool(ok_run) and all(torch.equal(bits(x), bits(y)) for x, y in zip(a, b))
        digest = hashlib.sha1(b"".join(t.cpu().numpy().tobytes() for t in b)).hexdigest()[:16]
        for dst, src in zip((self.coords, self.lnp, self.naccept), start):
            dst.copy_(src)
        res = self._gather_objects((same, bool(ok_run), digest))
        allsame = all(r[0] for r in res) and len({r[2] for r in res}) == 1
        self.peer_verified = bool(allsame)
        self.peer_verify_detail = {"steps": k, "ranks_identical": [bool(r[0]) for r in res], "peer_run_completed": [bool(r[1]) for r in res],
                                   "halfstep_state_sha1": [r[2] for r in res],
                                   "outcome": "identical" if allsame else
                                              ("abandoned by the watchdog on rank(s) %s (nothing was compared there)"
                                               % [i for i, r in enumerate(res) if not r[1]] if not all(r[1] for r in res) else "differed")}
        return self.peer_verified

    # --- "rank0": the ensemble on one GPU, one broadcast per call -----------------------------------------
    def _bcast(self, t):
        if t is None or not self._sharded:
            return                      # (a group of ONE rank still makes the call: the RCCL path of the one-rank GPU test)
        if t.is_cuda and self._dist.get_backend(self.group) == "gloo":     # (rehearsals: ranks that share one GPU)
            h = t.cpu()
            self._dist.broadcast(h, src=self._dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            if self.rank != 0:
                t.copy_(h)
        else:
            self._dist.broadcast(t, src=self._dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)

    def _unsharded(self, nsteps, chain, chain_lnp):
        """nsteps steps on THIS rank alone, with the schedule a sampler without a group uses."""
        if self.engine is not None and not self.time_solves:
            from .engine import EngineError, RX_E_TIMEOUT
            start = (self.coords.clone(), self.lnp.clone(), self.naccept.clone())
            try:
                self.engine.sampler_run_async_torch(self.nens, self.nwalkers, self.ndim // 4, self.a, self.seed,
                                                    self.step_counter, nsteps, self.coords, self.lnp, self.naccept,
                                                    chain, chain_lnp, ens_src=self.ens_src)
                self.engine.sampler_wait(self.coords.device)
            except EngineError as exc:
                if exc.rc != RX_E_TIMEOUT or not self.fallback:
                    raise
                # (as without a group: the dataflow kernel's own give-up is repeated per half-step from the state the call began with)
                for dst, src in zip((self.coords, self.lnp, self.naccept), start):
                    dst.copy_(src)
                self.engine.sampler_run_torch(self.nens, self.nwalkers, self.ndim // 4, self.a, self.seed, self.step_counter, nsteps,
                                              self.coords, self.lnp, self.naccept, chain, chain_lnp, ens_src=self.ens_src,
                                              time_solves=False)
        elif self.engine is not None:
            self.last_solve_ms = self.engine.sampler_run_torch(
                self.nens, self.nwalkers, self.ndim // 4, self.a, self.seed, self.step_counter, nsteps,
                self.coords, self.lnp, self.naccept, chain, chain_lnp, ens_src=self.ens_src, time_solves=True)
        else:
            sharded, self._sharded = self._sharded, False
            try:
                self._halfsteps(nsteps, chain, chain_lnp)
            finally:
                self._sharded = sharded

    def _run_rank0(self, nsteps, chain, chain_lnp, broadcast=True):
        """One run_mcmc call under "rank0": rank 0 advances the ensemble, then ONE broadcast hands the state to the others --
        a status word, positions, log-probabilities and acceptance counts packed into one buffer (a collective per tensor, or an
        object gather for the status, costs more than the 40 KB they move); the stored chain follows in a second one.  An error on
        rank 0 travels in the status word and is raised on EVERY rank."""
        import torch
        from .engine import EngineError
        N, nd = self.N, self.ndim
        pack = getattr(self, "_pack", None)
        if pack is None or pack.numel() != 1 + N * nd + 2 * N:
            pack = self._pack = torch.zeros(1 + N * nd + 2 * N, dtype=torch.float64, device=self.coords.device)
        err = None
        if self.rank == 0:
            try:
                self._unsharded(nsteps, chain, chain_lnp)
            except EngineError as exc:
                err = exc
            pack[0] = 0.0 if err is None else 1.0
            pack[1:1 + N * nd].copy_(self.coords.reshape(-1))
            pack[1 + N * nd:1 + N * nd + N].copy_(self.lnp)
            pack[1 + N * nd + N:].copy_(self.naccept)           # (int32 counts are exact in a double)
        if broadcast or err is not None:
            self._bcast(pack)
        if self.rank == 0 and err is not None:
            raise err
        if broadcast:
            if float(pack[0]) != 0.0:
                raise EngineError("sampler (schedule rank0): the run failed on rank 0 (its message is on rank 0)")
            if self.rank != 0:
                self.coords.copy_(pack[1:1 + N * nd].reshape(N, nd))
                self.lnp.copy_(pack[1 + N * nd:1 + N * nd + N])
                self.naccept.copy_(pack[1 + N * nd + N:].to(torch.int32))
            if self.broadcast_chain:
                self._bcast(chain)
                self._bcast(chain_lnp)

    def _timed(self, fn, k):
        """Seconds this rank needs for k steps of fn from the present state, which is restored."""
        import time
        import torch
        start = (self.coords.clone(), self.lnp.clone(), self.naccept.clone())
        self._barrier()
        if self.coords.is_cuda:
            torch.cuda.synchronize(self.coords.device)
        t0 = time.perf_counter()
        ok = fn(k)
        if self.coords.is_cuda:
            torch.cuda.synchronize(self.coords.device)
        self._barrier()
        dt = time.perf_counter() - t0
        for dst, src in zip((self.coords, self.lnp, self.naccept), start):
            dst.copy_(src)
        return dt if ok is not False else float("inf")

    def _choose(self, nsteps):
        """"auto", collective, once per sampler: the rule, or the timed probe (class docstring).  Sets schedule_choice."""
        import torch
        tasks = self.nq
        if tasks <= self.AUTO_RANK0_TASKS or self.world == 1:
            self.schedule_choice = "rank0"
            self.schedule_reason = ("auto, by rule: %d proposals per half-step <= %d (one GPU's latency regime: sharding adds a "
                                    "collective per half-step and removes nothing)" % (tasks, self.AUTO_RANK0_TASKS))
            return
        k = max(self.verify_peer_steps, 1)
        cand = {}
        # the peer-write schedule is a candidate only where the replicas connect AND its chain verifies
        peer = (self.engine is not None and not self.time_solves
                and (self.peer_state is True or (self.peer_state is None and self._peer_setup())))
        if peer and self.peer_verified is None and self.verify_peer_steps > 0:
            if not self._verify_peer():
                self.peer_state = "the verification of the peer-write chain against the half-step chain over the first %d steps: %s" % (
                    self.verify_peer_steps, self.peer_verify_detail["outcome"])
                self._peer_teardown()
                peer = False
        # (one untimed pass first: code objects, RCCL channels and IPC mappings come into being on first use)
        for timed in (False, True):
            t = {"halfsteps": self._timed(lambda n: self._halfsteps(n, None, None), k),
                 "rank0": self._timed(lambda n: self._run_rank0(n, None, None), k)}
            if peer:
                t["dataflow-peer"] = self._timed(lambda n: self._run_peer(n, None, None), k)
            cand = t
        # the slowest rank's clock decides, identically everywhere
        allt = self._gather_objects(cand)
        worst = {name: max(float(r.get(name, float("inf"))) for r in allt) for name in cand}
        # (ties and near-ties go to the schedule with the least machinery: rank0, then halfsteps)
        order = sorted(worst, key=lambda n: (worst[n] * (1.0 if n == "rank0" else 1.03 if n == "halfsteps" else 1.06)))
        self.schedule_choice = order[0]
        self.auto_probe = {"steps": k, "seconds": worst, "peer_candidate": bool(peer),
                           "peer_state": None if self.peer_state in (None, True) else str(self.peer_state)}
        self.schedule_reason = "auto, by probe over %d steps: %s" % (k, ", ".join("%s %.3g ms/step" % (n, 1e3 * worst[n] / k) for n in order))
        if self.schedule_choice != "dataflow-peer" and peer:
            self._peer_teardown()
            self.peer_state = None

    # --- sampling ---------------------------------------------------------------------------------------
    def run_mcmc(self, initial_state, nsteps, progress=False, store=True):
        import torch
        from .engine import EngineError, RX_E_TIMEOUT
        if isinstance(initial_state, State):
            coords, lp = initial_state.coords, initial_state.log_prob
        else:
            coords, lp = np.asarray(initial_state, dtype=np.float64), None
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        if coords.size != self.N * self.ndim:
            raise ValueError("incompatible input dimensions")
        if np.any(np.isinf(coords)):
            raise ValueError("At least one parameter value was infinite")
        if np.any(np.isnan(coords)):
            raise ValueError("At least one parameter value was NaN")
        self.coords.copy_(torch.from_numpy(coords.reshape(self.N, self.ndim)))
        if lp is None:
            self.lnp.copy_(self.compute_log_prob(coords))
        else:
            self.lnp.copy_(torch.from_numpy(np.ascontiguousarray(lp, dtype=np.float64).reshape(self.N)))
        if bool(torch.isnan(self.lnp).any()):
            raise ValueError("The initial log_prob was NaN")
        nsteps = int(nsteps)
        chain = chain_lnp = None
        if self.peer_state is True and getattr(self.engine, "_peer_owner", None) is not self:
            self.peer_state = None        # the handle's replica block now belongs to another sampler: set up again
        if self._sharded and self.requested_schedule in ("auto", "rank0") and nsteps > 0:
            if self.requested_schedule == "rank0":
                self.schedule_choice = "rank0"
            elif self.schedule_choice is None:
                self._choose(nsteps)
            auto_reason = self.schedule_reason if self.requested_schedule == "auto" else "requested"
            if self.schedule_choice == "rank0":
                if store:
                    chain = torch.zeros(nsteps, self.N, self.ndim, dtype=torch.float64, device=self.coords.device)
                    chain_lnp = torch.zeros(nsteps, self.N, dtype=torch.float64, device=self.coords.device)
                self._run_rank0(nsteps, chain, chain_lnp)
                self.last_schedule, self.schedule_reason = "rank0", auto_reason
                return self._finish_run(nsteps, chain, chain_lnp)
            self.schedule = "dataflow" if self.schedule_choice == "dataflow-peer" else "halfsteps"
        else:
            auto_reason = None
        peer = (self.engine is not None and self._sharded and self.schedule == "dataflow" and not self.time_solves
                and nsteps > 0 and (self.peer_state is True or (self.peer_state is None and self._peer_setup())))
        if store and nsteps > 0:
            # (peer schedule: every rank fills the rows of the walkers it updated, the sum over ranks is the chain)
            alloc = torch.zeros if peer else torch.empty
            chain = alloc(nsteps, self.N, self.ndim, dtype=torch.float64, device=self.coords.device)
            chain_lnp = alloc(nsteps, self.N, dtype=torch.float64, device=self.coords.device)
        self.last_schedule = "halfsteps"
        self.schedule_reason = ("requested" if self.schedule == "halfsteps" else
                                "time_solves" if self.time_solves else
                                ("peer replicas unavailable: %s" % self.peer_state) if isinstance(self.peer_state, str) else None)
        if peer and self.peer_verified is None and self.verify_peer_steps > 0 and self.world > 1:
            if not self._verify_peer():
                self.peer_state = "the verification of the peer-write chain against the half-step chain over the first %d steps: %s (%s)" % (
                    self.verify_peer_steps, self.peer_verify_detail["outcome"], self.peer_verify_detail)
                self.schedule_reason = self.peer_state
                if self.rank == 0:
                    import warnings
                    warnings.warn("multi-GPU dataflow sampler: %s; half-steps + all_gather instead" % self.peer_state)
                self._peer_teardown()
                peer = False
                if chain is not None:
                    chain = torch.empty_like(chain)
                    chain_lnp = torch.empty_like(chain_lnp)
        if peer:
            start = (self.coords.clone(), self.lnp.clone(), self.naccept.clone())
            if self._run_peer(nsteps, chain, chain_lnp):
                self.last_schedule = "dataflow-peer"
                self.schedule_reason = "requested" + ("" if self.peer_verified is None else
                                                      "; first %d steps identical to the half-step schedule on every rank" % self.verify_peer_steps)
            else:
                self.schedule_reason = "the peer-write run was abandoned (watchdog on some rank): repeated per half-step"
                if not self.fallback:
                    raise EngineError("multi-GPU dataflow sampler: a task waited longer than the timeout", RX_E_TIMEOUT)
                import warnings
                warnings.warn("dataflow sampler abandoned its run (timeout on some rank); repeating it per half-step")
                self.coords.copy_(start[0]); self.lnp.copy_(start[1]); self.naccept.copy_(start[2])
                peer = False
        dataflow = (self.engine is not None and not self._sharded and self.schedule == "dataflow" and not self.time_solves)
        if peer:
            pass
        elif dataflow:
            start = (self.coords.clone(), self.lnp.clone(), self.naccept.clone())
            try:
                self.engine.sampler_run_async_torch(self.nens, self.nwalkers, self.ndim // 4, self.a, self.seed,
                                                    self.step_counter, nsteps, self.coords, self.lnp, self.naccept,
                                                    chain, chain_lnp, ens_src=self.ens_src)
                self.engine.sampler_wait(self.coords.device)
            except EngineError as exc:
                # only the sampler's own give-up (RX_E_TIMEOUT: a task waited longer than the timeout, the
                # grid drained) is retried; argument errors, RX_E_UNSUPP and HIP faults are the caller's
                if exc.rc != RX_E_TIMEOUT or not self.fallback:
                    raise
                # (never observed in practice): the same chain under the half-step schedule, from the
                # state this call started with
                import warnings
                warnings.warn("dataflow sampler abandoned its run (%s); repeating it per half-step" % exc)
                self.schedule_reason = "the dataflow run was abandoned (%s): repeated per half-step" % exc
                self.coords.copy_(start[0]); self.lnp.copy_(start[1]); self.naccept.copy_(start[2])
                dataflow = False
        if peer:
            pass
        elif dataflow:
            self.last_schedule = "dataflow"
        elif self.engine is not None and not self._sharded:
            # one call enqueues every kernel of every step on the current stream
            self.last_solve_ms = self.engine.sampler_run_torch(
                self.nens, self.nwalkers, self.ndim // 4, self.a, self.seed, self.step_counter, nsteps,
                self.coords, self.lnp, self.naccept, chain, chain_lnp, ens_src=self.ens_src,
                time_solves=self.time_solves)
        else:
            self._halfsteps(nsteps, chain, chain_lnp)
        if self.schedule_reason is None:
            self.schedule_reason = "requested"
        if auto_reason is not None and self.schedule_reason.startswith("requested"):
            self.schedule_reason = auto_reason
        return self._finish_run(nsteps, chain, chain_lnp)

    def _finish_run(self, nsteps, chain, chain_lnp):
        self.step_counter += nsteps
        self.iteration += nsteps
        if chain is not None:
            # finished blocks leave HBM (65536 walkers x 1000 steps are 2 GB per call): get_chain joins them on the host
            self._chain.append(chain.cpu().numpy())
            self._chain_lnp.append(chain_lnp.cpu().numpy())
        shape = (self.N, self.ndim) if self.nens == 1 else (self.nens, self.nwalkers, self.ndim)
        lshape = (self.N,) if self.nens == 1 else (self.nens, self.nwalkers)
        return State(self.coords.cpu().numpy().reshape(shape), self.lnp.cpu().numpy().reshape(lshape))
