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

Multi-GPU: `ShardedLogProb` splits that batch into contiguous blocks, one per rank
(one process per GPU), evaluates its block locally and all-gathers the log-probabilities
(torch.distributed: "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests).  Positions and
the random stream are replicated (same seed on every rank), so proposals need no exchange
and every rank performs the identical accept/reject.
"""
from __future__ import annotations

import numpy as np


class State:
    def __init__(self, coords, log_prob=None, random_state=None):
        self.coords = np.array(coords, dtype=np.float64, copy=True)
        self.log_prob = None if log_prob is None else np.array(log_prob, dtype=np.float64, copy=True)
        self.random_state = random_state

    def __iter__(self):
        return iter((self.coords, self.log_prob, self.random_state))


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, log_prob_fn, args=None, kwargs=None, pool=None,
                 vectorize=False, a=2.0, seed=None):
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

    def _step(self, state):
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


class ShardedLogProb:
    """Evaluate a [N, ndim] batch across the ranks of a torch.distributed group.

    Rank r evaluates rows [r*ceil(N/G), ...) with `local_fn` (its own GPU), then ONE
    all_gather of float64 log-probabilities makes the full vector available everywhere
    (N/(2G) doubles per rank per half-step; latency-bound)."""

    def __init__(self, local_fn, group=None, device=None):
        import torch.distributed as dist
        self.local_fn = local_fn
        self.group = group
        self.dist = dist
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device

    def partition(self, N):
        per = -(-N // self.world)
        lo = min(self.rank * per, N)
        return lo, min(lo + per, N), per

    def __call__(self, P):
        import torch
        P = np.ascontiguousarray(P, dtype=np.float64)
        N = len(P)
        lo, hi, per = self.partition(N)
        local = np.full(per, -np.inf)
        if hi > lo:
            local[:hi - lo] = np.asarray(self.local_fn(P[lo:hi]), dtype=np.float64)
        dev = self.device if self.device is not None else "cpu"
        mine = torch.from_numpy(local).to(dev)
        out = torch.empty(per * self.world, dtype=torch.float64, device=dev)
        try:
            self.dist.all_gather_into_tensor(out, mine, group=self.group)
        except (RuntimeError, NotImplementedError):      # backend without the flat form
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine, group=self.group)
            out = torch.cat(parts)
        # blocks are contiguous and every rank but the last non-empty one is full
        return out.cpu().numpy()[:N]
