"""Engine: thin Python owner of one rx_handle (one GPU).

Mirrors what the reference keeps per pool worker -- one `pyradex.Radex` object
created by `init_radex` [/root/reference/emcee/emcee_radex.py:104-117] plus the
closure arguments of `lnprob` -- but evaluates whole walker batches per call.
All numerics happen in libradex_emcee_amd.so (HIP, gfx950).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .molecule import default_molfile

METHODS = {"sphere": 1, "lvg": 2, "slab": 3}      # core.py:690-700


class EngineError(RuntimeError):
    """A C-ABI call failed; `rc` is its RX_E_* return code."""

    def __init__(self, msg, rc=None):
        super().__init__(msg)
        self.rc = rc


RX_E_TIMEOUT = -7        # include/radex_emcee_amd.h: a task of the dataflow sampler gave up waiting


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def peer_topology(dev_a, dev_b):
    """(can_access_peer, link_type, hops) between two HIP devices (rx_peer_topology): 1 / 0 / -1, -1 = unknown."""
    out = (C.c_int32 * 3)()
    rc = _lib.load().rx_peer_topology(int(dev_a), int(dev_b), out)
    if rc:
        return None
    return int(out[0]), int(out[1]), int(out[2])


class Engine:
    def __init__(self, molfile: str | None = None, species: str = "co", escapeProbGeom: str = "lvg",
                 deltav: float = 1.0, device: int = 0):
        self._L = _lib.load()
        self._h = None
        if escapeProbGeom not in METHODS:
            raise ValueError("Invalid escapeProbGeom, must be one of " + ",".join(METHODS))
        self.molfile = molfile or default_molfile(species)
        err = C.create_string_buffer(512)
        self._h = self._L.rx_create(self.molfile.encode(), METHODS[escapeProbGeom], float(deltav),
                                    int(device), err, 512)
        if not self._h:
            raise EngineError("rx_create failed: " + err.value.decode("utf-8", "replace"))
        self.device = int(device)
        self.nlev = self._L.rx_nlev(self._h)
        self.nline = self._L.rx_nline(self._h)
        self.npart = self._L.rx_npart(self._h)
        ids = np.zeros(self.npart, dtype=np.int32)
        self._L.rx_partner_ids(self._h, _ip(ids))
        self.partner_ids = [int(x) for x in ids]
        self.xnu = np.zeros(self.nline)
        self.spfreq = np.zeros(self.nline)
        self.iupp = np.zeros(self.nline, dtype=np.int32)
        self.ilow = np.zeros(self.nline, dtype=np.int32)
        self._L.rx_line_data(self._h, _dp(self.xnu), _dp(self.spfreq), _ip(self.iupp), _ip(self.ilow))
        self._sources = {}

    # -- lifetime ---------------------------------------------------------------
    def close(self):
        if self._h:
            self._L.rx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise EngineError("%s failed (rc=%d): %s" % (what, rc, self._L.rx_last_error(self._h).decode()), rc)

    @property
    def kernel_name(self) -> str:
        return self._L.rx_kernel_name(self._h).decode()

    # -- configuration -----------------------------------------------------------
    def set_fortho(self, fortho: float):
        self._chk(self._L.rx_set_fortho(self._h, float(fortho)), "rx_set_fortho")

    def set_iteration_limits(self, miniter=10, maxiter=200):
        self._chk(self._L.rx_set_iteration_limits(self._h, int(miniter), int(maxiter)), "rx_set_iteration_limits")

    def set_source(self, tbg, Jup=(), flux=(), eflux=(), bounds=None, ncomp=1, T_d=None, src=0):
        Jup = np.ascontiguousarray(Jup, dtype=np.int32)
        flux = np.ascontiguousarray(flux, dtype=np.float64)
        eflux = np.ascontiguousarray(eflux, dtype=np.float64)
        if bounds is None:
            bounds = np.tile(np.array([[-np.inf, np.inf]]), (4 * ncomp, 1))
        bounds = np.ascontiguousarray(bounds, dtype=np.float64).reshape(4 * ncomp, 2)
        if not (len(Jup) == len(flux) == len(eflux)):
            raise ValueError("Jup, flux, eflux must have equal lengths")
        td = float("nan") if T_d is None else float(T_d)
        rc = self._L.rx_set_source(self._h, int(src), float(tbg), len(Jup), _ip(Jup), _dp(flux), _dp(eflux),
                                   _dp(bounds), int(ncomp), td)
        self._chk(rc, "rx_set_source")
        self._sources[int(src)] = dict(nJ=len(Jup), ncomp=int(ncomp))

    def background(self, src=0):
        """backrad_'s outputs for a source slot as the kernels hold them (rx_background): (backi[nline] = totalb, trj = tbg)
        [/root/reference/emcee/pyradex/core.py:845-854]."""
        backi = np.zeros(self.nline)
        tbg = C.c_double(0.0)
        self._chk(self._L.rx_background(self._h, int(src), _dp(backi), C.byref(tbg)), "rx_background")
        return backi, tbg.value

    def set_source_prior(self, src=0, enabled=True):
        """enabled=False: lnprob_batch returns the log-likelihood alone for this slot (rx_set_source_prior)."""
        self._chk(self._L.rx_set_source_prior(self._h, int(src), 1 if enabled else 0), "rx_set_source_prior")

    # -- batched evaluation, host buffers --------------------------------------------
    def lnprior_batch(self, params, src=0):
        """lnprior of slot `src` alone for [N, 4*ncomp] parameter vectors (rx_lnprior_batch; no solve)."""
        if int(src) not in self._sources:
            raise EngineError("source slot %d not set" % int(src))
        ncomp = self._sources[int(src)]["ncomp"]
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4 * ncomp)
        out = np.empty(params.shape[0])
        self._chk(self._L.rx_lnprior_batch(self._h, int(src), params.shape[0], _dp(params), _dp(out)),
                  "rx_lnprior_batch")
        return out

    def lnprob_batch(self, params, src_index=None, return_info=False):
        si = None if src_index is None else np.ascontiguousarray(src_index, dtype=np.int32).ravel()
        # the batch layout follows the sources the batch addresses (the library checks they all agree)
        slot = 0 if si is None or len(si) == 0 else int(si[0])
        if slot not in self._sources:
            raise EngineError("source slot %d not set" % slot)
        ncomp = self._sources[slot]["ncomp"]
        params = np.ascontiguousarray(params, dtype=np.float64)
        if params.ndim != 2 or params.shape[1] != 4 * ncomp:
            if params.size % (4 * ncomp):
                raise ValueError("params must be [N, %d] for a %d-component source" % (4 * ncomp, ncomp))
            params = params.reshape(-1, 4 * ncomp)
        N = params.shape[0]
        if si is not None and len(si) != N:
            raise ValueError("src_index must have one entry per walker (%d != %d)" % (len(si), N))
        lnp = np.empty(N)
        status = np.empty(N, dtype=np.int32)
        niter = np.empty(N, dtype=np.int32)
        self._chk(self._L.rx_lnprob_batch(self._h, N, _dp(params), _ip(si), _dp(lnp), _ip(status), _ip(niter)),
                  "rx_lnprob_batch")
        return (lnp, status, niter) if return_info else lnp

    def model_flux_batch(self, params, src=0, return_info=False):
        info = self._sources[int(src)]
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 4 * info["ncomp"])
        N = params.shape[0]
        flux = np.empty((N, info["nJ"]))
        status = np.empty(N, dtype=np.int32)
        niter = np.empty(N, dtype=np.int32)
        self._chk(self._L.rx_model_flux_batch(self._h, int(src), N, _dp(params), _dp(flux), _ip(status), _ip(niter)),
                  "rx_model_flux_batch")
        return (flux, status, niter) if return_info else flux

    def solve_batch(self, tkin, cdmol, dens, src=0):
        """dens: [N, npart] in file partner order (see partner_ids)."""
        tkin = np.ascontiguousarray(np.atleast_1d(tkin), dtype=np.float64)
        cdmol = np.ascontiguousarray(np.atleast_1d(cdmol), dtype=np.float64)
        N = tkin.shape[0]
        dens = np.ascontiguousarray(dens, dtype=np.float64).reshape(N, self.npart)
        out = dict(xpop=np.empty((N, self.nlev)), tex=np.empty((N, self.nline)),
                   tau=np.empty((N, self.nline)), sb=np.empty((N, self.nline)),
                   status=np.empty(N, dtype=np.int32), niter=np.empty(N, dtype=np.int32))
        self._chk(self._L.rx_solve_batch(self._h, int(src), N, _dp(tkin), _dp(cdmol), _dp(dens), _dp(out["xpop"]),
                                         _dp(out["tex"]), _dp(out["tau"]), _dp(out["sb"]), _ip(out["status"]),
                                         _ip(out["niter"])), "rx_solve_batch")
        return out

    def lubksb_batch(self, A, return_pivots=False):
        """A: [N, n, n]; returns x[N, n] (last row <- 1, rhs = e_last; see rx_lubksb_batch), and with
        return_pivots the pivot row of every elimination step [N, n] (rx_lubksb_pivots_batch)."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        if A.ndim == 2:
            A = A[None]
        N, n, _ = A.shape
        x = np.empty((N, n))
        if not return_pivots:
            self._chk(self._L.rx_lubksb_batch(self._h, N, n, _dp(A), _dp(x)), "rx_lubksb_batch")
            return x
        piv = np.empty((N, n), dtype=np.int32)
        self._chk(self._L.rx_lubksb_pivots_batch(self._h, N, n, _dp(A), _dp(x), _ip(piv)), "rx_lubksb_pivots_batch")
        return x, piv

    def escprob_batch(self, tau, method=2):
        """beta(tau) of the device escprob_ for geometry `method` (1 sphere, 2 lvg, 3 slab); method 0:
        the kernel's natural logarithm (rx_escprob_batch)."""
        tau = np.ascontiguousarray(tau, dtype=np.float64).ravel()
        out = np.empty_like(tau)
        self._chk(self._L.rx_escprob_batch(self._h, int(method), len(tau), _dp(tau), _dp(out)), "rx_escprob_batch")
        return out

    # -- batched evaluation, device-resident torch tensors -------------------------------
    def set_issue_order(self, hottest_first=True):
        """Scheduling of large batches (rx_set_issue_order); results do not depend on it."""
        self._chk(self._L.rx_set_issue_order(self._h, 1 if hottest_first else 0), "rx_set_issue_order")

    @staticmethod
    def _stream(dev, stream):
        import torch
        return torch.cuda.current_stream(dev).cuda_stream if stream is None else stream

    def lnprob_batch_torch(self, params, lnp=None, status=None, niter=None, src_index=None, stream=None):
        """params: CUDA float64 tensor [N, 4*ncomp] on this engine's device; asynchronous on `stream`."""
        import torch
        assert params.is_cuda and params.dtype == torch.float64 and params.is_contiguous() and params.dim() == 2
        N, ndim = params.shape
        assert ndim in (4, 8), "params must be [N, 4] or [N, 8]"
        dev = params.device
        if lnp is None:
            lnp = torch.empty(N, dtype=torch.float64, device=dev)
        if status is None:
            status = torch.empty(N, dtype=torch.int32, device=dev)
        if niter is None:
            niter = torch.empty(N, dtype=torch.int32, device=dev)
        if src_index is not None:
            assert src_index.is_cuda and src_index.dtype == torch.int32 and src_index.numel() == N
        si = 0 if src_index is None else src_index.data_ptr()
        self._chk(self._L.rx_lnprob_batch_device(self._h, N, ndim // 4, params.data_ptr(), si, lnp.data_ptr(),
                                                 status.data_ptr(), niter.data_ptr(), self._stream(dev, stream)),
                  "rx_lnprob_batch_device")
        return lnp, status, niter

    def model_flux_batch_torch(self, params, src=0, flux=None, status=None, niter=None, stream=None):
        """model_lvg on device buffers (rx_model_flux_batch_device): params a CUDA float64 tensor [N, 4*ncomp(src)]; returns
        flux [N, nJ(src)], status, niter as CUDA tensors; asynchronous on `stream`."""
        import torch
        assert params.is_cuda and params.dtype == torch.float64 and params.is_contiguous() and params.dim() == 2
        N = params.shape[0]
        dev = params.device
        if flux is None:
            flux = torch.empty((N, self._sources[int(src)]["nJ"]), dtype=torch.float64, device=dev)
        if status is None:
            status = torch.empty(N, dtype=torch.int32, device=dev)
        if niter is None:
            niter = torch.empty(N, dtype=torch.int32, device=dev)
        self._chk(self._L.rx_model_flux_batch_device(self._h, int(src), N, params.data_ptr(), flux.data_ptr(), status.data_ptr(),
                                                     niter.data_ptr(), self._stream(dev, stream)), "rx_model_flux_batch_device")
        return flux, status, niter

    def time_lnprob_torch(self, params, lnp, status, niter, reps=10, src_index=None, stream=None):
        """Mean per-launch kernel time [ms] from HIP events recorded on the launch stream."""
        ms = C.c_double(0.0)
        si = 0 if src_index is None else src_index.data_ptr()
        self._chk(self._L.rx_time_lnprob_device(self._h, params.shape[0], params.shape[1] // 4, params.data_ptr(), si,
                                                lnp.data_ptr(), status.data_ptr(), niter.data_ptr(),
                                                self._stream(params.device, stream), int(reps),
                                                C.byref(ms)), "rx_time_lnprob_device")
        return ms.value

    # -- the stretch move on the device (rx_stretch_*_device, rx_sampler_run_device) ---------------
    def stretch_propose_torch(self, nens, nwalkers, a, seed, step, split, coords, q, factor, widx,
                              ens_src=None, qsrc=None, stream=None):
        ndim = coords.shape[-1]
        self._chk(self._L.rx_stretch_propose_device(
            self._h, int(nens), int(nwalkers), int(ndim), float(a), int(seed), int(step), int(split),
            0 if ens_src is None else ens_src.data_ptr(), coords.data_ptr(), q.data_ptr(), factor.data_ptr(),
            widx.data_ptr(), 0 if qsrc is None else qsrc.data_ptr(), self._stream(coords.device, stream)),
            "rx_stretch_propose_device")

    def stretch_accept_torch(self, nens, nwalkers, seed, step, split, q, lnp_q, factor, widx, coords, lnp,
                             naccept=None, stream=None):
        ndim = coords.shape[-1]
        self._chk(self._L.rx_stretch_accept_device(
            self._h, int(nens), int(nwalkers), int(ndim), int(seed), int(step), int(split), q.data_ptr(),
            lnp_q.data_ptr(), factor.data_ptr(), widx.data_ptr(), coords.data_ptr(), lnp.data_ptr(),
            0 if naccept is None else naccept.data_ptr(), self._stream(coords.device, stream)),
            "rx_stretch_accept_device")

    def sampler_run_torch(self, nens, nwalkers, ncomp, a, seed, step0, nsteps, coords, lnp, naccept=None,
                          chain=None, chain_lnp=None, ens_src=None, stream=None, time_solves=False):
        """nsteps stretch-move steps enqueued on `stream`; everything stays in HBM (asynchronous).
        time_solves: wait for the steps and return the summed solve-kernel time [ms] (HIP events)."""
        p = lambda t: 0 if t is None else t.data_ptr()
        ms = C.c_double(0.0)
        self._chk(self._L.rx_sampler_run_device(
            self._h, int(nens), int(nwalkers), int(ncomp), float(a), int(seed), int(step0), int(nsteps),
            p(ens_src), coords.data_ptr(), lnp.data_ptr(), p(naccept), p(chain), p(chain_lnp),
            C.byref(ms) if time_solves else None,
            self._stream(coords.device, stream)), "rx_sampler_run_device")
        return ms.value if time_solves else None

    def sampler_run_async_torch(self, nens, nwalkers, ncomp, a, seed, step0, nsteps, coords, lnp, naccept=None,
                                chain=None, chain_lnp=None, ens_src=None, stream=None):
        """The same chain as sampler_run_torch as ONE persistent dataflow kernel (rx_sampler_run_async_device)."""
        p = lambda t: 0 if t is None else t.data_ptr()
        self._chk(self._L.rx_sampler_run_async_device(
            self._h, int(nens), int(nwalkers), int(ncomp), float(a), int(seed), int(step0), int(nsteps),
            p(ens_src), coords.data_ptr(), lnp.data_ptr(), p(naccept), p(chain), p(chain_lnp),
            self._stream(coords.device, stream)), "rx_sampler_run_async_device")

    def sampler_wait(self, device=None, stream=None):
        import torch
        dev = torch.device("cuda", self.device) if device is None else device
        self._chk(self._L.rx_sampler_wait(self._h, self._stream(dev, stream)), "rx_sampler_wait")

    # -- the dataflow sampler across GPUs: replicas written by peers (rx_sampler_peer_*) -----------------
    def sampler_peer_setup(self, nranks, rank, nens, nwalkers, ncomp, export=True):
        """Allocates this rank's replica; returns its IPC handle (64 bytes) or None with export=False."""
        buf = C.create_string_buffer(_lib.RX_IPC_HANDLE_BYTES) if export else None
        self._chk(self._L.rx_sampler_peer_setup(self._h, int(nranks), int(rank), int(nens), int(nwalkers), int(ncomp),
                                                C.cast(buf, C.c_void_p) if export else None), "rx_sampler_peer_setup")
        return bytes(buf.raw) if export else None

    def sampler_peer_base(self):
        return self._L.rx_sampler_peer_base(self._h)

    def bus_id(self):
        """PCI bus id of the handle's GPU: the same string in every process of the node (rx_device_bus_id)."""
        buf = C.create_string_buffer(_lib.RX_BUS_ID_BYTES)
        self._chk(self._L.rx_device_bus_id(self._h, buf), "rx_device_bus_id")
        return buf.value.decode()

    def sampler_peer_connect(self, ipc_handles=None, bases=None, bus_ids=None):
        """ipc_handles: one 64-byte handle per rank, in rank order -- or bases: device pointers (same process).
        bus_ids: every rank's Engine.bus_id() in rank order: same-GPU / peer-access decisions are taken from them."""
        if bus_ids is not None:
            blob = b"".join(b.encode().ljust(_lib.RX_BUS_ID_BYTES, b"\0")[:_lib.RX_BUS_ID_BYTES] for b in bus_ids)
            self._chk(self._L.rx_sampler_peer_set_bus_ids(self._h, blob), "rx_sampler_peer_set_bus_ids")
        else:
            self._chk(self._L.rx_sampler_peer_set_bus_ids(self._h, None), "rx_sampler_peer_set_bus_ids")
        if ipc_handles is not None:
            blob = b"".join(bytes(x) for x in ipc_handles)
            buf = C.create_string_buffer(blob, len(blob))
            rc = self._L.rx_sampler_peer_connect(self._h, C.cast(buf, C.c_void_p), None)
        else:
            arr = (C.c_void_p * len(bases))(*[C.c_void_p(int(b)) for b in bases])
            rc = self._L.rx_sampler_peer_connect(self._h, None, C.cast(arr, C.c_void_p))
        self._chk(rc, "rx_sampler_peer_connect")

    def sampler_peer_begin(self, coords, lnp, naccept=None, stream=None):
        self._chk(self._L.rx_sampler_peer_begin(self._h, coords.data_ptr(), lnp.data_ptr(),
                                                0 if naccept is None else naccept.data_ptr(),
                                                self._stream(coords.device, stream)), "rx_sampler_peer_begin")

    def sampler_peer_run(self, a, seed, step0, nsteps, device, chain=None, chain_lnp=None, ens_src=None, stream=None):
        p = lambda t: 0 if t is None else t.data_ptr()
        self._chk(self._L.rx_sampler_peer_run(self._h, float(a), int(seed), int(step0), int(nsteps), p(ens_src),
                                              p(chain), p(chain_lnp), self._stream(device, stream)), "rx_sampler_peer_run")

    def sampler_peer_finish(self, coords, lnp, naccept=None, stream=None):
        self._chk(self._L.rx_sampler_peer_finish(self._h, coords.data_ptr(), lnp.data_ptr(),
                                                 0 if naccept is None else naccept.data_ptr(),
                                                 self._stream(coords.device, stream)), "rx_sampler_peer_finish")

    def sampler_peer_disconnect(self):
        """Unmaps the peers' replicas, keeps this rank's own block (every rank, then a barrier, then close / the next set-up)."""
        self._chk(self._L.rx_sampler_peer_disconnect(self._h), "rx_sampler_peer_disconnect")

    def sampler_peer_close(self):
        self._chk(self._L.rx_sampler_peer_close(self._h), "rx_sampler_peer_close")

    def set_sampler_grid_limit(self, cus=0):
        """The peer form's dataflow launches occupy at most `cus` compute units (0: the whole GPU) -- ranks sharing one GPU."""
        self._chk(self._L.rx_set_sampler_grid_limit(self._h, int(cus)), "rx_set_sampler_grid_limit")

    def sampler_stats(self, enable=True):
        """Counters of the dataflow launches since the last call (rx_sampler_stats), then counting on / off."""
        v = (C.c_uint64 * 6)()
        self._chk(self._L.rx_sampler_stats(self._h, 1 if enable else 0, v), "rx_sampler_stats")
        t = dict(tasks=int(v[0]), solved=int(v[1]), niter_sum=int(v[2]), maxiter_solves=int(v[3]),
                 busy_ticks=int(v[4]), wait_ticks=int(v[5]))
        w = (C.c_uint64 * 2)()
        self._chk(self._L.rx_sampler_spec_stats(self._h, w), "rx_sampler_spec_stats")
        t.update(head_starts=int(w[0]), evaluated_twice=int(w[1]))      # (rx_set_sampler_speculation)
        return t

    def set_sampler_speculation(self, mode=-1):
        """The dataflow sampler's head start on the partner's previous position: -1 = automatic (on where one
        wavefront runs per SIMD), 0 = off, 1 = on.  The chain is the same either way (rx_set_sampler_speculation)."""
        self._chk(self._L.rx_set_sampler_speculation(self._h, int(mode)), "rx_set_sampler_speculation")

    def set_sampler_timeout_ms(self, ms):
        self._chk(self._L.rx_set_sampler_timeout_ms(self._h, float(ms)), "rx_set_sampler_timeout_ms")

    def set_sampler_stall_ms(self, ms):
        """No-progress watchdog of the dataflow kernels' waits (rx_set_sampler_stall_ms; default 100 ms)."""
        self._chk(self._L.rx_set_sampler_stall_ms(self._h, float(ms)), "rx_set_sampler_stall_ms")

    def sampler_peer_abort(self):
        """Raise the abort word in every connected replica from the host: ends the peers' kernels at once."""
        self._chk(self._L.rx_sampler_peer_abort(self._h), "rx_sampler_peer_abort")

    def sampler_peer_same_device(self):
        return int(self._L.rx_sampler_peer_same_device(self._h))

    def set_refinement(self, enabled=True):
        """How matrix_'s linear solve is made from iteration 12 on (rx_set_refinement): refinement of a kept solution
        (default) or the pivoted elimination every iteration, as the reference does."""
        self._chk(self._L.rx_set_refinement(self._h, 1 if enabled else 0), "rx_set_refinement")

    def set_refinement_counting(self, enabled=True):
        """Counting for refinement_counters (rx_set_refinement_counting): off by default -- the launches that follow use the
        instantiation of the solve kernel that carries the counters (same results bit for bit, ~2.5 % slower at 1024 walkers)."""
        self._chk(self._L.rx_set_refinement_counting(self._h, 1 if enabled else 0), "rx_set_refinement_counting")

    def refinement_counters(self, reset=True):
        """Totals over the batches evaluated while counting was on (set_refinement_counting) since the last reset: iterations,
        solves made as refinements, corrections, attempts given up, inverses kept."""
        import ctypes as C
        out = (C.c_uint64 * 5)()
        self._chk(self._L.rx_refinement_counters(self._h, out, 1 if reset else 0), "rx_refinement_counters")
        return dict(zip(("iterations", "refined", "corrections", "failed", "kept"), (int(v) for v in out)))

    def set_waves_per_simd(self, waves=0):
        """Scheduling of the solve launches: 0 = automatic, 1 = one wavefront per SIMD, 2 = two."""
        self._chk(self._L.rx_set_waves_per_simd(self._h, int(waves)), "rx_set_waves_per_simd")
