"""ctypes loader for libradex_emcee_amd.so (the HIP/gfx950 engine).

There is no CPU fallback: if the shared library is missing this raises, and
every compute entry point of the library itself fails without a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RADEX_EMCEE_AMD_LIB") or os.path.join(_HERE, "libradex_emcee_amd.so")
CSRC = os.path.join(_HERE, "csrc")

RX_OK, RX_MAXITER, RX_INVALID, RX_PRIOR = 0, 1, 2, 3
RX_MAX_SOURCES = 64
RX_MAX_NJ = 32

# every symbol include/radex_emcee_amd.h declares
EXPORTS = [
    "rx_abi_version", "rx_create", "rx_destroy", "rx_last_error", "rx_nlev", "rx_nline",
    "rx_npart", "rx_partner_ids", "rx_line_data", "rx_set_fortho", "rx_set_iteration_limits",
    "rx_set_source", "rx_lnprob_batch", "rx_lnprob_batch_device", "rx_model_flux_batch",
    "rx_model_flux_batch_device", "rx_solve_batch", "rx_lubksb_batch", "rx_lubksb_pivots_batch", "rx_escprob_batch", "rx_time_lnprob_device",
    "rx_kernel_name", "rx_set_issue_order", "rx_stretch_propose_device", "rx_stretch_accept_device",
    "rx_sampler_run_device", "rx_set_source_prior", "rx_sampler_run_async_device", "rx_sampler_wait",
    "rx_set_sampler_timeout_ms", "rx_set_waves_per_simd",
    "rx_sampler_peer_setup", "rx_sampler_peer_base", "rx_sampler_peer_connect", "rx_sampler_peer_begin",
    "rx_sampler_peer_run", "rx_sampler_peer_finish", "rx_sampler_peer_close", "rx_set_sampler_grid_limit",
    "rx_sampler_stats", "rx_lnprior_batch", "rx_set_sampler_speculation", "rx_sampler_spec_stats",
    "rx_sampler_peer_same_device", "rx_set_sampler_stall_ms", "rx_sampler_peer_abort", "rx_peer_topology",
    "rx_sampler_peer_disconnect", "rx_set_refinement", "rx_set_refinement_counting", "rx_refinement_counters", "rx_background",
    "rx_device_bus_id", "rx_sampler_peer_set_bus_ids",
]
ABI_VERSION = 7
RX_MAX_RANKS = 8
RX_IPC_HANDLE_BYTES = 64
RX_BUS_ID_BYTES = 32


class EngineLibraryMissing(ImportError):
    pass


def build(fast: bool = False, force: bool = False) -> str:
    """Compile the HIP extension in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("rx_api.hip", "rx_kernel.hip.inc", "rx_refine.hip.inc", "rx_sampler.hip.inc", "rx_tables.h", "rx_lamda.h")]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "radex_emcee_amd.h"))
    stale = (not os.path.exists(LIB_PATH)
             or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs))
    if force or stale:
        cmd = ["make", "-C", CSRC, "-B"] + (["FAST=1"] if fast else [])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def kernel_source_sha256() -> str:
    """Hash of everything the DEVICE code is compiled from (kernel sources, table layouts, build flags;
    not the host side of the library): profiles record it, and bench.py refuses a PMC summary that was
    measured on other kernel sources."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rx_kernel.hip.inc", "rx_refine.hip.inc", "rx_sampler.hip.inc", "rx_tables.h"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.join(CSRC, "Makefile")) as fh:          # the compiler flags of the product library
        h.update("".join(l for l in fh if l.startswith("CXXFLAGS")).encode())
    return h.hexdigest()


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineLibraryMissing(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  radex_emcee_amd has no CPU fallback." % LIB_PATH)
    # PyTorch wheels bundle their own libamdhip64; whichever copy is loaded first serves the whole
    # process.  Import torch first so that torch tensors/streams and this library share ONE HIP
    # runtime (loading /opt/rocm's copy first leaves torch with "No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
    L.rx_abi_version.restype = C.c_int
    L.rx_create.restype = vp
    L.rx_create.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_char_p, C.c_size_t]
    L.rx_destroy.argtypes = [vp]
    L.rx_destroy.restype = None
    L.rx_last_error.restype = C.c_char_p
    L.rx_last_error.argtypes = [vp]
    L.rx_kernel_name.restype = C.c_char_p
    L.rx_kernel_name.argtypes = [vp]
    for f in (L.rx_nlev, L.rx_nline, L.rx_npart):
        f.argtypes = [vp]
    L.rx_partner_ids.argtypes = [vp, ip]
    L.rx_line_data.argtypes = [vp, dp, dp, ip, ip]
    L.rx_background.argtypes = [vp, C.c_int, dp, dp]
    L.rx_set_fortho.argtypes = [vp, C.c_double]
    L.rx_set_iteration_limits.argtypes = [vp, C.c_int, C.c_int]
    L.rx_set_source.argtypes = [vp, C.c_int, C.c_double, C.c_int, ip, dp, dp, dp, C.c_int, C.c_double]
    L.rx_lnprob_batch.argtypes = [vp, C.c_int, dp, ip, dp, ip, ip]
    L.rx_lnprob_batch_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    L.rx_set_issue_order.argtypes = [vp, C.c_int]
    L.rx_set_waves_per_simd.argtypes = [vp, C.c_int]
    L.rx_set_refinement.argtypes = [vp, C.c_int]
    L.rx_set_refinement_counting.argtypes = [vp, C.c_int]
    L.rx_refinement_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.rx_set_source_prior.argtypes = [vp, C.c_int, C.c_int]
    L.rx_lnprior_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    u64, i64 = C.c_uint64, C.c_int64
    L.rx_stretch_propose_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, u64, i64, C.c_int,
                                            vp, vp, vp, vp, vp, vp, vp]
    L.rx_stretch_accept_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, u64, i64, C.c_int,
                                           vp, vp, vp, vp, vp, vp, vp, vp]
    L.rx_sampler_run_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, u64, i64, C.c_int,
                                        vp, vp, vp, vp, vp, vp, dp, vp]
    L.rx_model_flux_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp, ip, ip]
    L.rx_model_flux_batch_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.rx_solve_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, ip, ip]
    L.rx_lubksb_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.rx_lubksb_pivots_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp, ip]
    L.rx_escprob_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.rx_sampler_run_async_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, u64, i64, C.c_int,
                                              vp, vp, vp, vp, vp, vp, vp]
    L.rx_sampler_wait.argtypes = [vp, vp]
    L.rx_set_sampler_timeout_ms.argtypes = [vp, C.c_double]
    L.rx_sampler_peer_setup.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.rx_sampler_peer_base.argtypes = [vp]
    L.rx_sampler_peer_base.restype = vp
    L.rx_sampler_peer_connect.argtypes = [vp, vp, vp]
    L.rx_device_bus_id.argtypes = [vp, C.c_char_p]
    L.rx_sampler_peer_set_bus_ids.argtypes = [vp, C.c_char_p]
    L.rx_sampler_peer_begin.argtypes = [vp, vp, vp, vp, vp]
    L.rx_sampler_peer_run.argtypes = [vp, C.c_double, u64, i64, C.c_int, vp, vp, vp, vp]
    L.rx_sampler_peer_finish.argtypes = [vp, vp, vp, vp, vp]
    L.rx_sampler_peer_close.argtypes = [vp]
    L.rx_sampler_peer_disconnect.argtypes = [vp]
    L.rx_set_sampler_grid_limit.argtypes = [vp, C.c_int]
    L.rx_sampler_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
    L.rx_sampler_spec_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rx_set_sampler_speculation.argtypes = [vp, C.c_int]
    L.rx_time_lnprob_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, dp]
    L.rx_sampler_peer_same_device.argtypes = [vp]
    L.rx_set_sampler_stall_ms.argtypes = [vp, C.c_double]
    L.rx_sampler_peer_abort.argtypes = [vp]
    L.rx_peer_topology.argtypes = [C.c_int, C.c_int, ip]
    if L.rx_abi_version() != ABI_VERSION:
        raise EngineLibraryMissing("%s has ABI version %d, this package needs %d: rebuild it"
                                   % (LIB_PATH, L.rx_abi_version(), ABI_VERSION))
    _lib = L
    return L
