"""Diagnostic: the in-process two-handle form of the peer dataflow sampler on several shapes; on a timeout prints
how far every walker got in both replicas.  usage: python3 scripts/dbg/peer_inproc.py shape nw nsteps [shape nw nsteps ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from radex_emcee_amd import workloads
from radex_emcee_amd.engine import Engine, EngineError
from radex_emcee_amd.sampler import DeviceEnsembleSampler

def hip():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    raise RuntimeError("no hip runtime mapped")

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

STREAMS = None

def run(shape, nw, nsteps, limit=None, same_stream_order=False):
    dev = torch.device("cuda", 0)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    engs = [Engine(), Engine()]
    for e in engs:
        p0, nc = setup(e, shape, nw)
        e.set_sampler_grid_limit(limit or ncu // 2)
    ndim = 4 * nc
    ref = DeviceEnsembleSampler(nw, ndim, engine=engs[0], seed=5)
    st_ref = ref.run_mcmc(p0, nsteps)
    lnp0 = ref.compute_log_prob(p0)
    for r, e in enumerate(engs):
        e.sampler_peer_setup(2, r, 1, nw, nc, export=False)
    bases = [e.sampler_peer_base() for e in engs]
    global STREAMS
    if os.environ.get("REUSE_STREAMS") and STREAMS:
        streams = STREAMS
    else:
        streams = STREAMS = [torch.cuda.Stream(device=dev) for _ in engs]
    print("streams", [hex(s.cuda_stream) for s in streams], flush=True)
    state = []
    for r, e in enumerate(engs):
        e.sampler_peer_connect(bases=bases)
        state.append((torch.from_numpy(np.ascontiguousarray(p0)).to(dev), lnp0.clone(), torch.zeros(nw, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()
    for r, e in enumerate(engs):
        e.sampler_peer_begin(*state[r], stream=streams[r].cuda_stream)
    for r, e in enumerate(engs):
        e.sampler_peer_run(2.0, 5, 0, nsteps, dev, None, None, stream=streams[r].cuda_stream)
    ok = True
    for r, e in enumerate(engs):
        e.sampler_wait(dev, stream=streams[r].cuda_stream)
    for r, e in enumerate(engs):
        try:
            e.sampler_peer_finish(*state[r], stream=streams[r].cuda_stream)
        except EngineError as exc:
            ok = False
            print("rank", r, "FAILED:", exc.rc)
    torch.cuda.synchronize()
    if ok:
        same = all(np.array_equal(state[r][0].cpu().numpy(), st_ref.coords) for r in range(2))
        print("%s nw=%d nsteps=%d: finished, chain identical to one-GPU run: %s" % (shape, nw, nsteps, same), flush=True)
    else:
        H = hip()
        for r, b in enumerate(bases):
            v = np.zeros(nw + 16, dtype=np.uint32)
            H.hipMemcpy(v.ctypes.data_as(C.c_void_p), C.c_void_p(b), C.c_size_t(v.nbytes), C.c_int(2))
            vals, cnt = np.unique(v[:nw], return_counts=True)
            print("replica %d: versions %s" % (r, dict(zip(vals.tolist(), cnt.tolist()))), flush=True)
    for e in engs:
        e.sampler_peer_close(); e.close()
    return ok

a = sys.argv[1:]
for i in range(0, len(a), 3):
    run(a[i], int(a[i + 1]), int(a[i + 2]))
