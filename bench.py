#!/usr/bin/env python3
"""bench.py -- walker-lnlike evaluations / second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (rx_lnprob_batch_device: prior + RADEX LVG solve +
likelihood) over one batch of synthetic walkers (BASELINE configs[1]: CO SLED J=1..10, 1 component,
walkers uniform in the prior box), parameters already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: the batch is the 1024 walkers of config 2.
N > 1: the partitioning north_star names -- the global batch (N x 1024 walkers; rank r's contiguous
block is the config-2 draw with seed 1234 + r) is sharded in blocks of 1024, every rank evaluates its
block on its GPU and ONE all_gather_into_tensor of the log-probabilities (RCCL over xGMI, device
buffers) makes the full vector available on every rank before the stretch move would run: the
collective is INSIDE the timed region.  Per-GPU work is fixed -> "scaling": "weak"; at N = 1 the
collective degenerates and the line is the single-GPU number.  The strong-scaling shapes of
BASELINE configs[3] and [4] (2048 two-component walkers; 65536 walkers) are timed as well, through
the device-resident sampler with the same sharding ("sharded": {...}, walker-steps/s), and the 16
independent ensembles of configs[2] are dealt out 16/N per rank as replicas (no collective).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Work per evaluation, SURVEY.md section 8(d):
ALGO_BYTES_PER_EVAL = 40.0                 # 4 x f64 params in + 1 x f64 lnp out
def flops_per_eval(niter_mean, n=41, L=40, ncoll=820, npart=2):
    return niter_mean * (2.0 / 3.0 * n ** 3 + 7.0 * n * n + 60.0 * L) + 10.0 * ncoll * npart
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6             # MI355X datasheet fp64 vector


def _cgroup_cpu_max():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(float(q) / float(p), 2)
    except Exception:
        return None


def cpu_baseline(cfg, truth_flux, seconds=15.0):
    """The CPU oracle (a restatement of the reference's path; kind 'port') timed on this box's host
    cores on a bounded sample of the same walkers.  cores = the OpenMP team actually used = the CPU
    time this job is GRANTED (cgroup quota), not the hardware threads it can see."""
    from oracle import oracle as O
    from radex_emcee_amd.molecule import default_molfile
    mol = O.Molecule(default_molfile())
    src = O.Source(cfg["tbg"], cfg["Jup"], truth_flux, 0.1 * truth_flux, cfg["bounds"])
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cgroup_cpu_max()
    cores = int(max(1, min(avail, round(quota) if quota else avail)))
    W = cfg["walkers"]
    t0 = time.time()
    O.lnprob_batch(mol, src, W[:64], nthreads=1)
    per = (time.time() - t0) / 64
    n1 = int(max(64, min(len(W), 0.25 * seconds / per)))
    t0 = time.time()
    _, _, nit = O.lnprob_batch(mol, src, W[:n1], nthreads=1)
    dt1 = time.time() - t0
    t0 = time.time()
    O.lnprob_batch(mol, src, W, nthreads=cores)
    d = time.time() - t0
    reps = int(max(1, min(200, 0.6 * seconds / d)))
    t0 = time.time()
    for _ in range(reps):
        O.lnprob_batch(mol, src, W, nthreads=cores)
    dta = time.time() - t0
    return {"value": round(len(W) * reps / dta, 1), "unit": "evals/s", "cores": cores, "kind": "port",
            "single_core_value": round(n1 / dt1, 1),
            "us_per_iteration_single_core": round(dt1 / float(nit.sum()) * 1e6, 2),
            "hw_threads_visible": avail, "cgroup_cpu_max": quota,
            "sample": "%d x the same 1024 config-2 walkers on %d OpenMP threads (%.1f s); "
                      "%d walkers on 1 thread" % (reps, cores, dta, n1)}


def _measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/latest_pmc_summary.json: FETCH_SIZE and WRITE_SIZE collected in separate passes by
    scripts/prof_pmc.sh on this same command).  rocprofv3 reports KiB; FETCH_SIZE is doubled as
    MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-B requests at 64 B).  The summary records
    the hash of the kernel sources it was measured on; a summary of another kernel is refused."""
    from radex_emcee_amd import _lib
    try:
        s = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc_summary.json")))
        d = s["counters"]
        have, want = s.get("kernel_source_sha256"), _lib.kernel_source_sha256()
        if have != want:
            return {"bytes_per_launch": None,
                    "note": "profiles/latest_pmc_summary.json was measured on other kernel sources "
                            "(%s != %s): re-run scripts/prof_pmc.sh" % (str(have)[:12], want[:12])}
        return {"bytes_per_launch": int((2.0 * d["FETCH_SIZE"]["mean_per_dispatch"]
                                         + d["WRITE_SIZE"]["mean_per_dispatch"]) * 1024),
                "kernel_source_sha256": want[:16],
                "source": "profiles/latest_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--walkers", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-batch", action="store_true")
    ap.add_argument("--no-sampler", action="store_true")
    ap.add_argument("--no-config3", action="store_true")
    ap.add_argument("--no-sharded", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from radex_emcee_amd import workloads
    from radex_emcee_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RX_BENCH_SHARE_GPU=1: every rank uses GPU 0 and the collectives run over gloo on host copies --
    # only to exercise the multi-rank control flow on a one-GPU box; RCCL needs one GPU per rank.
    share = os.environ.get("RX_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    # RX_BENCH_FORCE_DIST=1: initialise the process group and run every collective even with ONE rank
    # (nccl = RCCL with world size 1): exercises the N > 1 code path on a one-GPU box
    force = os.environ.get("RX_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    nw = args.walkers

    # this rank's block of the global batch: same distribution, rank-specific seed
    cfg = workloads.config2(nw, seed=1234 + rank)
    eng = Engine(device=local)
    eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    truth_flux = eng.model_flux_batch(cfg["truth"][None, :])[0]
    eng.set_source(cfg["tbg"], cfg["Jup"], truth_flux, 0.1 * truth_flux, cfg["bounds"])

    P = torch.from_numpy(cfg["walkers"]).to(dev)
    lnp = torch.empty(nw, dtype=torch.float64, device=dev)
    st = torch.empty(nw, dtype=torch.int32, device=dev)
    nit = torch.empty(nw, dtype=torch.int32, device=dev)
    lnp_all = torch.empty(nw * world, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def gather(out, mine):
        if share:                                  # gloo: host copies (control-flow rehearsal only)
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, mine.cpu())
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, mine)

    def step():
        eng.lnprob_batch_torch(P, lnp, st, nit, stream=stream)
        if use_dist:
            gather(lnp_all, lnp)                   # log-probabilities of the whole batch on every rank

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # the gathered vector is the same on every rank and holds this rank's block where it belongs
        assert torch.equal(torch.nan_to_num(lnp_all[rank * nw:(rank + 1) * nw], neginf=-1e300),
                           torch.nan_to_num(lnp, neginf=-1e300))

    # kernel time from HIP events recorded on the launch stream (inside the library)
    kreps = max(5, min(50, args.steps))
    kms = eng.time_lnprob_torch(P, lnp, st, nit, reps=kreps, stream=stream)
    nitc = nit.cpu().numpy()
    stc = st.cpu().numpy()
    solved = int((stc != 3).sum())
    n_simd = 4 * torch.cuda.get_device_properties(dev).multi_processor_count

    out = None
    if rank == 0:
        evals = nw * world * args.steps
        value = evals / dt
        niter_mean = float(nitc[stc != 3].mean()) if solved else 0.0
        algo_bytes = ALGO_BYTES_PER_EVAL * nw
        fl = flops_per_eval(niter_mean) * solved
        out = {
            "metric": "walker-lnlike evals/sec (1024 walkers, CO 1-comp)",
            "value": round(value, 1), "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic CO SLED J=1..10, 1-component, "
                                   "%d walkers uniform in the prior box per GPU, z=2.5" % nw
                                   + ("" if world == 1 else "; global batch of %d walkers sharded in blocks of %d, "
                                      "all_gather of log-probabilities (device, RCCL) inside the timed step" % (nw * world, nw)),
                       "molecule": os.path.basename(eng.molfile), "walkers_per_gpu": nw,
                       "kernel": eng.kernel_name, "niter_mean": round(niter_mean, 2),
                       "niter_max": int(nitc.max()), "maxiter_walkers": int((stc == 1).sum()),
                       "collective": None if world == 1 else "all_gather_into_tensor(%d x f64) per step, %s"
                                     % (nw * world, "gloo rehearsal on one GPU" if share else "nccl (RCCL)")},
            "roofline": {"bound": "hbm", "achieved": round(algo_bytes / (kms * 1e-3) / 1e9, 6),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": round(kms, 4), "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "path is fp64-VALU/latency bound (SURVEY 8d); fp64 fraction below"},
            "fp64": {"achieved_tflops": round(fl / (kms * 1e-3) / 1e12, 4),
                     "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                     "frac": fl / (kms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                     "flops_per_eval": round(flops_per_eval(niter_mean), 1)},
            # share of the SIMD-time of the launch that executes iterations: the launch lasts as long as
            # its slowest walker (niter_max iterations on one SIMD) while the mean walker needs niter_mean
            "simd_time_utilization": round(float(nitc.sum()) / (n_simd * max(1, int(nitc.max()))), 4),
        }
        out["roofline"]["traffic"] = _measured_traffic()

    # ---- strong-scaling shapes through the device-resident sampler (all ranks take part) ------------
    if not args.no_sharded:
        from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
        sharded = {}
        for name, ndim, nwk, nst in (("config4", 8, 2048, 12), ("config5", 4, 65536, 6)):
            if name == "config4":
                c = workloads.config4(nwk)
                eng.set_source(c["tbg"], c["Jup"], np.ones(10), np.ones(10), c["bounds"], 2, c["T_d"], src=1)
                tf = eng.model_flux_batch(c["truth"][None, :], src=1)[0]
                eng.set_source(c["tbg"], c["Jup"], tf, 0.1 * tf, c["bounds"], 2, c["T_d"], src=1)
                ens_src = [1]
            else:
                c = workloads.config2(nwk, seed=5678)
                ens_src = None
            grp = dist.group.WORLD if use_dist else None
            smp = DeviceEnsembleSampler(nwk, ndim, engine=eng, seed=2024, ens_src=ens_src, group=grp)
            state = smp.run_mcmc(c["walkers"], 1, store=False)              # initial log-probabilities + 1 step
            barrier()
            ts = time.perf_counter()
            smp.run_mcmc(State(state.coords, state.log_prob), nst, store=False)
            barrier()
            d = time.perf_counter() - ts
            if use_dist:
                tm = torch.tensor([d], dtype=torch.float64, device="cpu" if share else dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                d = float(tm.item())
            sharded[name] = {"walkers": nwk, "ndim": ndim, "steps": nst, "scaling": "strong",
                             "proposals_per_half_step": nwk // 2, "proposals_per_rank": -(-(nwk // 2) // world),
                             "ms_per_step": round(d / nst * 1e3, 3),
                             "walker_steps_per_s": round(nwk * nst / d, 1),
                             "solves_per_s": round(nwk * nst * (ndim // 4) / d, 1),
                             "schedule": "dataflow (one persistent kernel)" if grp is None else
                                         "half-steps: propose, block evaluation per rank, all-gather, accept",
                             "collective": "none (1 GPU)" if grp is None else
                                           "all_gather_into_tensor of %d f64 per half-step (%s)"
                                           % (nwk // 2, "gloo rehearsal, host copies" if share else "RCCL, device")}
            del smp
        if not args.no_config3:
            # BASELINE configs[2] across GPUs: independent ensembles need no exchange at all -- rank r advances
            # its 16 / N sources (1024 walkers each) with the dataflow sampler; replicas, no collective
            c3b = workloads.config3(1024, init="ball")
            per = max(1, 16 // world)
            mine = list(range(rank * per, min(16, (rank + 1) * per)))
            d = 0.0
            if mine:
                e3 = Engine(device=local)
                for k in mine:
                    s = c3b["sources"][k]
                    e3.set_source(s["tbg"], s["Jup"], s["flux"], s["eflux"], s["bounds"], src=k - mine[0])
                sm = DeviceEnsembleSampler(1024, 4, engine=e3, nens=len(mine), ens_src=np.arange(len(mine)), seed=11)
                st3 = sm.run_mcmc(c3b["walkers"][mine], 2, store=False)
                barrier()
                ts = time.perf_counter()
                sm.run_mcmc(State(st3.coords, st3.log_prob), 20, store=False)
                torch.cuda.synchronize()
                d = time.perf_counter() - ts
                e3.close()
            else:
                barrier()
            if use_dist:
                tm = torch.tensor([d], dtype=torch.float64, device="cpu" if share else dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                d = float(tm.item())
            nsrc = min(16, per * world)
            sharded["config3_replicas"] = {"sources": nsrc, "sources_per_rank": per, "walkers": 1024 * nsrc, "steps": 20,
                                           "scaling": "strong", "ms_per_step": round(d / 20 * 1e3, 3),
                                           "walker_steps_per_s": round(1024 * nsrc * 20 / d, 1),
                                           "schedule": "dataflow, one persistent kernel per rank",
                                           "collective": "none: the sources' ensembles are independent (replicas)"}
        if out is not None:
            out["sharded"] = sharded
        eng.set_source(cfg["tbg"], cfg["Jup"], truth_flux, 0.1 * truth_flux, cfg["bounds"])

    if rank == 0:
        if world == 1 and nw == 1024 and not args.no_large_batch:
            # throughput regime for reference (not the headline): 32768 walkers, 2 waves per SIMD
            cfgL = workloads.config2(32768, seed=5678)
            PL = torch.from_numpy(cfgL["walkers"]).to(dev)
            oL = [torch.empty(32768, dtype=t, device=dev) for t in (torch.float64, torch.int32, torch.int32)]
            eng.lnprob_batch_torch(PL, *oL, stream=stream)
            msL = eng.time_lnprob_torch(PL, *oL, reps=5, stream=stream)
            out["large_batch"] = {"walkers": 32768, "kernel_ms": round(msL, 3),
                                  "value": round(32768 / (msL * 1e-3), 1), "unit": "evals/s"}
            # two INDEPENDENT 1024-walker ensembles (two handles, two streams) in flight together: not the
            # headline either -- one ensemble's steps depend on each other -- but what multi-chain runs
            # see: most of a 1024-walker launch is the tail of its never-converging walkers, which
            # keeps ~24 of the 256 CUs busy; the other ensemble's workgroups run on the CUs that are free
            eng2 = Engine(device=local)
            eng2.set_source(cfg["tbg"], cfg["Jup"], truth_flux, 0.1 * truth_flux, cfg["bounds"])
            cfg2 = workloads.config2(nw, seed=4321)
            P2 = torch.from_numpy(cfg2["walkers"]).to(dev)
            o2 = [torch.empty(nw, dtype=t, device=dev) for t in (torch.float64, torch.int32, torch.int32)]
            s2 = torch.cuda.Stream(device=dev)
            reps2 = max(5, min(50, args.steps))
            eng2.lnprob_batch_torch(P2, *o2, stream=s2.cuda_stream)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(reps2):
                eng.lnprob_batch_torch(P, lnp, st, nit, stream=stream)
                eng2.lnprob_batch_torch(P2, *o2, stream=s2.cuda_stream)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t2
            out["two_ensembles"] = {"walkers": [nw, nw], "ms_per_pair_of_launches": round(d2 / reps2 * 1e3, 4),
                                    "value": round(2 * nw * reps2 / d2, 1), "unit": "evals/s",
                                    "note": "two handles on two HIP streams, independent ensembles; wall time"}
            eng2.close()
        if world == 1 and not args.no_config3:
            # BASELINE configs[2]: the 16 sources of flux.dat, 1024 walkers each, ONE launch per half-step
            # (8192 proposals with a per-walker source slot) -- and the whole 16384-walker ensemble set
            c3 = workloads.config3(1024)
            e3 = Engine(device=local)
            for s in c3["sources"]:
                e3.set_source(s["tbg"], s["Jup"], s["flux"], s["eflux"], s["bounds"], src=s["slot"])
            W3 = c3["walkers"]
            half = np.ascontiguousarray(W3[:, :512].reshape(-1, 4))
            P3h = torch.from_numpy(half).to(dev)
            i3h = torch.from_numpy(np.repeat(np.arange(16, dtype=np.int32), 512)).to(dev)
            o3 = [torch.empty(8192, dtype=t, device=dev) for t in (torch.float64, torch.int32, torch.int32)]
            e3.lnprob_batch_torch(P3h, *o3, src_index=i3h, stream=stream)
            ms3 = e3.time_lnprob_torch(P3h, *o3, reps=5, src_index=i3h, stream=stream)
            n3, s3 = o3[2].cpu().numpy(), o3[1].cpu().numpy()
            out["config3"] = {"workload": "BASELINE configs[2]: flux.dat, 16 sources x 1024 walkers, prior-box draw; "
                                          "one half-step = ONE launch of 8192 proposals, per-walker source slot",
                              "walkers_per_launch": 8192, "kernel_ms": round(ms3, 3),
                              "value": round(8192 / (ms3 * 1e-3), 1), "unit": "evals/s",
                              "niter_mean": round(float(n3[s3 != 3].mean()), 2),
                              "maxiter_walkers": int((s3 == 1).sum())}
            if not args.no_sampler:
                from radex_emcee_amd.sampler import DeviceEnsembleSampler, State
                c3b = workloads.config3(1024, init="ball")
                sm3 = DeviceEnsembleSampler(1024, 4, engine=e3, nens=16, ens_src=np.arange(16), seed=11)
                st3 = sm3.run_mcmc(c3b["walkers"], 2, store=False)
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                sm3.run_mcmc(State(st3.coords, st3.log_prob), 10, store=False)
                torch.cuda.synchronize()
                d3 = time.perf_counter() - t3
                out["config3"]["sampler"] = {"walker_steps_per_s": round(16 * 1024 * 10 / d3, 1), "steps": 10,
                                             "ms_per_step": round(d3 / 10 * 1e3, 3),
                                             "note": "16 chains advanced together on the device, walkers started in "
                                                     "the reference's ball (emcee_radex.py:477)"}
            e3.close()
        if world == 1 and not args.no_sampler:
            # the caller of the path: stretch-move chain, walkers in a ball around the
            # truth like emcee_radex.py:477, two half-ensemble launches per step
            from radex_emcee_amd.sampler import DeviceEnsembleSampler, EnsembleSampler, State
            rs = np.random.RandomState(99)
            p0 = cfg["truth"] + 1e-3 * rs.randn(nw, 4)
            dsm = DeviceEnsembleSampler(nw, 4, engine=eng, seed=7)       # schedule="dataflow"
            sd = dsm.run_mcmc(p0, 20, store=False)                       # a short burn-in
            torch.cuda.synchronize()
            nst = 100
            ts = time.perf_counter()
            sd = dsm.run_mcmc(State(sd.coords, sd.log_prob), nst, store=False)
            torch.cuda.synchronize()
            tsd = time.perf_counter() - ts
            # the same 100 steps again under the half-step schedule (identical proposals), then once more with
            # HIP events around every solve launch: the mean kernel time of THIS chain's half-steps (a
            # half-step lasts as long as its slowest proposal, which varies from one half-step to the next)
            dsm2 = DeviceEnsembleSampler(nw, 4, engine=eng, seed=7, schedule="halfsteps")
            s2 = dsm2.run_mcmc(p0, 20, store=False)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            s2b = dsm2.run_mcmc(State(s2.coords, s2.log_prob), nst, store=False)
            torch.cuda.synchronize()
            tsh = time.perf_counter() - ts
            same = bool(np.array_equal(s2b.coords, sd.coords) and np.array_equal(s2b.log_prob, sd.log_prob))
            dsm3 = DeviceEnsembleSampler(nw, 4, engine=eng, seed=7, schedule="halfsteps")
            s3 = dsm3.run_mcmc(p0, 20, store=False)
            dsm3.time_solves = True
            dsm3.run_mcmc(State(s3.coords, s3.log_prob), nst, store=False)
            hms = dsm3.last_solve_ms / (2 * nst)
            ideal = nw / (2.0 * hms * 1e-3)
            out["sampler"] = {"walker_steps_per_s": round(nw * nst / tsd, 1), "steps": nst,
                              "ms_per_step": round(tsd / nst * 1e3, 4),
                              "schedule": "dataflow: ONE persistent kernel, every proposal starts when the two "
                                          "walkers it reads are final (rx_sampler_run_async_device)",
                              "acceptance": round(float(dsm.acceptance_fraction.mean()), 3),
                              "half_step_schedule": {
                                  "walker_steps_per_s": round(nw * nst / tsh, 1), "ms_per_step": round(tsh / nst * 1e3, 4),
                                  "half_step_kernel_ms": round(hms, 4),
                                  "bound_walkers_over_two_half_step_kernels": round(ideal, 1),
                                  "fraction_of_bound": round(nw * nst / tsh / ideal, 4),
                                  "same_chain_as_dataflow": same,
                                  "note": "propose / solve / accept launches per half-step (rx_sampler_run_device); a "
                                          "half-step lasts as long as its slowest proposal"},
                              "note": "DeviceEnsembleSampler: positions, log-probabilities and the Philox stream "
                                      "resident in HBM; no PCIe per step"}
            smp = EnsembleSampler(nw, 4, eng.lnprob_batch, vectorize=True, seed=7)
            state = smp.run_mcmc(p0, 5, progress=False)
            ts = time.perf_counter()
            smp.run_mcmc(state, 20, progress=False)
            tsd = time.perf_counter() - ts
            out["sampler"]["host_sampler"] = {"walker_steps_per_s": round(nw * 20 / tsd, 1),
                                              "ms_per_step": round(tsd / 20 * 1e3, 3),
                                              "note": "EnsembleSampler on the host (numpy) + rx_lnprob_batch; "
                                                      "PCIe inclusive; the checker of the device sampler"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, truth_flux)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
