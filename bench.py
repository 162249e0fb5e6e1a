#!/usr/bin/env python3
"""bench.py -- walker-lnlike evaluations / second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (rx_lnprob_batch_device: prior + RADEX LVG solve +
likelihood) over one batch of 1024 synthetic walkers (BASELINE config 2: CO SLED J=1..10,
1 component, walkers uniform in the prior box), parameters already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N>1: one process per GPU; walkers are independent, so every rank evaluates its own
1024-walker batch (weak scaling, no data-path collective; the sampler's all-gather of
log-probabilities is exercised by tests and by radex_emcee_amd.sampler, not timed here).
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


def cpu_baseline(cfg, truth_flux, seconds=15.0):
    """The CPU oracle (a restatement of the reference's path; kind 'port') timed on this
    box's host cores on a bounded sample of the same walkers."""
    from oracle import oracle as O
    from radex_emcee_amd.molecule import default_molfile
    mol = O.Molecule(default_molfile())
    src = O.Source(cfg["tbg"], cfg["Jup"], truth_flux, 0.1 * truth_flux, cfg["bounds"])
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    W = cfg["walkers"]
    t0 = time.time()
    O.lnprob_batch(mol, src, W[:64], nthreads=1)
    per = (time.time() - t0) / 64
    n1 = int(max(64, min(len(W), 0.25 * seconds / per)))
    t0 = time.time()
    _, _, nit = O.lnprob_batch(mol, src, W[:n1], nthreads=1)
    dt1 = time.time() - t0
    # the box may expose more hardware threads than this job is allowed to use: probe a few
    # team sizes briefly and keep the fastest (cores = threads actually used)
    best, cores = None, 1
    for nt in sorted({1, 8, 16, 32, 64, 128, avail}):
        if nt > avail:
            continue
        t0 = time.time()
        O.lnprob_batch(mol, src, W, nthreads=nt)
        d = time.time() - t0
        if best is None or d < best:
            best, cores = d, nt
    reps = int(max(1, min(200, 0.6 * seconds / best)))
    t0 = time.time()
    for _ in range(reps):
        O.lnprob_batch(mol, src, W, nthreads=cores)
    dta = time.time() - t0
    return {"value": round(len(W) * reps / dta, 1), "unit": "evals/s", "cores": cores, "kind": "port",
            "single_core_value": round(n1 / dt1, 1),
            "us_per_iteration_single_core": round(dt1 / float(nit.sum()) * 1e6, 2),
            "hw_threads_visible": avail, "cgroup_cpu_max": _cgroup_cpu_max(),
            "sample": "%d x the same 1024 config-2 walkers on %d OpenMP threads (%.1f s; fastest of "
                      "team sizes 1..%d); %d walkers on 1 thread" % (reps, cores, dta, avail, n1)}


def _measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/latest_pmc_summary.json: FETCH_SIZE and WRITE_SIZE collected in separate passes by
    scripts/prof_pmc.sh on this same command).  rocprofv3 reports KiB; FETCH_SIZE is doubled as
    MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-B requests at 64 B)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc_summary.json")))["counters"]
        return {"bytes_per_launch": int((2.0 * d["FETCH_SIZE"]["mean_per_dispatch"]
                                         + d["WRITE_SIZE"]["mean_per_dispatch"]) * 1024),
                "source": "profiles/latest_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
    except Exception:
        return None


def _cgroup_cpu_max():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(float(q) / float(p), 2)
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
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from radex_emcee_amd import workloads
    from radex_emcee_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RX_BENCH_SHARE_GPU=1: every rank uses GPU 0 and the rendezvous runs over gloo -- only to
    # exercise the multi-rank control flow on a one-GPU box; RCCL needs one GPU per rank.
    share = os.environ.get("RX_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # per-rank batch: same distribution, rank-specific seed (weak scaling)
    cfg = workloads.config2(args.walkers, seed=1234 + rank)
    eng = Engine(device=local)
    eng.set_source(cfg["tbg"], cfg["Jup"], np.ones(10), np.ones(10), cfg["bounds"])
    truth_flux = eng.model_flux_batch(cfg["truth"][None, :])[0]
    eng.set_source(cfg["tbg"], cfg["Jup"], truth_flux, 0.1 * truth_flux, cfg["bounds"])

    P = torch.from_numpy(cfg["walkers"]).to(dev)
    lnp = torch.empty(args.walkers, dtype=torch.float64, device=dev)
    st = torch.empty(args.walkers, dtype=torch.int32, device=dev)
    nit = torch.empty(args.walkers, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.lnprob_batch_torch(P, lnp, st, nit, stream=stream)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.lnprob_batch_torch(P, lnp, st, nit, stream=stream)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # kernel time from HIP events recorded on the launch stream (inside the library)
    kreps = max(5, min(50, args.steps))
    kms = eng.time_lnprob_torch(P, lnp, st, nit, reps=kreps, stream=stream)
    nitc = nit.cpu().numpy()
    stc = st.cpu().numpy()
    solved = int((stc != 3).sum())

    if rank == 0:
        evals = args.walkers * world * args.steps
        value = evals / dt
        niter_mean = float(nitc[stc != 3].mean()) if solved else 0.0
        algo_bytes = ALGO_BYTES_PER_EVAL * args.walkers
        fl = flops_per_eval(niter_mean) * solved
        out = {
            "metric": "walker-lnlike evals/sec (1024 walkers, CO 1-comp)",
            "value": round(value, 1), "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic CO SLED J=1..10, 1-component, "
                                   "%d walkers uniform in the prior box per GPU, z=2.5" % args.walkers,
                       "molecule": os.path.basename(eng.molfile), "walkers_per_gpu": args.walkers,
                       "kernel": eng.kernel_name, "niter_mean": round(niter_mean, 2),
                       "niter_max": int(nitc.max()), "maxiter_walkers": int((stc == 1).sum())},
            "roofline": {"bound": "hbm", "achieved": round(algo_bytes / (kms * 1e-3) / 1e9, 6),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": round(kms, 4), "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "path is fp64-VALU/latency bound (SURVEY 8d); fp64 fraction below"},
            "fp64": {"achieved_tflops": round(fl / (kms * 1e-3) / 1e12, 4),
                     "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                     "frac": fl / (kms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                     "flops_per_eval": round(flops_per_eval(niter_mean), 1)},
        }
        out["roofline"]["traffic"] = _measured_traffic()
        if world == 1 and args.walkers == 1024 and not args.no_large_batch:
            # throughput regime for reference (not the headline): 32768 walkers, 2 waves per SIMD
            cfgL = workloads.config2(32768, seed=5678)
            PL = torch.from_numpy(cfgL["walkers"]).to(dev)
            oL = [torch.empty(32768, dtype=t, device=dev) for t in (torch.float64, torch.int32, torch.int32)]
            eng.lnprob_batch_torch(PL, *oL, stream=stream)
            msL = eng.time_lnprob_torch(PL, *oL, reps=5, stream=stream)
            out["large_batch"] = {"walkers": 32768, "kernel_ms": round(msL, 3),
                                  "value": round(32768 / (msL * 1e-3), 1), "unit": "evals/s"}
        if world == 1 and args.walkers == 1024 and not args.no_large_batch:
            # two INDEPENDENT 1024-walker ensembles (two handles, two streams) in flight together: not the
            # headline either -- one ensemble's steps depend on each other -- but what multi-chain runs
            # see: four fifths of a 1024-walker launch is the tail of its never-converging walkers, which
            # keeps ~23 of the 256 CUs busy; the other ensemble's workgroups run on the CUs that are free
            eng2 = Engine(device=local)
            eng2.set_source(cfg["tbg"], cfg["Jup"], truth_flux, 0.1 * truth_flux, cfg["bounds"])
            cfg2 = workloads.config2(args.walkers, seed=4321)
            P2 = torch.from_numpy(cfg2["walkers"]).to(dev)
            o2 = [torch.empty(args.walkers, dtype=t, device=dev) for t in (torch.float64, torch.int32, torch.int32)]
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
            out["two_ensembles"] = {"walkers": [args.walkers, args.walkers], "ms_per_pair_of_launches": round(d2 / reps2 * 1e3, 4),
                                    "value": round(2 * args.walkers * reps2 / d2, 1), "unit": "evals/s",
                                    "note": "two handles on two HIP streams, independent ensembles; wall time"}
            eng2.close()
        if world == 1 and not args.no_sampler:
            # the caller of the path: stretch-move chain, walkers in a ball around the
            # truth like emcee_radex.py:477, two half-ensemble launches per step, host buffers
            from radex_emcee_amd.sampler import EnsembleSampler
            rs = np.random.RandomState(99)
            p0 = cfg["truth"] + 1e-3 * rs.randn(args.walkers, 4)
            smp = EnsembleSampler(args.walkers, 4, eng.lnprob_batch, vectorize=True, seed=7)
            state = smp.run_mcmc(p0, 5, progress=False)
            ts = time.perf_counter()
            smp.run_mcmc(state, 40, progress=False)
            tsd = time.perf_counter() - ts
            out["sampler"] = {"walker_steps_per_s": round(args.walkers * 40 / tsd, 1), "steps": 40,
                              "ms_per_step": round(tsd / 40 * 1e3, 3),
                              "acceptance": round(float(smp.acceptance_fraction.mean()), 3),
                              "note": "EnsembleSampler on the host (numpy) + rx_lnprob_batch; PCIe inclusive"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, truth_flux)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
