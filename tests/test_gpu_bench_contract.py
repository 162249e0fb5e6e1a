"""bench.py's one-line JSON contract (needs the GPU: bench.py refuses to run without one)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--no-large-batch", "--no-cpu-baseline"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "evals/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["scaling"] is None and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 * max(r["frac"], 1e-12) + 1e-9
    # value and ms_per_step describe the same timed region
    walkers = d["config"]["walkers_per_gpu"]
    assert abs(d["value"] - walkers / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    # the HIP-event kernel time cannot exceed the wall time per step by more than noise
    assert 0.0 < r["kernel_ms"] < 1.5 * d["ms_per_step"] + 0.2
    # round 2: SIMD-time utilisation, config 3 in one launch, the sharded shapes, the sampler on the device
    assert 0.0 < d["simd_time_utilization"] <= 1.0
    c3 = d["config3"]
    assert c3["walkers_per_launch"] == 8192 and c3["value"] > 0 and c3["kernel_ms"] > 0
    for name in ("config4", "config5"):
        sh = d["sharded"][name]
        assert sh["scaling"] == "strong" and sh["walker_steps_per_s"] > 0
        assert sh["one_gpu_dataflow"]["walker_steps_per_s"] > 0 and sh["speedup_vs_1gpu_dataflow"] == 1.0
    smp = d["sampler"]
    assert smp["half_step_schedule"]["same_chain_as_dataflow"] is True
    # a loose speed guard: the dataflow schedule is there to beat the half-step schedule (2.2 M against 0.43 M walker-steps/s);
    # a regression that loses even that fails here
    assert smp["walker_steps_per_s"] > smp["half_step_schedule"]["walker_steps_per_s"]
    # round 3: the fp64-VALU roofline inside the parsed object, the prior-box ensemble of SURVEY 8(d) in the sampler
    # with the kernel's own counters, BASELINE configs[0] through the reference's call site
    fv = r["fp64_valu"]
    assert fv["bound"] == "fp64-valu" and abs(fv["frac"] - fv["achieved"] / fv["peak"]) < 1e-4 * fv["frac"] + 1e-9
    pb = d["sampler_config2_prior_box"]
    assert pb["steps"] >= 100 and pb["burn_in_steps"] == 20 and pb["walker_steps_per_s"] > 0
    assert pb["tasks"] == 1024 * pb["steps"] and 10.0 <= pb["niter_mean"] <= 200.0 and 0.0 <= pb["maxiter_fraction"] < 0.5
    assert 0.0 < pb["fraction_of_dependency_floor"] <= 2.0          # (two dependent evaluations per step; head starts can beat it)
    assert 0.0 < pb["tasks_with_a_head_start"] < 1.0 and pb["tasks_evaluated_again"] < pb["tasks_with_a_head_start"]
    assert pb["without_head_starts"]["same_chain"] is True and pb["without_head_starts"]["ms_per_step"] > 0
    c0 = d["config0"]
    assert c0["wall_s"] > 0 and c0["schedule"] == "dataflow" and 0.05 < c0["acceptance"] < 0.95
    # round 4: evaluations counted the way SURVEY 8(d) defines them (a proposal that reaches the solver), the sampler's number
    # at the top level of the parsed line, the preflight block
    assert 0.0 <= pb["proposals_outside_the_prior"] < 0.5
    assert abs(pb["evals_reaching_solver_per_s"] - pb["walker_steps_per_s"] * (1.0 - pb["proposals_outside_the_prior"])) < 2e-3 * pb["walker_steps_per_s"]
    assert 0.0 < pb["useful_fp64_frac"] < 1.0 and "evals_reaching_solver_per_s" in pb["unit"]
    assert d["walker_steps_per_s_1024"] == pb["walker_steps_per_s"] and d["evals_reaching_solver_per_s_1024"] == pb["evals_reaching_solver_per_s"]
    for blk in (c0, c3["sampler"], smp["kernel_counters"], d["sharded"]["config4"]["one_gpu_dataflow"], d["sharded"]["config5"]["one_gpu_dataflow"]):
        assert blk["evals_reaching_solver_per_s"] > 0 and 0.0 < blk["useful_fp64_frac"] < 1.0, blk
    pre = d["preflight"]
    assert pre["world_size_counted_by_all_reduce"] == 1 and pre["devices_visible"] >= 1
    assert pre["peer_access"][0][0]["can_access_peer"] == 1
    # traffic comes from the committed PMC summary and only if it was measured on THIS kernel source
    tr = r["traffic"]
    assert tr is None or tr["bytes_per_launch"] is None or tr["bytes_per_launch"] > r["algorithmic_bytes_per_launch"]


def test_bench_gpus_2_starts_its_own_ranks_on_a_shared_gpu():
    """`bench.py --gpus 2` with no launcher starts two rank processes itself (RX_BENCH_SHARE_GPU=1: both on GPU 0,
    gloo for the collectives -- the rehearsal of the multi-GPU control flow on a one-GPU box): rank 0 prints ONE
    line with n_gpus = 2, and the strong-scaling shapes ran under the peer-write dataflow schedule next to the
    one-GPU dataflow number."""
    env = dict(os.environ, RX_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no-config3"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    # the headline is BASELINE's metric: the SAME 1024 walkers, strong scaled (512 per rank), ONE all_gather inside the step
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["config"]["walkers"] == 1024 and d["config"]["walkers_per_gpu"] == 512
    assert "all_gather_into_tensor(1024 x f64)" in d["config"]["collective"]
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    # the weak-scaled pass of earlier rounds: an extra key, not the value
    assert d["weak"]["scaling"] == "weak" and d["weak"]["walkers_total"] == 2048 and d["weak"]["value"] > 0
    # preflight: what the run found, and per shape the schedule that ACTUALLY ran and why
    pre = d["preflight"]
    assert pre["world_size_counted_by_all_reduce"] == 2 and pre["ranks_share_one_gpu"] is True and pre["backend"] == "gloo"
    assert pre["peer_access"][0][0]["can_access_peer"] == 1
    for name in ("config2", "config4", "config5"):
        sc = pre["schedules"][name]
        assert sc["ran"] == "dataflow-peer" and sc["peer_verified_against_halfsteps"] is True, \
            json.dumps(sc) + "\n" + out.stderr[-1500:]                  # (the whole record -- "why" -- and the ranks' warnings)
        assert "identical" in sc["why"] and sc["replicas_on_this_device"] == 2
    c2 = d["sharded"]["config2"]                       # BASELINE configs[1] as ONE ensemble in the sampler, strong scaled
    assert c2["walkers"] == 1024 and c2["proposals_per_rank"] == 256
    for name in ("config2", "config4", "config5"):
        sh = d["sharded"][name]
        assert sh["n_gpus"] == 2 and sh["one_gpu_dataflow"]["walker_steps_per_s"] > 0
        m = sh["multi_gpu_dataflow"]
        assert m["schedule"].startswith("dataflow-peer"), m["schedule"]
        assert m["ranks_share_one_gpu"] is True and m["speedup_vs_1gpu_dataflow"] > 0 and m["speedup_vs_1gpu"] > 0
        assert m["schedule_actually_run"] == "dataflow-peer"
        # the same seed and start under all three schedules: the final states are the one-GPU run's, bit for bit
        assert m["same_final_state_as_one_gpu"] is True and sh["multi_gpu_halfsteps_allgather"]["same_final_state_as_one_gpu"] is True
        assert sh["multi_gpu_halfsteps_allgather"]["speedup_vs_1gpu"] > 0
        # the DEFAULT with a group, schedule="auto": the one-GPU chain bit for bit; ensembles in one GPU's latency regime go to
        # rank 0 alone by rule and cost what one GPU costs (the bar is loose: 12-60 steps on a box that also runs the other rank)
        au = sh["auto"]
        assert au["same_final_state_as_one_gpu"] is True and au["chosen"] in ("rank0", "halfsteps", "dataflow-peer")
        assert pre["schedules"][name]["auto"]["chosen"] == au["chosen"]
        if name in ("config2", "config4"):
            assert au["chosen"] == "rank0" and "by rule" in au["why"], au
            assert au["speedup_vs_1gpu"] > 0.8, au
        else:
            assert "by probe" in au["why"] and set(au["probe"]["seconds"]) >= {"rank0", "halfsteps"}, au
            assert sh["ms_per_step"] <= min(au["ms_per_step"], sh["multi_gpu_halfsteps_allgather"]["ms_per_step"]) + 1e-9


def test_bench_collectives_over_rccl_with_a_one_rank_group():
    """RX_BENCH_FORCE_DIST=1: the process group is initialised with the nccl (= RCCL) backend and every collective of the N > 1
    code path runs on DEVICE buffers -- the world-size count, the all_gather_into_tensor inside the timed step, the barriers, the
    max over ranks -- with ONE rank: what a one-GPU box can exercise of the path the shared-GPU rehearsal runs over gloo."""
    env = dict(os.environ, RX_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--no-large-batch", "--no-config3"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    pre = d["preflight"]
    assert pre["backend"] == "nccl" and pre["world_size_counted_by_all_reduce"] == 1 and pre["ranks_share_one_gpu"] is False
    assert "all_gather_into_tensor" in (d["config"]["collective"] or "")
    for name in ("config4", "config5"):                  # a group of one rank: the peer form of the kernel on its replica block
        sc = pre["schedules"][name]
        assert sc["ran"] == "dataflow-peer", json.dumps(sc)
        sh = d["sharded"][name]
        assert sh["multi_gpu_dataflow"]["same_final_state_as_one_gpu"] is True
        assert sh["multi_gpu_halfsteps_allgather"]["same_final_state_as_one_gpu"] is True
        # schedule="auto" with a group of one rank: rank 0 alone, and the state broadcast is a real RCCL call on device buffers
        assert sh["auto"]["chosen"] == "rank0" and sh["auto"]["same_final_state_as_one_gpu"] is True, sh["auto"]
