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
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["data"] == "synthetic"
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
    smp = d["sampler"]
    assert smp["half_step_schedule"]["same_chain_as_dataflow"] is True
    assert smp["walker_steps_per_s"] > smp["half_step_schedule"]["walker_steps_per_s"]
    assert 0.8 < smp["half_step_schedule"]["fraction_of_bound"] <= 1.05
    # traffic comes from the committed PMC summary and only if it was measured on THIS kernel source
    tr = r["traffic"]
    assert tr is None or tr["bytes_per_launch"] is None or tr["bytes_per_launch"] > r["algorithmic_bytes_per_launch"]
