"""Checks on the disassembly of every kernel instantiation (CPU only: hipcc cross-compiles).

1. The hand-written v_*_dpp instructions of the linear solve read a DPP source that a VALU instruction
   must not have written in the two preceding wait states; hipcc pads nothing inside or around asm
   statements.  A violation made rx_lubksb_kernel<32> return wrong solutions on the GPU.
2. The persistent item / task loops of rx_solve_kernel and rx_sampler_kernel must be scalar loops: in
   the exec-masked form hipcc 7.2 sometimes builds, wavefronts have been seen to loop for ever.
3. No register-allocator copy / spill traffic in front of the EXEC restore of a control-flow join (the cause of
   round 3's two-wavefront memory fault: profiles/r4_fault_bisect.txt, scripts/check_spill_exec.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def asm_path(tmp_path_factory):
    d = tmp_path_factory.mktemp("rxasm")
    src = os.path.join(ROOT, "radex_emcee_amd", "csrc", "rx_api.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
           "-mllvm", "-pragma-unroll-threshold=4000000", "-mllvm", "-disable-machine-licm", "-save-temps",
           "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src]
    with open(d / "build.log", "w") as log:              # (warnings and the resource-usage remarks: test 4 reads them)
        subprocess.run(cmd, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=log, timeout=1500)
    return str(d / "rx_api-hip-amdgcn-amd-amdhsa-gfx950.s")


def test_every_instantiation_reaches_the_occupancy_it_is_launched_for(asm_path):
    """4. rx_solve_kernel<NL, OCC, ...> / rx_sampler_kernel<NL, OCC, ...> are launched OCC wavefronts per SIMD (rx_api.hip:
    kernel_for): the build must not say `desired occupancy was 2, final occupancy is 1`, and the registers must allow OCC
    (round 5 carried four NL = 48 / 64 instantiations that could never run two per SIMD; they are no longer built)."""
    log = os.path.join(os.path.dirname(asm_path), "build.log")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "resource_usage.py"), "--check", log],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 finding(s)" in r.stdout.strip().splitlines()[-1], r.stdout[-2000:]


def test_no_dpp_read_after_write_hazard(asm_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_dpp_hazards.py"), asm_path],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "0 hazard(s)" in last and not last.startswith("0 DPP"), last


def test_item_loops_are_scalar_loops(asm_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_item_loop.py"), asm_path],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "0 with an exec-masked item loop" in last and not last.startswith("0 persistent"), last


def test_spill_checker_flags_the_faulting_build():
    """The checker on the excerpt of the build that faulted on the GPU (round 3; bisected in round 4): it must find the
    live-range-split copy in front of the EXEC restore -- and nothing once the two instructions are swapped, which is the
    change that made that build run."""
    chk = os.path.join(ROOT, "scripts", "check_spill_exec.py")
    bad = os.path.join(ROOT, "tests", "golden", "join_copy_excerpt.s")
    r = subprocess.run([sys.executable, chk, bad], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "v_mov_b32_e32 v78, v74" in r.stdout and "1 finding(s)" in r.stdout, r.stdout
    lines = open(bad).read().split("\n")
    i = next(k for k, l in enumerate(lines) if l.strip() == "v_mov_b32_e32 v78, v74")
    lines[i], lines[i + 1] = lines[i + 1], lines[i]
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write("\n".join(lines))
    r = subprocess.run([sys.executable, chk, f.name], capture_output=True, text=True, timeout=120)
    os.unlink(f.name)
    assert r.returncode == 0 and "0 finding(s)" in r.stdout, r.stdout


def test_no_vector_op_in_front_of_an_exec_restore(asm_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_spill_exec.py"), asm_path],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "0 finding(s)" in last and not last.startswith("0 functions"), last


def test_slowpath_build_passes_the_checkers(tmp_path):
    """The test build that forces every elimination step through the out-of-line pivot path (`make slowpath`, loaded by
    tests/test_gpu_slowpath.py on the GPU) is different code with different register pressure: the same checks on its CO and toy
    instantiations (the ones that test runs)."""
    src = os.path.join(ROOT, "radex_emcee_amd", "csrc", "rx_api.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
           "-mllvm", "-pragma-unroll-threshold=4000000", "-mllvm", "-disable-machine-licm", "-DRX_FORCE_SETTLE",
           "-DRX_NO_SAMPLER_KERNEL", "-DRX_NL_LIST=8,41", "-DRX_NL_CASES=RX_CASE(8) RX_CASE(41)", "-save-temps", "-c",
           "-o", "/dev/null", src]
    subprocess.run(cmd, cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    asm = str(tmp_path / "rx_api-hip-amdgcn-amd-amdhsa-gfx950.s")
    for script, ok in (("check_spill_exec.py", "0 finding(s)"), ("check_dpp_hazards.py", "0 hazard(s)")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), asm], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and ok in r.stdout.strip().splitlines()[-1], r.stdout[-2000:]
