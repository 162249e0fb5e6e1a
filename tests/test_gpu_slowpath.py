"""The linear solve predicts its pivots and confirms each prediction with one compare; a miss takes
an out-of-line path (exact isamax with LINPACK's tie rule, interchange of the two positions,
multipliers again) that ordinary inputs reach in ~0.1 % of the steps.  The test build
`libradex_emcee_amd_slowpath.so` (make slowpath, -DRX_FORCE_SETTLE) sends EVERY step through it; the
parity tests must pass unchanged.  (This is how the NaN rule of the exact search was found: isamax never
flags a column of NaNs as singular.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SLOW = os.path.join(ROOT, "radex_emcee_amd", "libradex_emcee_amd_slowpath.so")


@pytest.mark.gpu
def test_parity_with_every_step_on_the_out_of_line_pivot_path():
    assert os.path.exists(SLOW), "run __graft_entry__.build() (make slowpath) first"
    env = dict(os.environ, RADEX_EMCEE_AMD_LIB=SLOW)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
