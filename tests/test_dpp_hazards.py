"""The hand-written v_*_dpp instructions of the linear solve read a DPP source that a VALU
instruction must not have written in the two preceding wait states; hipcc pads nothing inside or
around asm statements, so the disassembly of every instantiation is checked (CPU only: hipcc
cross-compiles).  A violation made rx_lubksb_kernel<32> return wrong solutions on the GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_dpp_read_after_write_hazard():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_dpp_hazards.py"), "-"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert "0 hazard(s)" in last and not last.startswith("0 DPP"), last
