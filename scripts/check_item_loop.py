"""Checks on the gfx950 assembly that the persistent item / task loops of rx_solve_kernel and
rx_sampler_kernel are SCALAR loops: hipcc 7.2 sometimes builds them exec-masked (depending on unrelated
details of the body), and in that form wavefronts have been seen to loop for ever (rx_kernel.hip.inc, note
above rx_solve_kernel).  A scalar loop has no exec manipulation in the latch block in front of its header.

usage: python scripts/check_item_loop.py [file.s]      (default: builds the .s with -save-temps)
exit code 1 if an exec-masked item loop is found."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_asm():
    d = tempfile.mkdtemp(prefix="rxloop")
    src = os.path.join(ROOT, "radex_emcee_amd", "csrc", "rx_api.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
           "-mllvm", "-pragma-unroll-threshold=4000000", "-mllvm", "-disable-machine-licm", "-save-temps", "-c",
           "-o", "/dev/null", src] + sys.argv[2:]
    subprocess.run(cmd, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(d, "rx_api-hip-amdgcn-amd-amdhsa-gfx950.s")


def main():
    path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else build_asm()
    lines = open(path).read().splitlines()
    func, start, bad, seen = None, 0, 0, 0
    for n, line in enumerate(lines):
        m = re.match(r"^(_ZN3rx[ks]\d+rx_(?:solve|sampler)_kernel\S*):", line)
        if m:
            func, start = m.group(1), n
            continue
        if func and "This Loop Header: Depth=1" in line and "Inner" not in line:
            # the first depth-1 loop WITH child loops is the item / task loop; its latch block sits right above
            latch = [l for l in lines[max(start, n - 14):n] if not l.strip().startswith(";")]
            ops = [l.strip() for l in latch if re.search(r"s_andn2_b64 exec|s_cbranch_exec(n?)z", l)]
            seen += 1
            if ops:
                bad += 1
                print("EXEC-MASKED item loop in %s: %s" % (func, ops))
            func = None
        if line.strip() == "s_endpgm":
            func = None
    print("%d persistent kernels checked, %d with an exec-masked item loop" % (seen, bad))
    if seen == 0:
        print("no rx_solve_kernel / rx_sampler_kernel found in the assembly")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
