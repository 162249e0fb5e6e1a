#!/usr/bin/env python3
"""Per-kernel register / scratch / spill table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python scripts/resource_usage.py [build.log]      (no argument: runs `make resource-usage` itself, ~3.5 min)
    python scripts/resource_usage.py --check build.log   (exit 1 when an instantiation misses the occupancy it is launched for)

Prints one markdown row per kernel; DESIGN.md's "state of the tree" table is this output."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill",
        "VGPRs Spill", "LDS Size [bytes/block]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"\(.*$", "", l).replace("void ", "") for l in out.splitlines()]


def parse(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(?:\S+:\d+:\d+:\s+)?(Function Name|[A-Za-z ]+(?:\[[^\]]+\])?): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None and k in KEYS:
            cur[k] = v
    return rows


def check(text):
    """Every rx_solve_kernel / rx_sampler_kernel <NL, OCC, ...> must reach the OCC wavefronts per SIMD it is launched for:
    no `desired occupancy was N, final occupancy is M` (-Wpass-failed) in the build log, and the register-limited occupancy
    of the resource-usage remarks >= OCC.  Returns the list of findings (empty: fine) and the number of kernels looked at."""
    bad = [l.strip() for l in text.splitlines() if "pass-failed" in l or "desired occupancy" in l]
    rows = parse(text)
    names = demangle([r["name"] for r in rows])
    n = 0
    for r, nm in zip(rows, names):
        m = re.search(r"rx_(?:solve|sampler)_kernel<(\d+), (\d+)", nm)
        if not m:
            continue
        n += 1
        occ, got = int(m.group(2)), int(r.get("Occupancy [waves/SIMD]", 0))
        if got < occ:
            bad.append("%s: %d wavefront(s) per SIMD by its registers, launched for %d" % (nm, got, occ))
    return bad, n


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--check":
        bad, n = check(open(sys.argv[2]).read())
        for b in bad:
            print(b)
        print("%d solve / sampler kernel instantiation(s), %d finding(s)" % (n, len(bad)))
        sys.exit(1 if bad or not n else 0)
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "radex_emcee_amd", "csrc"), "resource-usage"],
                           capture_output=True, text=True)
        text = r.stdout + r.stderr
    rows = parse(text)
    names = demangle([r["name"] for r in rows])
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spills | SGPR spills | waves/SIMD | LDS B/block |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        print("| `%s` | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            n, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
            r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
