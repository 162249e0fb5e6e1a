"""Checks the gfx950 assembly of the HIP engine for what hipcc does not check around asm statements:
(1) a VALU write of a VGPR needs two wait states before a v_*_dpp reads it as its DPP source (hipcc
    pads nothing for instructions inside an asm statement);
(2) the destination of an LDS read must not be touched before an s_waitcnt lgkmcnt(N) that covers it
    (hipcc does not count the ds_read instructions of asm statements; LDS operations return in order).

usage: python scripts/check_dpp_hazards.py [file.s]      (default: builds the .s with -save-temps)
exit code 1 if a violation is found."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    tok = tok.strip().strip("|").lstrip("-").strip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def build_asm():
    d = tempfile.mkdtemp(prefix="rxasm")
    src = os.path.join(ROOT, "radex_emcee_amd", "csrc", "rx_api.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
           "-mllvm", "-pragma-unroll-threshold=4000000", "-mllvm", "-disable-machine-licm", "-save-temps", "-c", "-o", "/dev/null", src] + sys.argv[2:]
    subprocess.run(cmd, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(d, "rx_api-hip-amdgcn-amd-amdhsa-gfx950.s")


def main():
    path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else build_asm()
    func, hist, bad, ndpp = None, [], 0, 0
    pend, nlds, bad2 = [], 0, 0           # outstanding LGKM operations in issue order: (line, text, dst regs)
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        if t.endswith(":"):
            if not t.startswith(".L"):
                func, hist = t[:-1], []
            pend = []                     # (2) is checked inside basic blocks (hipcc's own waits may sit in a predecessor)
            continue                      # a branch target: keep the DPP history (conservative for fall-through)
        op, _, rest = t.partition(" ")
        ops = [x.strip() for x in rest.split(",")]
        # ---- (2) LDS read destinations -------------------------------------------------------
        if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64", "s_endpgm"):
            pend = []
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                pend = pend[len(pend) - n:] if n else []
            elif re.fullmatch(r"(0x[0-9a-fA-F]+|\d+)", rest.strip()):
                n = (int(rest.strip(), 0) >> 8) & 0xf
                pend = pend[len(pend) - n:] if n else []
        else:
            touched = set()
            for o in ops:
                touched |= regs(o.split(" ")[0])
            for pl, pt, pd in pend:
                if pd & touched:
                    bad2 += 1
                    print("%s:%d  [%s]  %s   <- touches the destination of the unwaited  %s  (line %d)" % (path, ln, func, t, pt, pl))
                    break
            if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op in ("s_memtime", "s_memrealtime"):
                dst = regs(ops[0]) if (op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle")) else set()
                pend.append((ln, t, dst))
                nlds += 1 if dst else 0
        if op.endswith("_dpp"):
            ndpp += 1
            src0 = regs(ops[1].split(" ")[0])
            states = 0
            for pop, pdst in reversed(hist):
                if states >= 2:
                    break
                if pop.startswith("v_") and (pdst & src0):
                    bad += 1
                    print("%s:%d  [%s]  %s   <- written %d wait state(s) earlier by %s" % (path, ln, func, t, states, pop))
                    break
                states += 1
        if op == "s_nop":
            n = int(ops[0], 0)
            hist.extend([("s_nop", set())] * (n + 1))
        else:
            dst = regs(ops[0]) if op.startswith("v_") and not op.startswith("v_cmp") else set()
            if op.startswith("v_cmpx"):
                dst = set()
            hist.append((op, dst))
        hist = hist[-4:]
    print("%d LDS reads checked, %d touched before their wait" % (nlds, bad2))
    print("%d DPP instructions checked, %d hazard(s)" % (ndpp, bad))
    return 1 if (bad or bad2) else 0


if __name__ == "__main__":
    sys.exit(main())
