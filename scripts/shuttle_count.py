"""Instruction classes per segment of the iteration of rx_solve_kernel<41, 1, true> (static, from the assembly of a -DRX_MARKS
build: comment markers at the segment boundaries): how much of what a lone wavefront issues is register shuttling
(v_accvgpr_read / write, v_readlane / v_writelane = SGPR spill traffic, scratch), plain moves and pads, and how much is work.
Weighted with the measured path mix (refinement counters of the 1024-walker headline) it gives executed counts per iteration.

    python scripts/shuttle_count.py [--kernel 'rx_solve_kernelILi41ELi1ELb1ELb0']

Static counts are upper bounds for segments with internal branches (the escape-probability branches of Phase A, the rare
pivot path of the elimination); the correction loop body is one basic block chain and exact."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kern = "rx_solve_kernelILi41ELi1ELb1ELb0"          # (the kernel every launch uses; ...ELb1ELb1: the counting instantiation)
if "--kernel" in sys.argv:
    kern = sys.argv[sys.argv.index("--kernel") + 1]
d = tempfile.mkdtemp(prefix="rxmarks")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-mllvm",
       "-pragma-unroll-threshold=4000000", "-mllvm", "-disable-machine-licm", "-DRX_MARKS", "-DRX_NL_LIST=8,41",
       "-DRX_NL_CASES=RX_CASE(8) RX_CASE(41)", "-S", "--cuda-device-only", "-o", os.path.join(d, "k.s"),
       os.path.join(ROOT, "radex_emcee_amd", "csrc", "rx_api.hip")]
if not os.environ.get("RX_MARKS_ASM"):
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
else:
    d = os.path.dirname(os.environ["RX_MARKS_ASM"])
lines = open(os.path.join(d, "k.s")).read().split("\n")
sym = kern if kern.startswith("_ZN") else "_ZN3rxk15" + kern          # (the sampler: --kernel _ZN3rxs17rx_sampler_kernelILi41ELi1ELb1)
start = next(i for i, l in enumerate(lines) if l.startswith(sym) and ":" in l.split(";")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])


def cls(op):
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"): return "readlane/writelane"
    if op.startswith("scratch_"): return "scratch"
    if op in ("v_mov_b32_e32", "v_mov_b64_e32", "v_mov_b32_e64"): return "v_mov"
    if op == "s_nop": return "s_nop"
    if op == "s_waitcnt": return "s_waitcnt"
    if op.startswith("v_fmac_f64") or op.startswith("v_fma_f64") or op.startswith("v_mul_f64") or op.startswith("v_add_f64") \
       or op.startswith("v_fmac_f32_dpp"): return "fp FMA/mul/add"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): return "VMEM"
    if op.startswith("v_"): return "other VALU"
    if op.startswith("s_"): return "SALU"
    return "other"


seg = "before the iteration loop"
seen = {}
count = collections.OrderedDict()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r";\s*RXMARK (\w+)", t)
    if m:
        seg = m.group(1)
        seen[seg] = seen.get(seg, 0) + 1
        if seen[seg] > 1: seg += " #%d" % seen[seg]                   # (a kernel that holds the solve more than once)
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    count.setdefault(seg, collections.Counter())[cls(op)] += 1
cols = ["fp FMA/mul/add", "other VALU", "SALU", "LDS", "VMEM", "v_mov", "accvgpr", "readlane/writelane", "scratch", "s_nop", "s_waitcnt"]
print("| segment (static instructions) | total | " + " | ".join(cols) + " |")
print("|---|---|" + "---|" * len(cols))
for s, c in count.items():
    print("| %s | %d | %s |" % (s, sum(c.values()), " | ".join(str(c.get(k, 0)) for k in cols)))
