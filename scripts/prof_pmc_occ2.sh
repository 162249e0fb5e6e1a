#!/bin/bash
# usage: scripts/prof_pmc_occ2.sh <outdir>   (run on the GPU box via gpurun)
# The kernels behind BASELINE configs[2] and [4] -- the two-wavefronts-per-SIMD builds -- under rocprofv3: kernel trace
# + stats in one pass, then ONE --pmc pass per counter set (never combined with a trace domain), the program directly behind `--`:
#   rx_solve_kernel<41, 2, true>     python3 scripts/large_batch_once.py 32768        (4 launches over 32768 config-2 walkers)
#   rx_sampler_kernel<41, 2, true>   python3 scripts/sampler_prior_box.py 65536 6     (65536 walkers: 20 steps burn-in, 6 steps timed)
set -u
OUT=${1:-gpurun_out/prof_occ2}
mkdir -p $OUT
export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
      "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
      "FETCH_SIZE"
      "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE")
run() {   # tag, program args...
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_trace -- "$@" > $OUT/${tag}_trace.log 2>&1
  local i=1
  for set in "${SETS[@]}"; do
    rocprofv3 --pmc $set --output-format csv -d $OUT/${tag}_pmc$i -- "$@" > $OUT/${tag}_pmc$i.log 2>&1
    i=$((i+1))
  done
}
run solve python3 scripts/large_batch_once.py 32768
run sampler python3 scripts/sampler_prior_box.py 65536 6
python3 - <<PY
import csv, glob, collections, json, hashlib, sys
sys.path.insert(0, ".")
from radex_emcee_amd import _lib
base = {"kernel_source_sha256": _lib.kernel_source_sha256(),
        "library_sha256": hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest(),
        "notes": "FETCH_SIZE/WRITE_SIZE are in KiB as reported by rocprofv3 (FETCH_SIZE to be doubled on gfx950, MI355X_MICROARCH.md); "
                 "SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles; one rocprofv3 --pmc pass per counter set (scripts/prof_pmc_occ2.sh)"}
for tag, kern, what in (("solve", "rx_solve_kernel", "python3 scripts/large_batch_once.py 32768: 4 launches of 32768 config-2 walkers (seed 5678), two wavefronts per SIMD"),
                        ("sampler", "rx_sampler_kernel", "python3 scripts/sampler_prior_box.py 65536 6: 65536 prior-box walkers as ONE ensemble, dataflow schedule: "
                                                         "a 20-step burn-in launch and the 6-step timed launch (per_dispatch lists both)")):
    s = dict(base, workload=what, counters={})
    for f in sorted(glob.glob("$OUT/%s_pmc*/*/*counter_collection.csv" % tag)):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                s["kernel"] = row["Kernel_Name"]
        for k, v in sorted(acc.items()):
            s["counters"][k] = {"per_dispatch": v, "mean_per_dispatch": sum(v) / len(v), "max_dispatch": max(v)}
            print("%-8s %-24s %s" % (tag, k, ["%.6g" % x for x in v]))
    for f in sorted(glob.glob("$OUT/%s_trace/*/*kernel_trace.csv" % tag))[:1]:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
        s["kernel_trace_ns"] = d
    try:
        s["run"] = [l.strip() for l in open("$OUT/%s_trace.log" % tag) if l.startswith(("{", "niter"))][-1]
    except Exception:
        pass
    json.dump(s, open("$OUT/occ2_%s_pmc_summary.json" % tag, "w"), indent=1)
    for f in sorted(glob.glob("$OUT/%s_trace/*/*kernel_stats.csv" % tag))[:1]:
        open("$OUT/occ2_%s_kernel_stats.csv" % tag, "w").write(open(f).read())
PY
