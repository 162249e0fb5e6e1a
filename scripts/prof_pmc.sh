#!/bin/bash
# usage: scripts/prof_pmc.sh <outdir>   (run on the GPU box via gpurun)
# Separate rocprofv3 passes: kernel-trace/stats, then PMC sets (never combined with trace domains).
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-large-batch --no-sampler --no-config3 --no-sharded"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
# the whole default bench as well (sampler kernels, config 3, sharded shapes): per-kernel time only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/trace_full.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- $CMD > $OUT/pmc4.log 2>&1
# the dataflow sampler kernel on SURVEY 8(d)'s config 2 (1024 prior-box walkers as one ensemble: bench.py's
# sampler_config2_prior_box): the same counter sets, one pass each, the program directly behind `--`
CMDS="python3 scripts/sampler_prior_box.py 1024 120"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/strace -- $CMDS > $OUT/strace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/spmc1 -- $CMDS > $OUT/spmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/spmc2 -- $CMDS > $OUT/spmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/spmc3 -- $CMDS > $OUT/spmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/spmc4 -- $CMDS > $OUT/spmc4.log 2>&1
python3 - <<PY
import csv, glob, collections, json, hashlib, sys
sys.path.insert(0, ".")
from radex_emcee_amd import _lib
summary = {"kernel_source_sha256": _lib.kernel_source_sha256(),
           "library_sha256": hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest(),
           "command": "rocprofv3 --pmc <set> -- $CMD (one pass per counter set, scripts/prof_pmc.sh)",
           "workload": "1024 config-2 walkers per dispatch", "counters": {},
           "notes": "FETCH_SIZE/WRITE_SIZE are in KiB as reported by rocprofv3; SQ_*_CYCLES/SQ_ACTIVE_*/SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md)"}
for d in ("pmc1","pmc2","pmc3","pmc4"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rx_solve_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                summary["kernel"] = row["Kernel_Name"]
        for k, v in sorted(acc.items()):
            print("%-24s mean/dispatch %.6g  (n=%d)" % (k, sum(v)/len(v), len(v)))
            summary["counters"][k] = {"mean_per_dispatch": sum(v)/len(v), "dispatches": len(v)}
json.dump(summary, open("$OUT/pmc_summary.json", "w"), indent=1)
# ---- rx_sampler_kernel: the 120-step launch (the dispatch with the most wave cycles; the other is the 20-step burn-in)
ssum = {"kernel_source_sha256": summary["kernel_source_sha256"], "library_sha256": summary["library_sha256"],
        "command": "rocprofv3 --pmc <set> -- $CMDS (one pass per counter set, scripts/prof_pmc.sh)",
        "workload": "dataflow sampler, 1024 prior-box config-2 walkers as one ensemble: per dispatch = ONE launch of 120 steps "
                    "(122880 tasks) after a 20-step burn-in launch", "counters": {}, "notes": summary["notes"]}
for d in ("spmc1","spmc2","spmc3","spmc4"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rx_sampler_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                ssum["kernel"] = row["Kernel_Name"]
        for k, v in sorted(acc.items()):
            print("sampler %-24s per dispatch %s" % (k, ["%.6g" % x for x in v]))
            ssum["counters"][k] = {"per_dispatch": v, "steps_per_dispatch": [20, 120][:len(v)] if len(v) == 2 else None,
                                   "timed_launch": max(v)}
try:
    ssum["run"] = json.loads([l for l in open("$OUT/strace.log") if l.startswith("{")][-1])
except Exception:
    pass
json.dump(ssum, open("$OUT/sampler_pmc_summary.json", "w"), indent=1)
for f in sorted(glob.glob("$OUT/strace/*/*kernel_stats.csv"))[:1]:
    print(open(f).read())
    open("$OUT/sampler_kernel_stats.csv", "w").write(open(f).read())
for f in sorted(glob.glob("$OUT/trace/*/*kernel_stats.csv"))[:1]:
    print(open(f).read())
    open("$OUT/kernel_stats.csv", "w").write(open(f).read())
for f in sorted(glob.glob("$OUT/trace_full/*/*kernel_stats.csv"))[:1]:
    print(open(f).read())
    open("$OUT/kernel_stats_full_bench.csv", "w").write(open(f).read())
# the stats file averages over every dispatch of the kernel, including the one-walker set-up call
# of bench.py; split by grid size so that the 1024-walker launches can be read off directly
for f in sorted(glob.glob("$OUT/trace/*/*kernel_trace.csv"))[:1]:
    by = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "rx_solve_kernel" in row["Kernel_Name"]:
            by[int(row["Grid_Size_X"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    with open("$OUT/kernel_by_grid.csv", "w") as o:
        o.write("kernel,grid_size_x,workgroups,calls,average_ns,min_ns,max_ns\n")
        for g, v in sorted(by.items()):
            line = "rx_solve_kernel,%d,%d,%d,%.1f,%d,%d" % (g, g // 256, len(v), sum(v) / len(v), min(v), max(v))
            print(line); o.write(line + "\n")
PY
