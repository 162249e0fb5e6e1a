# `RX_BENCH_SHARE_GPU=1 bench.py --gpus 4` several times (default 3), printing per shape the schedule that ran and why
mkdir -p gpurun_out/r4
for i in $(seq 1 ${1:-3}); do
RX_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 4 --steps 3 --warmup 1 --no-config3 > gpurun_out/r4/n4_$i.log 2> gpurun_out/r4/n4_$i.err || exit 1
python - <<PY
import json
l=[x for x in open("gpurun_out/r4/n4_$i.log") if x.startswith("{")][0]
d=json.loads(l)
for k,v in d["preflight"]["schedules"].items(): print($i, k, v["ran"], "|", v["why"][:160])
PY
done
