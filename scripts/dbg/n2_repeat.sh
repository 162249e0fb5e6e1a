mkdir -p gpurun_out/r4
for i in 1 2 3; do
RX_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-config3 > gpurun_out/r4/n2_$i.log 2> gpurun_out/r4/n2_$i.err || exit 1
python - <<PY
import json
l=[x for x in open("gpurun_out/r4/n2_$i.log") if x.startswith("{")][0]
d=json.loads(l)
for k,v in d["preflight"]["schedules"].items(): print($i, k, json.dumps(v)[:600])
PY
done
