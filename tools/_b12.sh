export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
p=29520
for c in torch native single; do
  p=$((p+1)); export MASTER_PORT=$p
  if [ $c = single ]; then unset PCV_BENCH_FORCE_DIST; extra=""; else export PCV_BENCH_FORCE_DIST=1; extra="--collective $c"; fi
  timeout -k 10 300 python bench.py --rows 12500000 --steps 200 --warmup 10 --no-cpu-baseline $extra > gpurun_out/b12_$c.log 2>&1 || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/b12_$c.log").read().strip().splitlines()[-1])
print("$c", "ms/step", round(d["ms_per_step"],4), "kernel", round(d["roofline"]["kernel_ms"],4), "min", round(d["roofline"]["kernel_ms_min"],4), "frac", round(d["roofline"]["frac"],4))
PY
done
