export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
p=29520
report() { python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/b12_$1.log") if l.startswith("{")][-1])
print("$1", "ms/step", round(d["ms_per_step"],4), "kernel", round(d["roofline"]["kernel_ms"],4), "min", round(d["roofline"]["kernel_ms_min"],4))
PY
}
A="--rows 12500000 --steps 300 --warmup 10 --no-cpu-baseline"
for rep in 1; do
for c in torch native; do
  p=$((p+1)); export MASTER_PORT=$p PCV_BENCH_FORCE_DIST=1
  timeout -k 10 300 python bench.py $A --collective $c > gpurun_out/b12_$c.log 2>&1 || exit 1
  report $c
done
unset PCV_BENCH_FORCE_DIST
timeout -k 10 300 python bench.py $A > gpurun_out/b12_single.log 2>&1 || exit 1
report single
timeout -k 10 300 python -c "import torch, sys, runpy; sys.argv=['bench.py']+'$A'.split(); runpy.run_path('bench.py', run_name='__main__')" > gpurun_out/b12_single_torchrt.log 2>&1 || exit 1
report single_torchrt
done
