cd $GRAFT_REPO_ROOT
for f in 0x0 0x2 0x200000 0x100000 0x200002; do
  for rows in 12500000 100000000; do
    PCV_SCAN_FLAGS=$f python bench.py --no-cpu-baseline --no-extra --steps 15 --rows $rows > /tmp/o.log 2>&1
    python - <<PY
import json
d = json.loads(open("/tmp/o.log").read().strip().splitlines()[-1]); r = d["roofline"]
print("flags $f rows $rows kernel_ms %.4f fixed_us %.1f ms/step %.4f cand/q %.1f" % (r["kernel_ms"], r["fixed_cost_us"], d["ms_per_step"], d["candidates_per_query"]))
PY
  done
done
