export MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PCV_BENCH_FORCE_DIST=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in native torch; do
  export MASTER_PORT=$((29530 + ${#c})) PCV_BENCH_COLLECTIVE=$c
  timeout -k 10 400 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/trace_$c -o t -- python3 $R/bench.py --rows 12500000 --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/trace_$c.log 2>&1 || exit 1
  mkdir -p $R/gpurun_out/trace_$c
  find /tmp/trace_$c -name "*stats*.csv" -exec cp {} $R/gpurun_out/trace_$c/ \;
  # per-step timeline of the last steps: API calls + kernels, trimmed
  python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("/tmp/trace_$c/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api", r["Function"]))
for f in glob.glob("/tmp/trace_$c/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "krn", r["Kernel_Name"][:60]))
rows.sort()
# find the last scan_mfma kernel and print 120 events around the step that contains the one 3 before the end
idx=[i for i,r in enumerate(rows) if r[2]=="krn" and "scan_mfma" in r[3]]
if idx:
    a=idx[-4]; b=idx[-3]
    t0=rows[a][0]
    lo=max(0,a-40)
    with open("$R/gpurun_out/trace_$c/timeline.txt","w") as o:
        for r in rows[lo:b+5]:
            o.write("%10.1f %9.1f %s %s\n"%((r[0]-t0)/1e3,(r[1]-r[0])/1e3,r[2],r[3]))
PY
done
