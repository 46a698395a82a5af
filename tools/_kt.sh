R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ws in ${WSLIST:-1}; do
  export PCV_GEMM_WS=$ws
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$ws -o t -- python3 $R/tools/bench_encode.py --compute bf16x3 --steps 3 --warmup 1 > $R/gpurun_out/kt_$ws.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("/tmp/kt_$ws/**/*kernel_trace.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
g=[r for r in rows if "gemm_bf16x3" in r["Kernel_Name"]]
last=g[-24:]
print("WS=$ws last forward GEMM durations (us): QKV, out, FFN1, FFN2 per layer")
for l in range(6):
    print("  ", ["%.0f"%((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in last[4*l:4*l+4]])
PY
done
