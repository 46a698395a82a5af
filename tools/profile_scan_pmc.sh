# Pipe / stall counters of the MFMA scan at 64 and 128 queries (rocprofv3, one small counter set per run).
# Run on the GPU box:  gpurun --timeout 900 -- "bash tools/profile_scan_pmc.sh"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
G=$R/gpurun_out
mkdir -p $G
i=0
for B in 64 128; do
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES"; do
    i=$((i+1))
    rm -rf $G/pmc_scan_$i
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $G/pmc_scan_$i -o p -- \
      python3 $R/bench.py --no-cpu-baseline --no-extra --rows 50000000 --batch $B --steps 2 --warmup 1 > $G/pmc_scan_$i.log 2>&1 || { echo "FAILED set $i"; tail -3 $G/pmc_scan_$i.log; exit 1; }
  done
done
python3 - <<'PY'
import csv, glob, os, collections
G = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
out = open(G + "/pmc_scan_summary.txt", "w")
for i in range(1, 7):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for f in glob.glob(f"{G}/pmc_scan_{i}/*counter_collection.csv") + glob.glob(f"{G}/pmc_scan_{i}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "scan_mfma" not in k: continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[(k, row["Counter_Name"])] += 1
    for k, c in agg.items():
        short = k[k.find("scan_mfma"):][:60]
        for name, v in c.items():
            line = f"set {i}  {short:62s} {name:28s} {v / max(1, calls[(k, name)]):.4g} per launch"
            print(line); out.write(line + "\n")
PY
