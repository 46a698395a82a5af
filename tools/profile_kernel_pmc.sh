# SQ pipe / stall counters of ONE scan shape (rocprofv3, a few counters per run, kernel-trace only besides them).
#   gpurun --timeout 900 -- "bash tools/profile_kernel_pmc.sh <tag> <ab_scan.py arguments...>"
# e.g.  bash tools/profile_kernel_pmc.sh hold256 --rows 50000000 --batch 256
# PCV_PMC_PROG="tools/bench_encode.py" bash tools/profile_kernel_pmc.sh attn --compute f32 --steps 2 --warmup 1    # profiles another program
# writes gpurun_out/pmc_<tag>_summary.txt (per-launch averages of every counter for the scan kernels)
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
G=$R/gpurun_out
mkdir -p $G
if [ -n "$PCV_PMC_PROG" ]; then PROG="$R/$PCV_PMC_PROG"; else PROG="$R/tools/ab_scan.py --rounds 2"; fi
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rm -rf $G/pmc_${TAG}_$i
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $G/pmc_${TAG}_$i -o p -- \
    python3 $PROG "$@" > $G/pmc_${TAG}_$i.log 2>&1 || { echo "FAILED set $i"; tail -5 $G/pmc_${TAG}_$i.log; }
done
python3 - "$TAG" <<'PY'
import csv, glob, os, collections, sys
tag = sys.argv[1]
G = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
out = open(f"{G}/pmc_{tag}_summary.txt", "w")
for i in range(1, 7):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{G}/pmc_{tag}_{i}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "scan_" not in k and "attention" not in k and "gemm" not in k: continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[(k, row["Counter_Name"])] += 1
    for f in glob.glob(f"{G}/pmc_{tag}_{i}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for k, c in agg.items():
        short = k[:70]
        d = dur.get(k, [0])
        for name, v in c.items():
            line = f"set {i}  {short:72s} {name:28s} {v / max(1, calls[(k, name)]):.5g} per launch   (kernel {sum(d)/len(d):.3f} ms avg under the profiler, {len(d)} launches)"
            print(line); out.write(line + "\n")
import shutil
for d in glob.glob(f"{G}/pmc_{tag}_[0-9]*"):
    if os.path.isdir(d): shutil.rmtree(d)
PY
