# timing-only ablations of the fixed per-pass kernels (diagnostic build with -DPCV_DIAG; results are wrong by design)
cd $GRAFT_REPO_ROOT
cp perceive_amd/libperceive_hip.so /tmp/good.so; cp perceive_amd/libperceive_hip_diag.so perceive_amd/libperceive_hip.so
export TMPDIR=/tmp
for f in 0x2 0x12 0x22 0x32 0x82 0x102 0x182; do
  rm -rf /tmp/tl
  (cd /tmp && PCV_SCAN_FLAGS=$f timeout -k 10 60 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 10 --rows 12500000 > /tmp/o.log 2>&1)
  python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/tl/**/*kernel_stats.csv", recursive=True):
    r = {}
    for x in csv.DictReader(open(f)):
        for key in ("prep_seed", "rescore_select", "upload", "scan_mfma"):
            if key in x["Name"]:
                r[key] = round(float(x["AverageNs"]) / 1e3, 1)
    print("flags $f", r)
PY
done
cp /tmp/good.so perceive_amd/libperceive_hip.so
