# PMC stall attribution of the encoder GEMMs (separate --pmc passes, kernel-trace only), summary in gpurun_out/pmc_enc_summary.txt.
# COMPUTE=f32|bf16x3 gpurun -- "bash tools/profile_encoder_pmc.sh"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python3 $R/tools/bench_encode.py --compute ${COMPUTE:-bf16x3} --steps 2 --warmup 1 > $R/gpurun_out/pmc_enc_$i.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "gemm_bf16x3" if "gemm_bf16x3" in k else "gemm_f32" if "gemm_f32" in k else "attention" if "attention" in k else None
        if not k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_enc_summary.txt", "w")
for k, d in agg.items():
    out.write(k + "\n")
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    for n, v in sorted(d.items()):
        out.write("  %-28s %16.0f  %6.3f of WAVE_CYCLES\n" % (n, v, v / wc))
out.close()
print(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_enc_summary.txt").read())
PY
