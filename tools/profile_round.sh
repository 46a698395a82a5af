# Round profiles, as committed under profiles/ by tools/summarize_profiles.py <tag>:
#   kernel-trace statistics of the default bench (100M x 384, batch 64), of BASELINE configs[1]
#   (10M rows, one query) and of the encoder in its three compute modes; FETCH_SIZE / WRITE_SIZE PMC
#   passes of the two scan kernels (counters in their own runs, kernel-trace only).
# Run on the GPU box:  gpurun --timeout 1100 -- "bash tools/profile_round.sh"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
G=$R/gpurun_out
mkdir -p $G
run() {  # run <dir under gpurun_out> <rocprof args...> -- <program...>
  d=$1; shift
  rm -rf $G/$d
  timeout -k 10 400 rocprofv3 "$@" > $G/$d.log 2>&1 || { echo "FAILED: $d"; tail -5 $G/$d.log; exit 1; }
  sleep 7  # the driver clears the ~190 GB the run gave back at ~34 GB/s in the background: 2.6 % of HBM bandwidth meanwhile
}
BENCH="python3 $R/bench.py --no-cpu-baseline --no-extra"
run prof_default     --kernel-trace --stats --output-format csv -d $G/prof_default     -o p -- $BENCH --steps 10
export PCV_SCREEN_COPY=1   # the bf16 screening copy, then the f32 rows themselves, on the same workload
run prof_default_bf16 --kernel-trace --stats --output-format csv -d $G/prof_default_bf16 -o p -- $BENCH --steps 10
run pmc_fetch_bf16    --pmc FETCH_SIZE --kernel-trace --output-format csv -d $G/pmc_fetch_bf16 -o p -- $BENCH --steps 3 --warmup 1
run pmc_write_bf16    --pmc WRITE_SIZE --kernel-trace --output-format csv -d $G/pmc_write_bf16 -o p -- $BENCH --steps 3 --warmup 1
export PCV_SCREEN_COPY=0
run prof_default_f32  --kernel-trace --stats --output-format csv -d $G/prof_default_f32  -o p -- $BENCH --steps 10
unset PCV_SCREEN_COPY
run prof_10m_b1      --kernel-trace --stats --output-format csv -d $G/prof_10m_b1      -o p -- $BENCH --steps 20 --rows 10000000 --batch 1
run prof_clustered   --kernel-trace --stats --output-format csv -d $G/prof_clustered   -o p -- $BENCH --steps 10 --clustered
run prof_12p5m       --kernel-trace --stats --output-format csv -d $G/prof_12p5m       -o p -- $BENCH --steps 20 --rows 12500000
run pmc_fetch        --pmc FETCH_SIZE --kernel-trace --output-format csv -d $G/pmc_fetch        -o p -- $BENCH --steps 3 --warmup 1
run pmc_write        --pmc WRITE_SIZE --kernel-trace --output-format csv -d $G/pmc_write        -o p -- $BENCH --steps 3 --warmup 1
run pmc_fetch_10m_b1 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $G/pmc_fetch_10m_b1 -o p -- $BENCH --steps 3 --warmup 1 --rows 10000000 --batch 1
run pmc_write_10m_b1 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $G/pmc_write_10m_b1 -o p -- $BENCH --steps 3 --warmup 1 --rows 10000000 --batch 1
for c in f32 bf16x3 f16x2; do
  run prof_enc_$c --kernel-trace --stats --output-format csv -d $G/prof_enc_$c -o p -- python3 $R/tools/bench_encode.py --compute $c --steps 7 --warmup 2
done
# keep only the summaries (the traces themselves are large)
find $G/prof_* $G/pmc_* -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete 2>/dev/null
find $G/prof_* $G/pmc_* -name "*.csv" | head -40
echo done
