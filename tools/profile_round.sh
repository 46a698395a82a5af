# Kernel-trace statistics of the encoder (f32, bf16x3) and of the default bench, as committed under profiles/.
# Run on the GPU box:  gpurun -- "bash tools/profile_round.sh"  then copy gpurun_out/prof/*.csv to profiles/.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -o enc -- python3 $R/tools/bench_encode.py --steps 5 --warmup 2 > $R/gpurun_out/prof_enc.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc3 -o enc3 -- python3 $R/tools/bench_encode.py --compute bf16x3 --steps 5 --warmup 2 > $R/gpurun_out/prof_enc3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc2 -o enc2 -- python3 $R/tools/bench_encode.py --compute f16x2 --steps 5 --warmup 2 > $R/gpurun_out/prof_enc2.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_scan -o scan -- python3 $R/bench.py --steps 10 --no-cpu-baseline > $R/gpurun_out/prof_scan.log 2>&1 || exit 1
mkdir -p $R/gpurun_out/prof
find /tmp/prof_enc /tmp/prof_enc3 /tmp/prof_enc2 /tmp/prof_scan -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof/ \;
ls $R/gpurun_out/prof; tail -1 $R/gpurun_out/prof_scan.log | cut -c1-400
