# Round profiles, as committed under profiles/ by tools/summarize_profiles.py <tag>: every leg bench.py reports, run
# under rocprofv3 through bench.py itself (--no-extra for the headline-shaped legs, --only <leg> for the others), so the
# kernel each leg names has a kept kernel-trace summary, and the scan legs a FETCH_SIZE / WRITE_SIZE pass each
# (counters in their own runs, kernel-trace only besides them).
# Run on the GPU box, one part per call (a part takes 8-12 minutes):
#   gpurun --timeout 1100 -- "bash tools/profile_round.sh stats"      (the scan legs)
#   gpurun --timeout 1100 -- "bash tools/profile_round.sh encoders"   (the encoder and end-to-end legs)
#   gpurun --timeout 1100 -- "bash tools/profile_round.sh pmc 0 6"    (FETCH / WRITE passes of scan legs 0..5)
#   gpurun --timeout 1100 -- "bash tools/profile_round.sh pmc 6 12"
R=$GRAFT_REPO_ROOT
PART=${1:-stats}
cd /tmp && export TMPDIR=/tmp
G=$R/gpurun_out
mkdir -p $G
run() {  # run <dir under gpurun_out> <rocprof args...> -- <program...>
  d=$1; shift
  rm -rf $G/$d
  timeout -k 10 400 rocprofv3 "$@" > $G/$d.log 2>&1 || { echo "FAILED: $d"; tail -5 $G/$d.log; exit 1; }
  # the per-dispatch trace is large: keep, per kernel, the durations in dispatch order (the summary averages the timed
  # launches of a leg, i.e. the last `steps` of them: the --stats table also counts the warm-up passes)
  python3 - $G/$d <<'PY'
import csv, glob, json, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        if "scan_" in r["Kernel_Name"] or "attention" in r["Kernel_Name"]:
            d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
json.dump(d, open(sys.argv[1] + "/durations.json", "w"))
PY
  echo "ok $d"
  sleep 7  # the driver clears the ~190 GB the run gave back at ~34 GB/s in the background: 2.6 % of HBM bandwidth meanwhile
}
BENCH="python3 $R/bench.py --no-cpu-baseline"
# leg name -> bench.py arguments (the leg's record is the headline of a --no-extra run, or extra[<leg>] of an --only run)
legs=(
  "headline|--no-extra --steps 10"
  "f32_rows_b64|--no-extra --steps 10 --screen off"
  "bf16_copy_b64|--no-extra --steps 10 --screen bf16"
  "batch128|--steps 10 --only batch128"
  "batch256|--steps 10 --only batch256"
  "clustered_b64|--no-extra --steps 10 --warmup 6 --clustered"
  "d768_dot_b64_b128|--steps 10 --only d768_dot_b64,d768_dot_b128"
  "d768_dot_b1|--steps 10 --rows 100000000 --only d768_dot_b1"
  "config2_10m_b1|--no-extra --steps 20 --rows 10000000 --batch 1"
  "shard_12p5m_b64|--no-extra --steps 20 --rows 12500000"
  "shard_12p5m_b256|--steps 10 --only shard_12p5m_b256"
)
if [ "$PART" = stats ]; then
  for l in "${legs[@]}"; do
    name=${l%%|*}; args=${l#*|}
    run prof_$name --kernel-trace --stats --output-format csv -d $G/prof_$name -o p -- $BENCH $args
  done
elif [ "$PART" = encoders ]; then
  run prof_config5_end_to_end --kernel-trace --stats --output-format csv -d $G/prof_config5_end_to_end -o p -- $BENCH --steps 5 --only config5_end_to_end
  run prof_encoder_256x256 --kernel-trace --stats --output-format csv -d $G/prof_encoder_256x256 -o p -- $BENCH --steps 3 --rows 1000000 --only encoder_256x256
  run prof_encoder_256x256_split_precision --kernel-trace --stats --output-format csv -d $G/prof_encoder_256x256_split_precision -o p -- $BENCH --steps 3 --rows 1000000 --only encoder_256x256_split_precision
  run prof_encoder_bertbase_64x256 --kernel-trace --stats --output-format csv -d $G/prof_encoder_bertbase_64x256 -o p -- $BENCH --steps 3 --rows 1000000 --only encoder_bertbase_64x256
  run prof_encoder_32x256 --kernel-trace --stats --output-format csv -d $G/prof_encoder_32x256 -o p -- $BENCH --steps 3 --rows 1000000 --only encoder_32x256
  run prof_encoder_64x256 --kernel-trace --stats --output-format csv -d $G/prof_encoder_64x256 -o p -- $BENCH --steps 3 --rows 1000000 --only encoder_64x256
else
  FROM=${2:-0}; TO=${3:-${#legs[@]}}
  for l in "${legs[@]:$FROM:$((TO-FROM))}"; do
    name=${l%%|*}; args=${l#*|}
    args=${args/--steps 10/--steps 3}; args=${args/--steps 20/--steps 3}
    run pmc_fetch_$name --pmc FETCH_SIZE --kernel-trace --output-format csv -d $G/pmc_fetch_$name -o p -- $BENCH $args
    run pmc_write_$name --pmc WRITE_SIZE --kernel-trace --output-format csv -d $G/pmc_write_$name -o p -- $BENCH $args
  done
fi
# keep only the summaries (the traces themselves are large)
find $G/prof_* $G/pmc_* -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" ! -name "durations.json" -delete 2>/dev/null
find $G/prof_* $G/pmc_* -name "*.csv" | wc -l
echo done
