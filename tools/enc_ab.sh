#!/bin/bash
# Same-box A/B of encoder builds: enc_ab.sh "<build> <build> ..." "<batch> <batch> ..." [rounds] — every build (argument of
# tools/exp_build_enc.sh) at every batch size, `rounds` times in turn, device ms of a forward each.
R=$GRAFT_REPO_ROOT
for r in $(seq 1 ${3:-2}); do
  for e in $1; do
    bash $R/tools/exp_build_enc.sh $e || exit 1
    for b in $2; do
      python3 $R/tools/bench_encode.py --batch $b --seq 256 --steps 10 2>/dev/null | python3 -c "import sys,json; print('build $e batch $b', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['device_ms'], 3))"
    done
  done
done
