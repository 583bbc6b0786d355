#!/bin/bash
# A/B of the headline benchmark on one GPU box: tools/ab_bench.sh <tag> "<env A>" "<env B>" [bench args...]
# Runs bench.py alternately with the two environments (A B A B), prints value per run.
set -u
tag=$1; A=$2; B=$3; shift 3
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_$tag.txt
: > $out
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then e=$A; else e=$B; fi
    line=$(env $e timeout -k 10 150 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --roofline-rows 0 "$@" | grep '^{') || { echo "run $v failed" | tee -a $out; exit 1; }
    echo "$v [$e] $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["Msegments_per_s"],1))')" | tee -a $out
  done
done
