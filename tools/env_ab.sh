#!/bin/bash
# The headline benchmark under several environments, interleaved (2 rounds): tools/env_ab.sh <tag> "<env 1>" "<env 2>" ...
# (an environment is a string of VAR=value pairs, "-" for none)
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/env_$tag.txt
: > $out
for rep in 1 2; do
  for e in "$@"; do
    ee=$e; [ "$e" = "-" ] && ee="RTX_NOOP=1"
    line=$(env $ee timeout -k 10 150 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --roofline-rows 0 | grep '^{') || { echo "run failed: $e" | tee -a $out; exit 1; }
    echo "[$e] $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1))')" | tee -a $out
  done
done
