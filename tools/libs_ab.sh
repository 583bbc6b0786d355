#!/bin/bash
# The headline benchmark under several library builds, interleaved (2 rounds): tools/libs_ab.sh <tag> <lib> [<lib> ...]
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/libs_$tag.txt
: > $out
for rep in 1 2; do
  for lib in "$@"; do
    line=$(RTX_AMD_LIB=$lib timeout -k 10 150 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --roofline-rows 0 | grep '^{') || { echo "run failed: $lib" | tee -a $out; exit 1; }
    echo "$lib $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1))')" | tee -a $out
  done
done
