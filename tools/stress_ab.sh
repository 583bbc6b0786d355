#!/bin/bash
# The stress configuration (1 M primitives, 8 sample indices per call) under several library builds / environments:
# tools/stress_ab.sh <tag> "<env 1>" ...
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/stress_$tag.txt
: > $out
for e in "$@"; do
  [ "$e" = "-" ] && e="RTX_NOOP=1"
  r=$(env $e timeout -k 10 200 python3 tools/bench_configs.py --no-cpu --only stress 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_Msamples_per_s'],1), round(d['gpu_Msegments_per_s'],1), d['mean_radiance'])") || { echo "failed: $e" | tee -a $out; exit 1; }
  echo "[$e] $r" | tee -a $out
done
