#!/bin/bash
# `quick` (16 M paths per call) under several environments, 6 calls each (the first one warms up):
# tools/quick_ab.sh <tag> "<env 1>" "<env 2>" ...   (an environment is a string of VAR=value pairs, "-" for none)
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/quick_$tag.txt
: > $out
for e in "$@"; do
  [ "$e" = "-" ] && e="RTX_NOOP=1"
  r=$(env $e timeout -k 10 120 python3 tools/trace_config.py quick --repeat 6 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['calls'][1:]; print(' '.join('%.0f'%x['Msamples_per_s'] for x in c), 'iterations', c[-1]['iterations'])") || { echo "failed: $e" | tee -a $out; exit 1; }
  echo "[$e] $r" | tee -a $out
done
