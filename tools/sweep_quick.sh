#!/bin/bash
# Sweep of the wavefront loop's knobs on one configuration (run on the GPU box): pipelines, iterations per host check,
# survivors the finishing kernel takes over.  usage: tools/sweep_quick.sh <config> [trace_config args]
cfg=$1; shift
for lanes in 4 2 1; do for chunk in 8 4 2; do for fin in 65536 262144 1048576; do
  r=$(RTX_WF_LANES=$lanes RTX_WF_CHUNK=$chunk RTX_WF_FINISH=$fin timeout -k 10 120 python3 tools/trace_config.py $cfg --repeat 3 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['calls'][1:]; print(' '.join('%.0f'%x['Msamples_per_s'] for x in c), c[-1]['iterations'])") || exit 1
  echo "lanes=$lanes chunk=$chunk finish=$fin : $r"
done; done; done
