#!/bin/bash
# Everything a round's profiles/ directory holds besides tools/prof_gpu.sh and tools/prof_config.sh, in one go on the
# GPU box: tools/measure_round.sh <tag>  ->  gpurun_out/<tag>_*
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out
timeout -k 10 600 python3 tools/bench_configs.py --out $o/${tag}_configs.json > $o/${tag}_configs.log 2>&1 || { echo configs failed; tail -5 $o/${tag}_configs.log; exit 1; }
echo configs done
mkdir -p $o/${tag}_tl
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $o/${tag}_tl -o q -- python3 tools/trace_config.py quick --repeat 3 > $o/${tag}_tl.log 2>&1 || { echo timeline failed; exit 1; }
python3 tools/trace_timeline.py $o/${tag}_tl > $o/${tag}_quick_call_timeline.txt || { echo trace_timeline failed; exit 1; }
echo timeline done
RTX_BENCH_REHEARSAL=1 RTX_WF_CAPACITY=8388608 timeout -k 10 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 --rows 10 > $o/${tag}_bench_rehearsal_2ranks.json 2> $o/${tag}_rehearsal.err || { echo rehearsal failed; tail -5 $o/${tag}_rehearsal.err; exit 1; }
echo rehearsal done
RTX_AMD_LIB=tools/_build/librtx_amd_prof.so timeout -k 10 200 python3 tools/prof_trips.py > $o/${tag}_traverse_trip_cycles.txt 2>&1 || { echo trips failed; exit 1; }
RTX_AMD_LIB=tools/_build/librtx_amd_prof.so timeout -k 10 200 python3 tools/prof_shade.py > $o/${tag}_shade_section_cycles.txt 2>&1 || { echo shade sections failed; exit 1; }
echo done
