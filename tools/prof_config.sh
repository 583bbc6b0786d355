#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + PMC passes (two SQ, FETCH_SIZE, WRITE_SIZE; each its own run) of
# one configuration rendered by tools/trace_config.py with ONE pipeline (kernel durations are then exclusive).
# usage: tools/prof_config.sh <tag> <config> [trace_config args...]
set -u
tag=$1; cfg=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RTX_WF_LANES=${RTX_WF_LANES:-1}
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 tools/trace_config.py $cfg "$@" > $out/trace.log 2>&1 || { echo trace failed; tail -5 $out/trace.log; exit 1; }
grep "^{" $out/trace.log > $out/run_under_trace.json
cut -d, -f1-6 $out/trace/t_kernel_stats.csv | head -8
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmcA -o a -- python3 tools/trace_config.py $cfg "$@" > $out/pmcA.log 2>&1 || { echo pmcA failed; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $out/pmcB -o b -- python3 tools/trace_config.py $cfg "$@" > $out/pmcB.log 2>&1 || { echo pmcB failed; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmcC -o c -- python3 tools/trace_config.py $cfg "$@" > $out/pmcC.log 2>&1 || { echo pmcC failed; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmcD -o d -- python3 tools/trace_config.py $cfg "$@" > $out/pmcD.log 2>&1 || { echo pmcD failed; exit 1; }
grep -h "^{" $out/pmcB.log > $out/run_under_pmc.json
python3 tools/pmc_summary.py $out/pmcA $out/pmcB $out/pmcC $out/pmcD > $out/pmc_summary.txt
cp $out/trace/t_kernel_stats.csv $out/kernel_stats.csv
head -70 $out/pmc_summary.txt
