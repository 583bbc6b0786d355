#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + four PMC passes (two SQ, FETCH_SIZE, WRITE_SIZE; each its own run)
# of the benchmark with ONE pipeline (RTX_WF_LANES=1: kernels run one at a time, so durations are exclusive), 10 rows of
# the stratification per step; prints a summary.  The trace run includes bench.py's own roofline pass (the same kind of
# launches), the counter runs time one step and nothing else.
# usage: tools/prof_gpu.sh <tag> [bench args...]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RTX_WF_LANES=${RTX_WF_LANES:-1}
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 2 --warmup 1 --rows 10 --no-cpu-baseline "$@" > $out/trace.log 2>&1 || { echo trace failed; tail -5 $out/trace.log; exit 1; }
grep "^{" $out/trace.log > $out/bench_under_trace.json
cut -d, -f1-6 $out/trace/t_kernel_stats.csv | head -8
PMC="--steps 1 --warmup 0 --rows 10 --roofline-rows 0 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmcA -o a -- python3 bench.py $PMC "$@" > $out/pmcA.log 2>&1 || { echo pmcA failed; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $out/pmcB -o b -- python3 bench.py $PMC "$@" > $out/pmcB.log 2>&1 || { echo pmcB failed; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmcC -o c -- python3 bench.py $PMC "$@" > $out/pmcC.log 2>&1 || { echo pmcC failed; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmcD -o d -- python3 bench.py $PMC "$@" > $out/pmcD.log 2>&1 || { echo pmcD failed; exit 1; }
grep -h "^{" $out/pmcB.log > $out/bench_under_pmc.json
python3 tools/pmc_summary.py $out/pmcA $out/pmcB $out/pmcC $out/pmcD > $out/pmc_summary.txt
cp $out/trace/t_kernel_stats.csv $out/kernel_stats.csv
head -60 $out/pmc_summary.txt
