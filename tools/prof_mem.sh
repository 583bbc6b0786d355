#!/bin/bash
# Memory-pipeline counters of one bench step (run on the GPU box through gpurun): vector L1 / texture-address passes.
# usage: tools/prof_mem.sh <tag>
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/mem_$tag
mkdir -p $out
timeout -s KILL 200 rocprofv3 --pmc TCP_PERF_SEL_TOTAL_READ TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -o a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/p1.log 2>&1 || { echo p1 failed; tail -3 $out/p1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d $out/p2 -o b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/p2.log 2>&1 || { echo p2 failed; tail -3 $out/p2.log; exit 1; }
python3 tools/pmc_summary.py $out/p1 $out/p2 > $out/summary.txt
grep -A14 "k_wf_shade\|k_wf_traverse" $out/summary.txt
