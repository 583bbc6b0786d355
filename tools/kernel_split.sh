#!/bin/bash
# Per-kernel exclusive times (bench.py's single-pipeline roofline pass) for one or more library builds:
# tools/kernel_split.sh <tag> <lib> [<lib> ...]
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/split_$tag.txt
: > $out
for lib in "$@"; do
  # an argument VAR=value sets the environment for the builds that follow
  case "$lib" in *=*) export "$lib"; echo "env $lib" | tee -a $out; continue;; esac
  line=$(RTX_AMD_LIB=$lib timeout -k 10 150 python3 bench.py --steps 1 --warmup 1 --rows 10 --no-cpu-baseline --roofline-rows 10 | grep '^{') || { echo "run failed: $lib" | tee -a $out; exit 1; }
  echo "$lib" | tee -a $out
  echo "$line" | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("  value %.1f  pass: wall %.1f ms, kernels %.1f ms, %.1f Msamples/s" % (d["value"], r["pass"]["wall_ms"], r["pass"]["sum_kernel_ms"], r["pass"]["Msamples_per_s"]))
for k in [r]+r["other_kernels"]:
    print("  %-36s %8.2f ms  %5.1f %%" % (k["kernel"], k["gpu_ms"], 100*k["share_of_kernel_time"]))
' | tee -a $out
done
