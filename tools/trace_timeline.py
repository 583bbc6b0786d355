#!/usr/bin/env python3
"""Timeline of the LAST render call in a rocprofv3 kernel trace (csv): per kernel its start relative to the call's first
launch, its duration and its queue, then per-kernel totals and the idle gaps of the GPU.

    python3 tools/trace_timeline.py <trace dir> [--rows N]"""
import argparse
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:44]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--rows", type=int, default=80)
    args = ap.parse_args()
    files = glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + args.dir)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    begins = [k for k, r in enumerate(rows) if "k_wf_begin" in r[2]]
    if not begins:
        sys.exit("no k_wf_begin launch in the trace")
    # the last call: from the first k_wf_begin of the last group (pipelines begin together)
    first = begins[-1]
    while first - 1 in begins or (first > 0 and "k_wf_begin" in rows[first - 1][2]):
        first -= 1
    call = rows[first:]
    t0 = call[0][0]
    end = max(r[1] for r in call)
    print(f"last call: {len(call)} launches, {(end - t0) / 1e6:.3f} ms from first launch to last end")
    for s, e, name, q in call[:args.rows]:
        print(f"  {(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  q{q:>3}  {short(name)}")
    tot = {}
    for s, e, name, q in call:
        k = short(name)
        a = tot.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3
    print("per kernel (launches, summed us):")
    for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:6d} {us:12.1f}  {k}")
    # GPU idle: time covered by no kernel
    ev = sorted((s, e) for s, e, _, _ in call)
    idle, cur = 0, ev[0][1]
    for s, e in ev[1:]:
        if s > cur:
            idle += s - cur
        cur = max(cur, e)
    print(f"GPU idle inside the call: {idle / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
