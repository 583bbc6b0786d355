#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel (short kernel name) and print one table.
usage: pmc_summary.py <dir-or-csv> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name].add(row["Dispatch_Id"])
for name in sorted(tot, key=lambda n: -tot[n].get("SQ_WAVE_CYCLES", tot[n].get("SQ_INSTS_VALU", 0))):
    print(f"{name}  dispatches={len(calls[name])}")
    for c, v in sorted(tot[name].items()):
        print(f"    {c:28s} {v:.4g}")
