#!/usr/bin/env python3
"""Per-segment figures from a summary written by tools/pmc_summary.py: tools/pmc_derive.py <pmc_summary.txt> <segments>
VALU / scalar wave instructions per segment, lanes active per VALU instruction, share of wave cycles spent waiting,
HBM-side bytes per segment (FETCH_SIZE and WRITE_SIZE are in KB; fetch as reported and doubled), LDS conflict share."""
import re
import sys

path, segments = sys.argv[1], float(sys.argv[2])
kernels, cur = {}, None
for line in open(path):
    m = re.match(r"^(.*?)\s*dispatches=(\d+)", line)
    if m:
        cur = kernels.setdefault(m.group(1), {"dispatches": int(m.group(2))})
        continue
    m = re.match(r"^\s+(\w+)\s+(\S+)", line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
tot_valu = 0.0
for name, c in kernels.items():
    if not name.startswith("k_wf_") or c.get("SQ_INSTS_VALU", 0) < 1e6:
        continue
    valu = c["SQ_INSTS_VALU"] / segments
    tot_valu += valu
    f, w = c.get("FETCH_SIZE", 0) * 1024 / segments, c.get("WRITE_SIZE", 0) * 1024 / segments
    print(f"{name}: VALU {valu:.1f} SALU {c['SQ_INSTS_SALU'] / segments:.1f} per segment, lanes {c['SQ_THREAD_CYCLES_VALU'] / c['SQ_INSTS_VALU']:.1f}, "
          f"waiting {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.0f} %, VALU busy {100 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.0f} % of wave cycles, "
          f"bytes/segment fetch {f:.1f} write {w:.1f} -> {f + w:.1f} to {2 * f + w:.1f}, "
          f"LDS conflicts {100 * c['SQ_LDS_BANK_CONFLICT'] / max(1.0, c['SQ_LDS_IDX_ACTIVE']):.0f} % of {c['SQ_LDS_IDX_ACTIVE']:.3g} LDS cycles")
print(f"VALU wave instructions per segment, all kernels: {tot_valu:.1f}")
