#!/usr/bin/env python
"""per-kernel register / occupancy table of a hipcc -S listing:  python scripts/isa_regs.py file.s [filter]"""
import re, subprocess, sys
cur, rows = None, []
for ln in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1)
    m = re.match(r"^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", ln)
    if m and cur:
        if not rows or rows[-1][0] != cur:
            rows.append((cur, {}))
        rows[-1][1][m.group(1)] = int(m.group(2))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for name, d in rows:
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem.replace("(anonymous namespace)::", "").replace("void ", ""))
    if flt in dem:
        print(f"{dem:70s} vgpr {d.get('NumVgprs'):4d} agpr {d.get('NumAgprs'):4d} total {d.get('TotalNumVgprs'):4d} scratch {d.get('ScratchSize'):4d} lds {d.get('LDSByteSize', -1):6d} occ {d.get('Occupancy')}")
