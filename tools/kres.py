#!/usr/bin/env python3
"""Compact per-kernel resource table from a gfx950 .s file (hipcc -save-temps): VGPR/AGPR/spill/LDS/scratch."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    agpr, lds, name, scratch, sgpr, vgpr, spill = m.groups()
    try:
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dn = name
    dn = re.sub(r"\(anonymous namespace\)::", "", dn)
    dn = re.sub(r"\(GemmParams\)|void ", "", dn)
    if filt in dn:
        rows.append((dn[:90], int(vgpr), int(agpr), int(sgpr), int(spill), int(scratch), int(lds)))
print(f"{'kernel':90s} vgpr agpr sgpr spill scratch lds")
for r in rows:
    print(f"{r[0]:90s} {r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:5d} {r[5]:7d} {r[6]}")
