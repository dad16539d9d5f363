#!/usr/bin/env python
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr file) per kernel: registers, spills, occupancy, LDS.
usage: kres.py <remarks.txt> [name-filter]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
seen = set()
for b in t.split("Function Name: ")[1:]:
    name = b.split("\n")[0].strip()
    if name in seen:
        continue
    seen.add(name)

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"

    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("void mi355x::", "").replace("mi355x::", "")
    if flt and flt not in dn:
        continue
    occ = g(r"Occupancy \[waves/SIMD\]")
    lds = g(r"LDS Size \[bytes/block\]")
    print(f"{dn[:100]:100s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>4s} spill {g('VGPRs Spill'):>3s} sgpr {g('SGPRs'):>3s} occ {occ:>2s} lds {lds}")
