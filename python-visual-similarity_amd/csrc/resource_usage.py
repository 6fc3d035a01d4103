#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPR / spills / LDS / occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage.  Usage: python resource_usage.py vlad.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_ru.o"],
                     capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: ([A-Za-z ]+): (.+?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        if cur: rows.append(cur)
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
    else: cur[k] = v
if cur: rows.append(cur)
for r in rows:
    if flt and flt not in r["name"]: continue
    print(f'{r["name"][:70]:70s} vgpr={r.get("VGPRs","?"):>4} agpr={r.get("AGPRs","?"):>3} spill={r.get("VGPRs Spill","?"):>3} '
          f'scratch={r.get("ScratchSize [bytes/lane]","?"):>4} occ={r.get("Occupancy [waves/SIMD]","?")} lds={r.get("LDS Size [bytes/block]","?")}')
