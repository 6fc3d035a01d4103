#!/usr/bin/env python3
"""Finds GPU-idle gaps in a rocprofv3 --hip-trace --kernel-trace run and lists the HIP API calls that overlap them.
    rocprofv3 --hip-trace --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload learn ...
    python3 tests/tools/gap_trace.py DIR [min_gap_ms]"""
import csv, glob, sys

d = sys.argv[1]
min_gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 20e6
kt = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)[0]
ht = glob.glob(f"{d}/**/*_hip_api_trace.csv", recursive=True)[0]
ker = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in csv.DictReader(open(kt))))
api = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(ht))))
print(f"{len(ker)} kernels, {len(api)} HIP calls")
t_end = ker[0][1]
for i in range(1, len(ker)):
    s, e, n = ker[i]
    if s - t_end > min_gap:
        print(f"\nGAP {1e-6 * (s - t_end):.1f} ms between  {ker[i-1][2]}  and  {n}")
        lo, hi = t_end, s
        calls = [(a, b, f) for a, b, f in api if b > lo - 2e6 and a < hi + 2e6]
        for a, b, f in calls[:60]:
            print(f"   {1e-6 * (a - lo):9.3f} ms  +{1e-6 * (b - a):8.3f} ms  {f}")
        if len(calls) > 60:
            print(f"   ... {len(calls) - 60} more")
    t_end = max(t_end, e)
