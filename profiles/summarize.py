#!/usr/bin/env python3
"""Condense a gpurun_out/prof/ directory (rocprofv3 CSVs) into the small files committed under profiles/.

    python profiles/summarize.py gpurun_out/prof r02                       # the headline run
    python profiles/summarize.py gpurun_out/prof_vlad512 r02 vlad512 "bench.py --workload vlad512 ..."

Expects <dir>/stats (rocprofv3 --kernel-trace --stats), <dir>/fetch (--pmc FETCH_SIZE) and <dir>/write
(--pmc WRITE_SIZE) -- counters collected in their own passes, as the MI355X guide prescribes."""
import collections, csv, glob, json, sys

src, tag = sys.argv[1], sys.argv[2]
name = sys.argv[3] if len(sys.argv) > 3 else ""
command = sys.argv[4] if len(sys.argv) > 4 else "bench.py --steps 3 --warmup 1 --no-cpu-baseline"
stem = f"profiles/{tag}_{name}" if name else f"profiles/{tag}"


def one(pattern):
    hits = glob.glob(f"{src}/{pattern}", recursive=True)
    return hits[0] if hits else None


stats = one("stats/**/*_kernel_stats.csv")
if stats:
    rows = list(csv.reader(open(stats)))
    with open(f"{stem}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:110]
            w.writerow(r)
out = {}
for cname, pattern in (("FETCH_SIZE_KiB", "fetch/**/*_counter_collection.csv"),
                      ("WRITE_SIZE_KiB", "write/**/*_counter_collection.csv")):
    path = one(pattern)
    if not path:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "pvs::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[cname] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
for k, d in out.items():
    if "FETCH_SIZE_KiB" in d and "WRITE_SIZE_KiB" in d:
        f_, w_ = d["FETCH_SIZE_KiB"]["mean_per_launch"], d["WRITE_SIZE_KiB"]["mean_per_launch"]
        # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads -> doubled
        # (MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for 16-B-per-lane stores.
        d["hbm_bytes_per_launch_corrected"] = (2.0 * f_ + w_) * 1024.0
# MFMA pipe utilisation: busy cycles of the matrix pipes / (4 SIMDs x busy CU cycles), summed over the launches of a kernel
path = one("mfma/**/*_counter_collection.csv")
if path:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "pvs::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        if d.get("SQ_BUSY_CU_CYCLES"):
            out.setdefault(k, {})["mfma_pipe_busy_frac"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * d["SQ_BUSY_CU_CYCLES"]), 4)
# per-launch spread of every pvs kernel (kernel trace of the stats pass): min / median / max duration
trace = one("stats/**/*_kernel_trace.csv")
if trace:
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "pvs::" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    for k, v in dur.items():
        v.sort()
        out.setdefault(k, {})["launch_ms"] = {"n": len(v), "min": round(v[0], 4), "median": round(v[len(v) // 2], 4), "max": round(v[-1], 4),
                                               "mean": round(sum(v) / len(v), 4)}
# effective clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (MI355X guide, DVFS give-back; reads high below ~0.3 ms)
path = one("grbm/**/*_counter_collection.csv")
if path:
    clk = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "pvs::" in r["Kernel_Name"] and r.get("Counter_Name") == "GRBM_GUI_ACTIVE" and "End_Timestamp" in r:
            ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if ns > 300000:
                clk[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) / 8.0 / ns)
    for k, v in clk.items():
        v.sort()
        out.setdefault(k, {})["effective_clock_GHz"] = {"n": len(v), "min": round(v[0], 3), "median": round(v[len(v) // 2], 3), "max": round(v[-1], 3)}
if out:
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES (separate passes) "
                          "--output-format csv -- python3 " + command,
               "note": "FETCH_SIZE doubled per the gfx950 correction; Infinity-Cache hits are included in these "
                       "fabric-side counters, so this is traffic beyond L2, an upper bound on HBM bytes",
               "kernels": out}, open(f"{stem}_pmc_hbm.json" if not name else f"{stem}_pmc.json", "w"), indent=1)
print("wrote", glob.glob(f"{stem}_*"))
