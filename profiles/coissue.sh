#!/bin/bash
# MFMA / VALU co-issue on one SIMD (csrc/bench/coissue.hip): wall time + in-kernel cycles per variant, then the SQ counters
# of the same binary (counters only, no traces).   profiles/coissue.sh <tag>  -> profiles/<tag>_coissue.txt, <tag>_coissue_pmc.csv
set -e -o pipefail
tag=${1:-r02}
out=gpurun_out/coissue
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B=python-visual-similarity_amd/csrc/bench/coissue
$B > "$out/coissue.txt"
cat "$out/coissue.txt"
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$out/pmc" -- $B > "$out/pmc_run.txt" 2> "$out/pmc.log"
python3 - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for path in glob.glob(f"{out}/pmc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60], r.get("Workgroup_Size", ""))
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
with open(f"{out}/pmc_summary.csv", "w") as f:
    f.write("dispatch,kernel,wg,SQ_BUSY_CU_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,mfma_busy_frac\n")
    for (d, k, wg), c in sorted(rows.items()):
        b = c.get("SQ_BUSY_CU_CYCLES", 0.0)
        f.write(f"{d},{k},{wg},{b:.0f},{c.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.0f},{c.get('SQ_INSTS_VALU',0):.0f},"
                f"{c.get('SQ_ACTIVE_INST_VALU',0):.0f},{(c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*b) if b else 0):.4f}\n")
print(open(f"{out}/pmc_summary.csv").read())
PY
