#!/bin/bash
# SQ counters of the two encode kernels (assign16, vlad_aggregate), one rocprofv3 pass per counter group (counters only, no traces):
#   profiles/pmc_encode.sh      -> gpurun_out/pmc_encode/summary.json
set -e -o pipefail
out=gpurun_out/pmc_encode
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python3 $CMD > /dev/null 2> "$out/p$i.log"
  echo "pass $i done"
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"{out}/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "assign16" in n or "vlad_aggregate" in n or "assign_kernel" in n:
            acc[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
