#!/bin/bash
# traffic beyond L2 of the Fisher moments kernel in its two forms (second pass / norm division in the kernel)
set -e -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
for sc in 1 2; do
  rm -rf gpurun_out/fs_$sc
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fs_$sc/fetch -- python3 bench.py --workload fisher --fisher-scale $sc --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/fs_$sc.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/fs_$sc/write -- python3 bench.py --workload fisher --fisher-scale $sc --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> gpurun_out/fs_$sc.log
  python3 - <<PY
import csv, glob, collections
for what in ("fetch", "write"):
    f = glob.glob("gpurun_out/fs_$sc/%s/**/*counter_collection.csv" % what, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fisher" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:50]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        big = [x for x in v if x > 0.5 * max(v)]
        print("scale option $sc", what, k, "launches", len(v), "large launches: mean KiB", round(sum(big) / len(big)), "=> GB", round(sum(big) / len(big) * 1024 * (2 if what == "fetch" else 1) / 1e9, 2))
PY
done
