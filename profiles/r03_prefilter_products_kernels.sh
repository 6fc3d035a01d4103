#!/bin/bash
# per-kernel times of the prefilter variants (rocprofv3 kernel trace): assign16 (fp16 products) vs assign (exact pass on the rest)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for np_ in 3 2 1; do
  rm -rf gpurun_out/pfk_$np_
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pfk_$np_ -- python3 bench.py --workload vlad512 --images 32768 --no-cpu-baseline --steps 3 --warmup 1 --prefilter-products $np_ > /dev/null 2>&1
  echo "products $np_"
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/pfk_$np_/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "assign" in r["Name"] or "aggregate" in r["Name"]:
        print("   %-60s calls %4s  avg %9.3f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/pfk_$np_
done
