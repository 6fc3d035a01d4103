#!/bin/bash
# Fisher lines with the norm division as a second pass (0, default) and inside the moments kernel (2), alternating on one box:
#   bash profiles/r03_fisher_fold.sh
set -e -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q -k "fisher or gmm or Fisher or learn or em_" > gpurun_out/fold_tests.log 2>&1; tail -2 gpurun_out/fold_tests.log
for sc in 0 2 0 2; do
  python3 bench.py --workload fisher --fisher-scale $sc --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/fold_f32_$sc.json 2>> gpurun_out/fold.err
  python3 bench.py --workload fisher --retrieval f64 --fisher-scale $sc --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/fold_f64_$sc.json 2>> gpurun_out/fold.err
  python3 - <<PY
import json
for t in ("f32", "f64"):
    j = json.loads(open("gpurun_out/fold_%s_$sc.json" % t).read().strip().splitlines()[-1])
    print("scale option $sc", t, j["ms_per_step"], j["stages_ms_per_step"])
PY
done
