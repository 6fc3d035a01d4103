#!/bin/bash
# The N > 1 default at its FULL size (10^6 images, 65536 queries) with two ranks sharing ONE GPU, collectives staged through the
# host (gloo REHEARSAL): sizes, 64-bit indexing and the traveling-query plan end to end; not a measurement.
set -e -o pipefail
mkdir -p gpurun_out
( time PVS_BENCH_BACKEND=gloo python3 bench.py --gpus 4 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/full_rehearsal4.json 2> gpurun_out/full_rehearsal4.err ) 2> gpurun_out/full_rehearsal4.time
tail -3 gpurun_out/full_rehearsal4.err
cat gpurun_out/full_rehearsal4.time
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/full_rehearsal4.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("n_gpus", "ms_per_step", "queries_per_step", "exchange_plan", "self_check", "backend")})
print(j["config"]["workload"], j["corpus_build"])
PY
