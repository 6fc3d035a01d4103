#!/bin/bash
# Collects the rocprofv3 evidence behind a bench.py line on an MI355X box:
#   profiles/collect.sh <tag> [name] [bench.py arguments...]      (run from the repo root, e.g. through gpurun)
#   profiles/collect.sh r02                                   -> profiles/r02_kernel_stats.csv, r02_pmc_hbm.json, r02_bench*.json (headline)
#   profiles/collect.sh r02 vlad512 --workload vlad512 --images 32768
#                                                             -> profiles/r02_vlad512_kernel_stats.csv, r02_vlad512_pmc.json, r02_vlad512_bench.json
# Separate passes of the same command, as the MI355X guide prescribes: kernel trace + stats, then FETCH_SIZE, then WRITE_SIZE,
# then the MFMA-busy counters (counters never together with traces; the program itself follows `--`).
set -e -o pipefail
tag=${1:-r02}
name=${2:-}
shift || true
shift || true
out=gpurun_out/prof${name:+_$name}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
# the headline is profiled under the DRIVER's own arguments (--steps 20 --warmup 5); side workloads with 3 + 1 steps
if [ -z "$name" ]; then SW="--steps 20 --warmup 5"; else SW="--steps ${PVS_COLLECT_STEPS:-3} --warmup ${PVS_COLLECT_WARMUP:-1}"; fi
ARGS="$SW --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py $ARGS > "$out/bench_under_rocprof.json" 2> "$out/stats.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py $ARGS > /dev/null 2> "$out/fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py $ARGS > /dev/null 2> "$out/write.log"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$out/mfma" -- python3 bench.py $ARGS > /dev/null 2> "$out/mfma.log"
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/grbm" -- python3 bench.py $ARGS > /dev/null 2> "$out/grbm.log"
python3 profiles/summarize.py "$out" "$tag" "$name" "bench.py $ARGS"
if [ -z "$name" ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.log"     # exactly what the driver types
  cp "$out/bench.json" profiles/${tag}_bench.json
  cp "$out/bench_under_rocprof.json" profiles/${tag}_bench_under_rocprof.json
else
  python3 bench.py $SW --no-cpu-baseline $* > "$out/bench.json" 2> "$out/bench.log"
  cp "$out/bench.json" profiles/${tag}_${name}_bench.json
fi
tail -c 2500 "$out/bench.json"
