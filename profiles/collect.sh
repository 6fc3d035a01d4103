#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline block on an MI355X box:
#   profiles/collect.sh <tag>      (run from the repo root, e.g. through gpurun)
# Separate passes of the same command, as the MI355X guide prescribes: kernel trace + stats, then FETCH_SIZE, then
# WRITE_SIZE, then the MFMA-busy counters (counters never together with traces).  Afterwards: python3 profiles/summarize.py gpurun_out/prof <tag>
set -e -o pipefail
tag=${1:-r01}
out=gpurun_out/prof
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
CMD="bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $CMD > "$out/bench_under_rocprof.json" 2> "$out/stats.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 $CMD > /dev/null 2> "$out/fetch.log"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 $CMD > /dev/null 2> "$out/write.log"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$out/mfma" -- python3 $CMD > /dev/null 2> "$out/mfma.log"
python3 profiles/summarize.py "$out" "$tag"
cp profiles/${tag}_kernel_stats.csv profiles/${tag}_pmc_hbm.json "$out/" 2>/dev/null || true
python3 bench.py > "$out/bench.json" 2> "$out/bench.log"
tail -c 3000 "$out/bench.json"
