#!/bin/bash
# round 3, second GPU pass: full GPU suite with the 8-phase kernel in the product, fp16sim line
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/s2_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s2_pytest.log
tail -8 gpurun_out/s2_pytest.log
timeout -k 10 300 python3 bench.py --workload fp16sim --images 32768 --no-cpu-baseline > gpurun_out/s2_fp16sim.json 2> gpurun_out/s2_fp16sim.err; echo "fp16sim rc=$?"
tail -c 1200 gpurun_out/s2_fp16sim.json
