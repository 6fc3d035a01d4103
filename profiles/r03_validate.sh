#!/bin/bash
# Whole GPU suite + smoke + a randomised parity sweep:  bash profiles/r03_validate.sh [fuzz seconds] [seed]
set -e -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/validate_tests.log 2>&1 || { tail -40 gpurun_out/validate_tests.log; exit 1; }
tail -3 gpurun_out/validate_tests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
python3 tests/tools/fuzz_gpu.py ${1:-240} ${2:-31} > gpurun_out/validate_fuzz.log 2>&1 || { tail -20 gpurun_out/validate_fuzz.log; exit 1; }
tail -3 gpurun_out/validate_fuzz.log
