#!/bin/bash
# round 3, first GPU pass: new f64 tests, the 8-phase fp16 GEMM control beside the shipped schedules, fisher f64 line, headline
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/s1_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s1_pytest.log
tail -5 gpurun_out/s1_pytest.log
cd python-visual-similarity_amd/csrc/bench
timeout -k 10 300 ./gemm_variants 8192 32768 0 3 3 > ../../../gpurun_out/s1_gemm8_32768.log 2>&1; echo "gemm rc=$?"
timeout -k 10 200 ./gemm_variants 8192 8192 0 3 3 > ../../../gpurun_out/s1_gemm8_8192.log 2>&1; echo "gemm rc=$?"
cd ../../..
cat gpurun_out/s1_gemm8_32768.log gpurun_out/s1_gemm8_8192.log
timeout -k 10 400 python3 bench.py --workload fisher --retrieval f64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/s1_fisher_f64.json 2> gpurun_out/s1_fisher_f64.err; echo "fisher f64 rc=$?"
tail -c 1500 gpurun_out/s1_fisher_f64.json; tail -3 gpurun_out/s1_fisher_f64.err
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err; echo "bench rc=$?"
tail -c 1200 gpurun_out/s1_bench.json
