set -e -o pipefail
mkdir -p gpurun_out
python3 bench.py --workload serve > gpurun_out/serve.json 2> gpurun_out/serve.err
python3 tests/tools/index_latency.py 2> /dev/null > gpurun_out/index_latency.txt
tail -c 900 gpurun_out/serve.json; cat gpurun_out/index_latency.txt
