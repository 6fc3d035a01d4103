set -e -o pipefail
mkdir -p gpurun_out
( time python3 bench.py --gpus 1 --workload corpus1m --retrieval f16 --total-queries 4096 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/t_c1m_f16.json 2> gpurun_out/t_c1m_f16.err ) 2> gpurun_out/t_c1m_f16.time
cat gpurun_out/t_c1m_f16.time
( time python3 bench.py --gpus 1 --workload corpus1m --retrieval filtered --queries 2048 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/t_c1m_fil.json 2> gpurun_out/t_c1m_fil.err ) 2> gpurun_out/t_c1m_fil.time
cat gpurun_out/t_c1m_fil.time
