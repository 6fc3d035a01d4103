#!/bin/bash
# Everything DESIGN.md quotes for round 3, from tracked scripts:  bash profiles/r03_all.sh [part]
#   part 1: headline under the driver's arguments (+ PMC passes, per-launch spread, effective clock), fp16sim (+ PMC), fisher f32 / f64
#   part 2: vlad512 f32 / u8 (+ PMC), learn
#   part 3: the 1M-image corpus on one GPU: the N > 1 default's one-GPU base (fp16, 65536 queries), filtered (exact lists, 8192 queries)
#   part 4: the full 10^6 x 10^6 rankings on one GPU (fp16 and exact-filtered)
set -e -o pipefail
part=${1:-1}
mkdir -p gpurun_out
if [ "$part" = "1" ]; then
  bash profiles/collect.sh r03 > gpurun_out/collect_headline.log 2>&1; echo headline done
  bash profiles/collect.sh r03 fp16sim --workload fp16sim --images 32768 > gpurun_out/collect_fp16sim.log 2>&1; echo fp16sim done
  bash profiles/collect.sh r03 fisher --workload fisher > gpurun_out/collect_fisher.log 2>&1; echo fisher done
  bash profiles/collect.sh r03 fisher_f64 --workload fisher --retrieval f64 > gpurun_out/collect_fisher_f64.log 2>&1; echo fisher f64 done
elif [ "$part" = "2" ]; then
  bash profiles/collect.sh r03 vlad512 --workload vlad512 --images 32768 > gpurun_out/collect_vlad512.log 2>&1; echo vlad512 done
  bash profiles/collect.sh r03 vlad512_u8 --workload vlad512 --images 32768 --desc u8 > gpurun_out/collect_vlad512_u8.log 2>&1; echo vlad512 u8 done
  python3 bench.py --workload learn --no-cpu-baseline > gpurun_out/learn.json 2> gpurun_out/learn.err; cp gpurun_out/learn.json profiles/r03_learn_bench.json; echo learn done
elif [ "$part" = "3" ]; then
  python3 bench.py --gpus 1 --workload corpus1m --images 1000000 --retrieval f16 --total-queries 65536 --steps 2 --warmup 1 > gpurun_out/c1m_base.json 2> gpurun_out/c1m_base.err
  cp gpurun_out/c1m_base.json profiles/r03_corpus1m_base_bench.json; tail -c 1500 gpurun_out/c1m_base.json
  python3 bench.py --workload corpus1m --images 1000000 --steps 1 --warmup 1 --retrieval filtered --queries 8192 --no-cpu-baseline > gpurun_out/c1m_filtered.json 2> gpurun_out/c1m_filtered.err
  cp gpurun_out/c1m_filtered.json profiles/r03_corpus1m_filtered_bench.json; tail -c 1500 gpurun_out/c1m_filtered.json
elif [ "$part" = "4" ]; then
  python3 bench.py --workload corpus1m --images 1000000 --steps 1 --warmup 0 --retrieval f16 --queries 0 --no-cpu-baseline > gpurun_out/c1m_f16_full.json 2> gpurun_out/c1m_f16_full.err
  cp gpurun_out/c1m_f16_full.json profiles/r03_corpus1m_f16_full_bench.json; tail -c 1500 gpurun_out/c1m_f16_full.json
  python3 bench.py --workload corpus1m --images 1000000 --steps 1 --warmup 0 --retrieval filtered --queries 0 --no-cpu-baseline > gpurun_out/c1m_filtered_full.json 2> gpurun_out/c1m_filtered_full.err
  cp gpurun_out/c1m_filtered_full.json profiles/r03_corpus1m_filtered_full_bench.json; tail -c 1500 gpurun_out/c1m_filtered_full.json
fi
cp profiles/r03_* gpurun_out/ 2>/dev/null || true
