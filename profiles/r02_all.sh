#!/bin/bash
# Everything DESIGN.md quotes for round 2, from tracked scripts:  bash profiles/r02_all.sh [part]
#   part 1: headline (+ PMC passes), vlad512 two-kernel / fused / uint8 (+ PMC), fused phase profiles
#   part 2: fisher, fp16sim (+ PMC), learn
#   part 3: the 1M-image corpus on one GPU (configs[3]/[4]): filtered (exact lists) and fp16 retrieval, bench lines only
#   part 4: the full 10^6 x 10^6 fp16 ranking on one GPU
set -e -o pipefail
part=${1:-1}
if [ "$part" = "1" ]; then
  bash profiles/collect.sh r02 > gpurun_out/collect_headline.log 2>&1; echo headline done
  bash profiles/collect.sh r02 vlad512 --workload vlad512 --images 32768 > gpurun_out/collect_vlad512.log 2>&1; echo vlad512 done
  bash profiles/collect.sh r02 vlad512_fused --workload vlad512 --images 32768 --fused > gpurun_out/collect_vlad512_fused.log 2>&1; echo vlad512 fused done
  bash profiles/collect.sh r02 vlad512_u8 --workload vlad512 --images 32768 --desc u8 > gpurun_out/collect_vlad512_u8.log 2>&1; echo vlad512 u8 done
  python3 tests/tools/fused_profile.py 16384 f32 > gpurun_out/fp_f32.json 2> gpurun_out/fp_f32.err
  python3 tests/tools/fused_profile.py 16384 u8 > gpurun_out/fp_u8.json 2> gpurun_out/fp_u8.err
  python3 tests/tools/assign_profile.py 16384 f32 > gpurun_out/ap_f32.json 2> gpurun_out/ap_f32.err
  python3 tests/tools/assign_profile.py 16384 u8 > gpurun_out/ap_u8.json 2> gpurun_out/ap_u8.err
  cat gpurun_out/fp_f32.json gpurun_out/fp_u8.json gpurun_out/ap_f32.json gpurun_out/ap_u8.json
elif [ "$part" = "2" ]; then
  bash profiles/collect.sh r02 fisher --workload fisher > gpurun_out/collect_fisher.log 2>&1; echo fisher done
  bash profiles/collect.sh r02 fp16sim --workload fp16sim --images 32768 > gpurun_out/collect_fp16sim.log 2>&1; echo fp16sim done
  python3 bench.py --workload learn --no-cpu-baseline > gpurun_out/learn.json 2> gpurun_out/learn.err; cp gpurun_out/learn.json profiles/r02_learn_bench.json; echo learn done
elif [ "$part" = "3" ]; then
  python3 bench.py --workload corpus1m --images 1000000 --steps 1 --warmup 1 --retrieval filtered --queries 8192 > gpurun_out/c1m_filtered.json 2> gpurun_out/c1m_filtered.err
  cp gpurun_out/c1m_filtered.json profiles/r02_corpus1m_filtered_bench.json; tail -c 1500 gpurun_out/c1m_filtered.json
  python3 bench.py --workload corpus1m --images 1000000 --steps 1 --warmup 1 --retrieval f16 --queries 65536 > gpurun_out/c1m_f16.json 2> gpurun_out/c1m_f16.err
  cp gpurun_out/c1m_f16.json profiles/r02_corpus1m_f16_bench.json; tail -c 1500 gpurun_out/c1m_f16.json
elif [ "$part" = "4" ]; then
  # configs[4] at its real size on one GPU: every one of the 10^6 images is a query (about 65 s)
  python3 bench.py --workload corpus1m --images 1000000 --steps 1 --warmup 0 --retrieval f16 --queries 0 > gpurun_out/c1m_f16_full.json 2> gpurun_out/c1m_f16_full.err
  cp gpurun_out/c1m_f16_full.json profiles/r02_corpus1m_f16_full_bench.json; tail -c 1500 gpurun_out/c1m_f16_full.json
  python3 bench.py --workload corpus1m --images 1000000 --steps 1 --warmup 0 --retrieval filtered --queries 0 > gpurun_out/c1m_filtered_full.json 2> gpurun_out/c1m_filtered_full.err
  cp gpurun_out/c1m_filtered_full.json profiles/r02_corpus1m_filtered_full_bench.json; tail -c 1500 gpurun_out/c1m_filtered_full.json
fi
cp profiles/r02_* gpurun_out/ 2>/dev/null || true
