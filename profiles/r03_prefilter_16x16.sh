#!/bin/bash
# The prefilter on v_mfma_f32_32x32x16_f16 (default, "old") against assign16x_kernel on 16x16x32 (PVS_OPT_ASSIGN_PREFILTER = 4, "new"),
# alternating on one box
set -e -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q -k "prefilter or assign or vlad or encode or kmeans or fused" > gpurun_out/p16_tests.log 2>&1 || { tail -30 gpurun_out/p16_tests.log; exit 1; }
tail -2 gpurun_out/p16_tests.log
for v in new old new old; do
  fl=""; [ $v = new ] && fl="--prefilter-16x16"
  for w in "vlad512 --images 32768" "vlad512 --images 32768 --desc u8" "config2"; do
    python3 bench.py --workload $w $fl --steps 5 --warmup 2 --no-cpu-baseline 2>> gpurun_out/p16.err | python3 -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = j.get('stages_ms_per_step') or {k: v['ms_avg'] for k, v in j['stages'].items()}
print('$v', '$w', j['value'], j['ms_per_step'], {k: st[k] for k in ('assign', 'aggregate') if k in st})"
  done
done
