#!/bin/bash
# How much does a cheaper assignment prefilter buy?  Same labels in every variant (rows the margin does not settle go to the exact
# kernel); what changes is the fp16 MFMA work per row and the share of rows left to the exact f32 kernel.
cd "$(dirname "$0")/.."
for np_ in 3 2 1; do
  python3 bench.py --workload vlad512 --images 32768 --no-cpu-baseline --prefilter-products $np_ 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vlad512 f32   products', d['config']['prefilter_products'], 'images/s', d['value'], 'ms', d['ms_per_step'], d['stages_ms_per_step'])"
  python3 bench.py --workload vlad512 --images 32768 --desc u8 --no-cpu-baseline --prefilter-products $np_ 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vlad512 u8    products', d['config']['prefilter_products'], 'images/s', d['value'], 'ms', d['ms_per_step'], d['stages_ms_per_step'])"
  python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --prefilter-products $np_ 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config2 f32   products', $np_, 'images/s', d['value'], 'ms', d['ms_per_step'], {k:v['ms_avg'] for k,v in d['stages'].items()})"
done
