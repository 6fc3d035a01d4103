P=python-visual-similarity_amd/pvsim
run() { python bench.py --workload vlad512 --images 32768 --no-cpu-baseline $1 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('  vlad512 $1', d['ms_per_step'], d['stages_ms_per_step'])"; }
hl() { python bench.py --no-cpu-baseline --steps 8 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('  headline', d['ms_per_step'], 'agg', d['stages']['aggregate']['ms_avg'], 'assign', d['stages']['assign']['ms_avg'], 'gemm', d['stages']['cosine_gemm']['ms_avg'])"; }
for round in 1 2; do
for v in A B; do
  cp $P/lib$v.so $P/libpvsim_hip.so
  echo "== variant $v (round $round)"
  run ""; run "--desc u8"; hl
done
done
cp $P/libB.so $P/libpvsim_hip.so
