set -e -o pipefail
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q -k "fisher or gmm or Fisher or learn or em_" > gpurun_out/fold_tests.log 2>&1; tail -2 gpurun_out/fold_tests.log
for i in 1 2; do
python3 bench.py --workload fisher --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/f32.json 2>> gpurun_out/fold.err
python3 bench.py --workload fisher --retrieval f64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/f64.json 2>> gpurun_out/fold.err
python3 - <<PY
import json
for t in ("f32", "f64"):
    j = json.loads(open("gpurun_out/%s.json" % t).read().strip().splitlines()[-1])
    print(t, j["ms_per_step"], j["stages_ms_per_step"])
PY
done
python3 bench.py --workload learn --no-cpu-baseline > gpurun_out/learn.json 2> gpurun_out/learn.err; python3 -c "
import json; j=json.loads(open('gpurun_out/learn.json').read().strip().splitlines()[-1]); print('learn', j['ms_per_step'], j.get('phases') or j.get('stages_ms_per_step'))"
