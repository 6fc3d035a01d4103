set -e -o pipefail
mkdir -p gpurun_out
bash profiles/collect.sh r03 fisher --workload fisher > gpurun_out/collect_fisher.log 2>&1; echo fisher done
bash profiles/collect.sh r03 fisher_f64 --workload fisher --retrieval f64 > gpurun_out/collect_fisher_f64.log 2>&1; echo fisher f64 done
cp profiles/r03_fisher* gpurun_out/
