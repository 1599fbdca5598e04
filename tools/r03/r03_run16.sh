set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 280 --timeout-method thread > gpurun_out/r03_gpu_i.log 2>&1 || { tail -40 gpurun_out/r03_gpu_i.log; exit 1; }
tail -1 gpurun_out/r03_gpu_i.log
for w in c4 c2; do bash tools/ab_variants.sh --workload $w | sed "s/^/$w /"; done
