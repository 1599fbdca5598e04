set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_gpu_c.log 2>&1 || { tail -30 gpurun_out/r03_gpu_c.log; exit 1; }
tail -2 gpurun_out/r03_gpu_c.log
bash tools/ab_variants.sh --workload c4 && cp gpurun_out/ab_variants.txt gpurun_out/r03_ab_defer_c4.txt
