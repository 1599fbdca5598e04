set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 280 --timeout-method thread > gpurun_out/r03_gpu_f.log 2>&1 || { tail -40 gpurun_out/r03_gpu_f.log; exit 1; }
tail -2 gpurun_out/r03_gpu_f.log
bash tools/ab_variants.sh --workload c4 && cp gpurun_out/ab_variants.txt gpurun_out/r03_ab_prims128_c4.txt
bash tools/ab_variants.sh --workload c5 && cp gpurun_out/ab_variants.txt gpurun_out/r03_ab_prims128_c5.txt
bash tools/ab_variants.sh --workload c3 && cp gpurun_out/ab_variants.txt gpurun_out/r03_ab_prims128_c3.txt
