set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_rr.py tests/test_gpu_fuzz_oracle.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_gpu_d.log 2>&1 || { tail -30 gpurun_out/r03_gpu_d.log; exit 1; }
tail -1 gpurun_out/r03_gpu_d.log
bash tools/ab_variants.sh --workload c4 && cp gpurun_out/ab_variants.txt gpurun_out/r03_ab_defer2_c4.txt
