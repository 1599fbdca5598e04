set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 280 --timeout-method thread > gpurun_out/r03_gpu_e.log 2>&1 || { tail -40 gpurun_out/r03_gpu_e.log; exit 1; }
tail -2 gpurun_out/r03_gpu_e.log
timeout -k 10 600 python bench.py > gpurun_out/r03_bench_full.json 2> gpurun_out/r03_bench_full.err || { tail -20 gpurun_out/r03_bench_full.err; exit 1; }
tail -c 6000 gpurun_out/r03_bench_full.json
