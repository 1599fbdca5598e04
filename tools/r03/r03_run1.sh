set -o pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_gpu_a.log 2>&1 || { tail -20 gpurun_out/r03_gpu_a.log; exit 1; }
tail -2 gpurun_out/r03_gpu_a.log
timeout -k 10 300 python -u tools/fuzz_oracle.py --axis 0 400 > gpurun_out/r03_fuzz_axis.log 2>&1; echo "axis rc=$?"; tail -1 gpurun_out/r03_fuzz_axis.log
timeout -k 10 300 python -u tools/fuzz_oracle.py 0 300 > gpurun_out/r03_fuzz_std.log 2>&1; echo "std rc=$?"; tail -1 gpurun_out/r03_fuzz_std.log
timeout -k 10 300 python -u tools/fuzz_oracle.py --extreme 0 300 > gpurun_out/r03_fuzz_ext.log 2>&1; echo "ext rc=$?"; tail -1 gpurun_out/r03_fuzz_ext.log
timeout -k 10 200 python bench.py --no-cpu-baseline --workload c2 --spp 50 > gpurun_out/r03_c2_b.json 2> gpurun_out/r03_c2_b.err
