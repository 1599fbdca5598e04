set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_planes.py tests/test_gpu_spheres.py tests/test_gpu_edge.py tests/test_gpu_fuzz_oracle.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r03_gpu_b.log 2>&1 || { tail -30 gpurun_out/r03_gpu_b.log; exit 1; }
tail -2 gpurun_out/r03_gpu_b.log
bash tools/ab_variants.sh --workload c4 && cp gpurun_out/ab_variants.txt gpurun_out/r03_ab_nt_c4.txt
bash tools/ab_variants.sh --workload c5 --spp 256 --width 3840 --height 2160 && cp gpurun_out/ab_variants.txt gpurun_out/r03_ab_nt_c5.txt
