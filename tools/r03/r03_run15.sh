set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u tools/fuzz_oracle.py 8000 10000 > gpurun_out/r03_fz_std.log 2>&1; echo "std rc=$? $(tail -1 gpurun_out/r03_fz_std.log)"
timeout -k 10 300 python -u tools/fuzz_oracle.py --extreme 4000 5000 > gpurun_out/r03_fz_ext.log 2>&1; echo "ext rc=$? $(tail -1 gpurun_out/r03_fz_ext.log)"
timeout -k 10 300 python -u tools/fuzz_oracle.py --axis 400 2400 > gpurun_out/r03_fz_axis.log 2>&1; echo "axis rc=$? $(tail -1 gpurun_out/r03_fz_axis.log)"
timeout -k 10 300 python -u tools/fuzz_oracle.py --big 330 380 > gpurun_out/r03_fz_big.log 2>&1; echo "big rc=$? $(tail -1 gpurun_out/r03_fz_big.log)"
timeout -k 10 300 python -u tools/fuzz_entry.py 700 900 > gpurun_out/r03_fz_entry.log 2>&1; echo "entry rc=$? $(tail -1 gpurun_out/r03_fz_entry.log)"
timeout -k 10 300 python -u tools/fuzz_cli.py 900 1100 > gpurun_out/r03_fz_cli.log 2>&1; echo "cli rc=$? $(tail -1 gpurun_out/r03_fz_cli.log)"
