set -o pipefail
mkdir -p gpurun_out
for w in 16 12 8; do
  RT_WAVES_PER_BLOCK=$w timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || exit 1
  echo "RT_WAVES_PER_BLOCK=$w $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/ab_tmp.json | head -1)"
done | tee gpurun_out/r03_occupancy.txt
RTCORE_LIB=raytracer_amd/lib/variants/w3.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || exit 1
echo "w3.so (168 VGPRs, 12 waves/CU) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/ab_tmp.json | head -1)" | tee -a gpurun_out/r03_occupancy.txt
