set -o pipefail
mkdir -p gpurun_out
RT_DEBUG=1 RTCORE_LIB=raytracer_amd/lib/variants/w5.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
grep "threads/workgroup" gpurun_out/ab_tmp.err | tail -1
echo "w5.so (96 VGPRs, 20 waves/CU, 3 stack rows in LDS) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/ab_tmp.json | head -1)" | tee -a gpurun_out/r03_occupancy.txt
