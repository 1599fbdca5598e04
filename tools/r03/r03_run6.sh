set -o pipefail
mkdir -p gpurun_out
echo "== new" > gpurun_out/r03_census.txt
timeout -k 10 200 python tools/census.py 1920 1080 16 >> gpurun_out/r03_census.txt 2>gpurun_out/census.err || { tail gpurun_out/census.err; exit 1; }
echo "== base" >> gpurun_out/r03_census.txt
RTCORE_LIB=raytracer_amd/lib/variants/base.so timeout -k 10 200 python tools/census.py 1920 1080 16 >> gpurun_out/r03_census.txt 2>gpurun_out/census.err || { tail gpurun_out/census.err; exit 1; }
cat gpurun_out/r03_census.txt
