set -o pipefail
mkdir -p gpurun_out; : > gpurun_out/r03_timers.txt
for v in new_timers base_timers; do
  echo "== $v" >> gpurun_out/r03_timers.txt
  RTCORE_LIB=raytracer_amd/lib/variants/$v.so timeout -k 10 200 python tools/census.py 1920 1080 32 2>gpurun_out/census.err | grep -i "time in\|ms " >> gpurun_out/r03_timers.txt || { tail gpurun_out/census.err; exit 1; }
done
cat gpurun_out/r03_timers.txt
