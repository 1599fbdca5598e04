set -o pipefail
bash tools/profile.sh r03 --workload c4 > gpurun_out/r03_profile_c4.log 2>&1 || { tail -20 gpurun_out/r03_profile_c4.log; exit 1; }
tail -3 gpurun_out/r03_profile_c4.log
bash tools/profile.sh r03c5 --workload c5 > gpurun_out/r03_profile_c5.log 2>&1 || { tail -20 gpurun_out/r03_profile_c5.log; exit 1; }
tail -3 gpurun_out/r03_profile_c5.log
