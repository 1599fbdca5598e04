set -o pipefail
mkdir -p gpurun_out
for v in base defer_v1 h_base; do
  echo "== $v" | tee -a gpurun_out/r03_pmc_structure.txt
  bash tools/pmc_quick.sh $v raytracer_amd/lib/variants/$v.so --workload c4 | tee -a gpurun_out/r03_pmc_structure.txt || exit 1
done
