# the fuzz families in one call, on the shipped build: usage  bash tools/r03/r03_fuzz_mix.sh <first seed>   (about 18 minutes)
set -o pipefail
mkdir -p gpurun_out
a=$1
for fam in "" --degenerate --axis --nan; do
  timeout -k 10 250 python -u tools/fuzz_oracle.py $fam $a $((a + 4000)) > gpurun_out/r03_fzmix_${fam#--}_$a.log 2>&1
  echo "family '$fam' from $a: rc=$? lines $(wc -l < gpurun_out/r03_fzmix_${fam#--}_$a.log) mismatches $(grep -c MISMATCH gpurun_out/r03_fzmix_${fam#--}_$a.log)"
done
timeout -k 10 120 python -u tools/fuzz_oracle.py --big 480 500 > gpurun_out/r03_fzmix_big_480.log 2>&1; echo "big rc=$? mismatches $(grep -c MISMATCH gpurun_out/r03_fzmix_big_480.log)"
timeout -k 10 150 python -u tools/fuzz_entry.py 1100 1200 > gpurun_out/r03_fzmix_entry.log 2>&1; echo "entry rc=$?: $(tail -1 gpurun_out/r03_fzmix_entry.log)"
timeout -k 10 150 python -u tools/fuzz_cli.py 1100 1250 > gpurun_out/r03_fzmix_cli.log 2>&1; echo "cli rc=$?: $(tail -1 gpurun_out/r03_fzmix_cli.log)"
exit 0
