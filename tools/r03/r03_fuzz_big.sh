# usage: bash tools/r03/r03_fuzz_big.sh <family flag or ""> <first> <last+1>  (one family per gpurun call, <= 18 minutes)
set -o pipefail
mkdir -p gpurun_out
fam=$1; a=$2; b=$3
timeout -k 10 1100 python -u tools/fuzz_oracle.py $fam $a $b > gpurun_out/r03_fzbig_${fam#--}_$a.log 2>&1; rc=$?
echo "family '$fam' $a..$b rc=$rc: $(tail -1 gpurun_out/r03_fzbig_${fam#--}_$a.log); lines $(wc -l < gpurun_out/r03_fzbig_${fam#--}_$a.log)"
exit 0
