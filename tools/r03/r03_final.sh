#!/bin/bash
# Round 3's final evidence run on the GPU box (through gpurun): tests, the default bench line, rocprofv3 records of C4 and C5,
# the FETCH_SIZE calibration, per-rank timings, the single-process group line.  Outputs under gpurun_out/; tools/summarize_profile.py
# and a copy step on the CPU side move what is judged into profiles/.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 280 --timeout-method thread > gpurun_out/r03_final_gpu.log 2>&1 || { tail -40 gpurun_out/r03_final_gpu.log; exit 1; }
tail -1 gpurun_out/r03_final_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1 || { tail gpurun_out/r03_smoke.log; exit 1; }
tail -1 gpurun_out/r03_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { tail -20 gpurun_out/r03_bench.err; exit 1; }
echo "bench ok"
bash tools/profile.sh r03 --workload c4 > gpurun_out/r03_profile_c4.log 2>&1 || { tail -20 gpurun_out/r03_profile_c4.log; exit 1; }
bash tools/profile.sh r03c5 --workload c5 > gpurun_out/r03_profile_c5.log 2>&1 || { tail -20 gpurun_out/r03_profile_c5.log; exit 1; }
echo "profiles ok"
( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/calib/f -o f -- $GRAFT_REPO_ROOT/tools/calib/fetch_calib 8 16 > $GRAFT_REPO_ROOT/gpurun_out/calib_fetch.log 2>&1 \
  && timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/calib/e -o e -- $GRAFT_REPO_ROOT/tools/calib/fetch_calib 8 16 > $GRAFT_REPO_ROOT/gpurun_out/calib_ea.log 2>&1 ) || { tail -5 gpurun_out/calib_fetch.log gpurun_out/calib_ea.log; exit 1; }
grep "records of 64" gpurun_out/calib_fetch.log | tail -1
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/calib/*/*_counter_collection.csv")):
    agg = {}
    for r in csv.DictReader(open(f)):
        if "k_gather64" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("calib", k, "per launch", sum(v) / len(v), "launches", len(v))
PY
timeout -k 10 300 python tools/rank_timing.py c4 > gpurun_out/r03_rank_timing_c4.txt 2>&1 || { tail gpurun_out/r03_rank_timing_c4.txt; exit 1; }
tail -4 gpurun_out/r03_rank_timing_c4.txt
timeout -k 10 200 python bench.py --gpus 2 --single-process --devices 0,0 --steps 5 --warmup 2 > gpurun_out/r03_bench_group2.json 2> gpurun_out/r03_bench_group2.err || { tail gpurun_out/r03_bench_group2.err; exit 1; }
cat gpurun_out/r03_bench_group2.json
