set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/calib/f -o f -- $R/tools/calib/fetch_calib 8 16 > $R/gpurun_out/calib_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/calib/e -o e -- $R/tools/calib/fetch_calib 8 16 > $R/gpurun_out/calib_ea.log 2>&1 || exit 1
grep "records of" $R/gpurun_out/calib_fetch.log
cd $R && python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/calib/*/*_counter_collection.csv")):
    agg = {}
    for r in csv.DictReader(open(f)):
        if "k_gather" in r["Kernel_Name"]:
            agg.setdefault((r["Kernel_Name"].split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("calib", k[0], k[1], "per launch %.4e" % (sum(v) / len(v)), "launches", len(v))
PY
