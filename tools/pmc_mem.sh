#!/bin/bash
# Vector-memory pipeline counters (TA / TCP / TD) + GRBM for the default library or $2.
set -u
TAG=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmcm_$TAG; mkdir -p "$OUT"
[ $# -ge 2 ] && export RTCORE_LIB=$R/$2
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-rr --no-reference-count --others= --steps 1 --warmup 0"
run() { n=$1; shift; timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$n" -o $n -- $BENCH > "$OUT/$n.log" 2>&1 || { echo "pass $n failed"; tail -5 "$OUT/$n.log"; exit 1; }; }
run a TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE GRBM_COUNT
run b TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run c TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run d TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_render_tiles<float, false" in r["Kernel_Name"]:   # the timed variants: <float, false, true|false>
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
if not agg:
    sys.exit("no counter row matched k_render_tiles<float, false...>: the kernel was renamed?")
for k in sorted(agg): print(f"{k:44s} {agg[k]:.4e}")
PY
