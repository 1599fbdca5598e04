#!/bin/bash
# Counter passes for any bench workload / library:  tools/pmc_any.sh <tag> <lib.so|-> "<bench args>" "<counters pass 1>" "<counters pass 2>" ...
set -u
TAG=$1; LIB=$2; ARGS=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmca_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ "$LIB" != "-" ] && export RTCORE_LIB=$R/$LIB
BENCH="python3 $R/bench.py --no-cpu-baseline --no-rr --no-reference-count --others= --steps 1 --warmup 0 $ARGS"
n=0
for pass in "$@"; do
  n=$((n+1))
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d "$OUT/p$n" -o p$n -- $BENCH > "$OUT/p$n.log" 2>&1 || { echo "pass $n failed"; tail -5 "$OUT/p$n.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); cnt = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_render_tiles<float, false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg): print(f"{k:40s} {agg[k] / max(1, len(cnt[k])):.4e} per launch ({len(cnt[k])} launches)")
PY
