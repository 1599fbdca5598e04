#!/bin/bash
# Front-end counters (instruction fetch, branches, scalar cache) for the default library or $2.
set -u
TAG=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmcf_$TAG; mkdir -p "$OUT"
[ $# -ge 2 ] && export RTCORE_LIB=$R/$2
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-rr --no-reference-count --others= --steps 1 --warmup 0"
timeout -s KILL 300 rocprofv3 --pmc SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d "$OUT/a" -o a -- $BENCH > "$OUT/a.log" 2>&1 || { tail -5 "$OUT/a.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL --output-format csv -d "$OUT/b" -o b -- $BENCH > "$OUT/b.log" 2>&1 || { tail -5 "$OUT/b.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d "$OUT/c" -o c -- $BENCH > "$OUT/c.log" 2>&1 || { tail -5 "$OUT/c.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_render_tiles<float, false" in r["Kernel_Name"]:   # the timed variants: <float, false, true|false>
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
if not agg:
    sys.exit("no counter row matched k_render_tiles<float, false...>: the kernel was renamed?")
for k in sorted(agg): print(f"{k:28s} {agg[k]:.4e}")
PY
