#!/bin/bash
# Quick SQ counter passes for one library variant: tools/pmc_quick.sh <tag> <lib.so> [bench args]
set -u
TAG=$1; LIB=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RTCORE_LIB=$R/$LIB
BENCH="python3 $R/bench.py --no-cpu-baseline --no-rr --no-reference-count --others= --steps 1 --warmup 0 $*"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d "$OUT/a" -o a -- $BENCH > "$OUT/a.log" 2>&1 || { tail -5 "$OUT/a.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d "$OUT/b" -o b -- $BENCH > "$OUT/b.log" 2>&1 || { tail -5 "$OUT/b.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_render_tiles<float, false" in r["Kernel_Name"]:   # the timed variants: <float, false, true|false>
            agg[(f.split("/")[-2], r["Counter_Name"])] = float(r["Counter_Value"]) + agg[(f.split("/")[-2], r["Counter_Name"])]
if not agg:
    sys.exit("no counter row matched k_render_tiles<float, false...>: the kernel was renamed?")
d = {k[1]: v for k, v in agg.items() if k[0] == "a"}; e = {k[1]: v for k, v in agg.items() if k[0] == "b"}
wc = d["SQ_WAVE_CYCLES"]
print("waves", d["SQ_WAVES"], "INSTS_VALU %.3e" % d["SQ_INSTS_VALU"], "SALU %.3e" % e["SQ_INSTS_SALU"], "VMEM_RD %.3e" % e["SQ_INSTS_VMEM_RD"], "LDS %.3e" % e["SQ_INSTS_LDS"])
print("lane_util %.3f" % (e["SQ_THREAD_CYCLES_VALU"] / 64 / e["SQ_ACTIVE_INST_VALU"]), "wait_any %.3f" % (d["SQ_WAIT_ANY"] / wc), "issue_stall %.3f" % (d["SQ_WAIT_INST_ANY"] / wc),
      "issuing %.3f" % (d["SQ_ACTIVE_INST_ANY"] / wc), "valu %.3f" % (d["SQ_ACTIVE_INST_VALU"] / wc), "f64 insts %.3e" % (e["SQ_INSTS_VALU_FMA_F64"] + e["SQ_INSTS_VALU_MUL_F64"] + e["SQ_INSTS_VALU_ADD_F64"]))
PY
