#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   tools/profile.sh <tag> [bench args...]
# One run per pass (rocprofv3 does not split counters over passes): kernel trace + stats, then PMC passes.
# Outputs under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py turns them into profiles/<tag>_*.
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-rr --no-reference-count --others= $*"
echo "== kernel trace" && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $BENCH --steps 3 --warmup 1 > "$OUT/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.log"; exit 1; }
pass() {  # name, counters...
  local name=$1; shift
  echo "== pmc $name: $*"
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $BENCH --steps 1 --warmup 0 > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
pass l2 TCC_HIT_sum TCC_MISS_sum || exit 1
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU || exit 1
pass sq2 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
pass sq3 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F64 SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES || exit 1
find "$OUT" -name "*.csv" | head -40
echo done
