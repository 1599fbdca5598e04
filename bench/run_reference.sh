#!/bin/bash
# Pins the oracle against the REAL reference on a box that has GHC + cabal (this build image has neither):
# renders the repository's synthetic configs with the Haskell binary exactly as its README says
# (`cabal run raytracer <config.json> +RTS -N -qg -A32M`, README.md:10; output out/<timestamp>.ppm, Main.hs:48-58) and
# compares each frame with the same config rendered by the CPU oracle (oracle/liboracle.so through tools/oracle_ppm.py)
# and, when a GPU is present, by rtcli.
#
#   bench/run_reference.sh /path/to/omeldar-raytracer-checkout [workdir]
#
# Deterministic configs (antialiasing off, maxBounces <= 1: the reference draws no random number that matters)
# must agree to <= 1 level on >= 99.99 % of the 8-bit values; stochastic configs are compared in the mean image (the
# reference seeds from OS entropy): PSNR >= 30 dB at the sample counts below.  The time the reference prints
# ("Time taken to create and write file", Main.hs:58) is recorded beside each frame: that is the CPU baseline
# BASELINE.json asks for (kind "reference").
set -euo pipefail
REF=${1:?usage: run_reference.sh <reference checkout> [workdir]}
WORK=${2:-$(pwd)/gpurun_out/reference_run}
REPO=$(cd "$(dirname "$0")/.." && pwd)
command -v cabal >/dev/null || { echo "cabal not found: this script needs a box with GHC >= 9.4 and cabal"; exit 3; }
mkdir -p "$WORK"
cd "$REPO"
make -s host oracle
# name | scenes.py constructor call | deterministic?
CASES=(
  "c1_det|c1_testing(width=640, height=360, spp=1)|det"
  "c3_det|c3_blob988(width=480, height=270, spp=1, antialiasing=False, max_bounces=1)|det"
  "c2_mean|c2_spheres(width=320, height=180, spp=256)|mean"
  "c3_mean|c3_blob988(width=320, height=180, spp=256)|mean"
  "c4_mean|c4_blob63k(width=320, height=180, spp=256)|mean"
)
# RT_REF_CASES="name|ctor|kind;name|ctor|kind": another case list (tests/test_reference_hook.py feeds the script a fake `cabal`)
if [ -n "${RT_REF_CASES:-}" ]; then IFS=';' read -r -a CASES <<< "$RT_REF_CASES"; fi
status=0
for c in "${CASES[@]}"; do
  IFS='|' read -r name ctor kind <<< "$c"
  cfg=$(python3 - <<PY
from raytracer_amd import scenes
spec = scenes.$ctor
spec.name = "$name"
print(scenes.write_reference_files(spec, "$WORK"))
PY
)
  echo "== $name: reference"
  ( cd "$REF" && mkdir -p out && rm -f out/*.ppm && cabal run -v0 raytracer "$cfg" +RTS -N -qg -A32M -RTS | tee "$WORK/$name.reference.log" )
  cp "$(ls -t "$REF"/out/*.ppm | head -1)" "$WORK/$name.reference.ppm"
  echo "== $name: oracle"
  python3 tools/oracle_ppm.py "$cfg" "$WORK/$name.oracle.ppm"
  if [ "$kind" = det ]; then python3 tools/ppm_diff.py "$WORK/$name.reference.ppm" "$WORK/$name.oracle.ppm" --deterministic || status=1
  else python3 tools/ppm_diff.py "$WORK/$name.reference.ppm" "$WORK/$name.oracle.ppm" --psnr 30 || status=1; fi
  if [ -x raytracer_amd/bin/rtcli ] && [ -e /dev/kfd ]; then
    echo "== $name: rtcli (GPU)"
    ( cd "$WORK" && "$REPO/raytracer_amd/bin/rtcli" "$cfg" > "$name.rtcli.log" && cp "$(ls -t out/*.ppm | head -1)" "$name.gpu.ppm" )
    python3 tools/ppm_diff.py "$WORK/$name.oracle.ppm" "$WORK/$name.gpu.ppm" $([ "$kind" = det ] && echo --deterministic || echo "--psnr 30") || status=1
  fi
done
grep -H "Time taken" "$WORK"/*.reference.log || true
exit $status
