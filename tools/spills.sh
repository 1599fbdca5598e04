#!/bin/bash
# Register / spill metadata of the timed kernel instantiations (compile only; no GPU needed):  tools/spills.sh [extra hipcc flags]
out=$(mktemp /tmp/rtk.XXXXXX.s)
hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -DRT_MIN_WAVES=4 "$@" --cuda-device-only -S raytracer_amd/csrc/rtcore.hip -o $out || exit 1
python3 - "$out" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(_Z14k_render_tiles\w+)\s*\n(.*?)\.wavefront_size", t, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", body)
    vals = {k: int(g(k).group(1)) for k in ("private_segment_fixed_size", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count") if g(k)}
    tmpl = re.search(r"I([fd])((?:Lb[01]E)+)", name)
    args = ("float" if tmpl.group(1) == "f" else "double") + "," + ",".join("1" if b == "1" else "0" for b in re.findall(r"Lb([01])E", tmpl.group(2)))
    print(f"{args:18s} scratch {vals.get('private_segment_fixed_size'):4d}  sgpr_spill {vals.get('sgpr_spill_count'):4d}  vgpr {vals.get('vgpr_count'):4d}  vgpr_spill {vals.get('vgpr_spill_count'):4d}")
PY
rm -f $out
