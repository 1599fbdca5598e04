#!/usr/bin/env python3
"""Static instruction counts of k_render_tiles per body, from the gfx950 assembly of a build with -DRT_ASM_MARKERS (the markers
are comment-only inline asm statements, `; RTMARK <name>`, placed at the body boundaries in rt_render_kernel.h).
    python tools/isa_counts.py [float,false,true]  > profiles/rNN_isa_counts.txt
Counts are per WAVE-LEVEL pass through the straight-line text between two markers (static, not weighted by how often a
section runs; out-of-line functions - pow, literal_hit, tie_candidate_wins, sky_texel_exact - are listed separately)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else "float,false,true"
tmpl = {"float": "f", "double": "d"}[want.split(",")[0]] + "".join("Lb1E" if b == "true" else "Lb0E" for b in want.split(",")[1:])
if want.count(",") == 2:
    tmpl += "Lb0E"
sym = f"_Z14k_render_tilesI{tmpl}Ev6KScene6KFrame"

with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-DRT_MIN_WAVES=4", "-DRT_ASM_MARKERS",
                           "--cuda-device-only", "-S", os.path.join(ROOT, "raytracer_amd/csrc/rtcore.hip"), "-o", out])
    text = open(out).read().splitlines()


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_cmp", "v_cmpx")):
            return "valu cmp"
        if op.startswith("v_cndmask"):
            return "valu select"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            return "valu lane<->sgpr"
        if op.startswith(("v_mov", "v_accvgpr")):
            return "valu mov"
        if op.startswith(("v_min", "v_max", "v_med")):
            return "valu min/max"
        if "f64" in op:
            return "valu f64 arith"
        if "f32" in op:
            return "valu f32 arith"
        return "valu int/other"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "s branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "s load (kernarg / scalar)"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch (spill)"
    return "other"


funcs = collections.OrderedDict()
cur = None
for ln in text:
    m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", ln)
    if m and not m.group(1).startswith((".L", "BB")):
        cur = m.group(1)
        funcs[cur] = []
        continue
    if cur is not None:
        funcs[cur].append(ln)

body = funcs.get(sym)
if body is None:
    sys.exit(f"{sym} not found; kernels: {[f for f in funcs if 'k_render_tiles' in f]}")

sections = collections.OrderedDict()
name = "PROLOGUE (set-up, node cache fill, loop head, vote)"
sections[name] = collections.Counter()
for ln in body:
    s = ln.strip()
    m = re.match(r";\s*RTMARK (\w+)", s)
    if m:
        name = "after " + m.group(1)
        sections.setdefault(name, collections.Counter())
        continue
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        continue
    op = s.split()[0]
    if op in ("s_endpgm", "s_nop", "s_code_end"):
        continue
    sections[name][classify(op)] += 1

classes = ["valu f32 arith", "valu f64 arith", "valu min/max", "valu cmp", "valu select", "valu mov", "valu int/other", "valu lane<->sgpr",
           "salu", "s branch", "s_waitcnt", "s load (kernarg / scalar)", "lds", "vmem", "scratch (spill)", "other"]
print(f"kernel k_render_tiles<{want}>  ({sym}); static wave-level instructions between markers")
print(f"{'section':58s} {'VALU':>6s} {'SALU':>6s} {'mem':>5s} | " + " ".join(f"{c.split()[-1][:7]:>7s}" for c in classes))
tot = collections.Counter()
for name, c in sections.items():
    valu = sum(v for k, v in c.items() if k.startswith("valu"))
    salu = sum(v for k, v in c.items() if k in ("salu", "s branch", "s_waitcnt", "s load (kernarg / scalar)"))
    mem = c["lds"] + c["vmem"] + c["scratch (spill)"]
    print(f"{name:58s} {valu:6d} {salu:6d} {mem:5d} | " + " ".join(f"{c[k]:7d}" for k in classes))
    tot.update(c)
print(f"{'TOTAL':58s} {sum(v for k, v in tot.items() if k.startswith('valu')):6d}")
print("columns after the bar:", ", ".join(classes))
print()
for f, lines in funcs.items():
    if f == sym or "k_render_tiles" in f or f.startswith("__hip") or f.startswith("_Z") is False:
        continue
    n = collections.Counter()
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        n[classify(s.split()[0]).split()[0]] += 1
    if sum(n.values()) > 20 and not f.startswith("_Z9k_") and "k_" not in f[:6]:
        print(f"out-of-line {f}: {dict(n)}")
meta = [l.strip() for l in text if re.search(r"\.(sgpr|vgpr)_(count|spill_count)|\.private_segment_fixed_size|\.name:", l)]
for i, l in enumerate(meta):
    if l.endswith(sym):
        print("metadata:", " ".join(meta[max(0, i - 6):i + 6]))
