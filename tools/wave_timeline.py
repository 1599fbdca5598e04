#!/usr/bin/env python3
"""Per-wave timeline of one launch (needs the -DRT_TIMERS build: RTCORE_LIB=raytracer_amd/lib/variants/timers.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
os.environ["RT_WAVE_DUMP"] = "/tmp/waves.bin"
from raytracer_amd import api, scenes
spec = scenes.c4_blob63k(spp=spp)
g = api.GpuScene(spec, n_threads=8)
for _ in range(2):
    res, st = g.render(want=("color",))
raw = np.fromfile("/tmp/waves.bin", dtype=np.uint64)
d = raw[: len(raw) - 2 * 262144].reshape(-1, 4).astype(np.float64)
t0 = d[:, 0].min()
start, exh, end, chunks = (d[:, 0] - t0) / 1e5, (d[:, 1] - t0) / 1e5, (d[:, 2] - t0) / 1e5, d[:, 3]
print(f"spp {spp}: kernel {st.ms_render:.2f} ms; waves {len(d)}")
print(f"start  ms: min {start.min():.3f} max {start.max():.3f}")
print(f"exhausted ms: min {exh.min():.3f} p10 {np.percentile(exh, 10):.3f} median {np.median(exh):.3f} p90 {np.percentile(exh, 90):.3f} max {exh.max():.3f}")
print(f"end    ms: min {end.min():.3f} p10 {np.percentile(end, 10):.3f} median {np.median(end):.3f} p90 {np.percentile(end, 90):.3f} max {end.max():.3f}")
print(f"drain (end - exhausted) ms: median {np.median(end - exh):.3f} p90 {np.percentile(end - exh, 90):.3f} max {(end - exh).max():.3f}")
print(f"chunks per wave: min {chunks.min():.0f} median {np.median(chunks):.0f} max {chunks.max():.0f}")
wg = end.reshape(-1, 16).max(axis=1)
print(f"workgroup end ms: min {wg.min():.3f} median {np.median(wg):.3f} max {wg.max():.3f}")
for q in (95, 99, 99.9):
    print(f"end p{q}: {np.percentile(end, q):.3f}")
idx = np.argsort(-end)[:12]
for i in idx:
    print(f"wave {i} (wg {i // 16}, w {i % 16}): chunks {chunks[i]:.0f} exhausted {exh[i]:.3f} end {end[i]:.3f}")
print("slow waves (> 2x median):", (end > 2 * np.median(end)).sum(), "in", len(np.unique(np.nonzero(end > 2 * np.median(end))[0] // 16)), "workgroups")
# ---- per-chunk durations: time between a wave's consecutive grabs (the last chunk of a wave ends at its `exhausted`)
nw = len(d)
ch = raw[4 * nw:].reshape(-1, 2)
n_items = spec.width * spec.height * spp
n_chunks = min(len(ch), (n_items + 1023) // 1024)
ch = ch[:n_chunks]
who, when = ch[:, 0].astype(np.int64), (ch[:, 1].astype(np.float64) - t0) / 1e5
dur = np.zeros(n_chunks)
for w in np.unique(who):
    ids = np.nonzero(who == w)[0]
    o = ids[np.argsort(when[ids])]
    t = np.append(when[o], exh[w])
    dur[o] = np.diff(t)
print(f"chunk duration ms: median {np.median(dur):.3f} p90 {np.percentile(dur, 90):.3f} p99 {np.percentile(dur, 99):.3f} max {dur.max():.3f}")
tiles_x = (spec.width + 7) // 8
for c in np.argsort(-dur)[:12]:
    item0 = c * 1024
    tile = item0 // (64 * spp)
    print(f"chunk {c}: {dur[c]:.3f} ms, taken at {when[c]:.3f} ms by wave {who[c]}, first tile ({tile % tiles_x}, {tile // tiles_x}) = pixel ({tile % tiles_x * 8}, {tile // tiles_x * 8})")
# cost map by tile row
rows = (np.arange(n_chunks) * 1024 // (64 * spp)) // tiles_x
for r0 in range(0, 135, 15):
    m = (rows >= r0) & (rows < r0 + 15)
    if m.any():
        print(f"tile rows {r0:3d}-{r0 + 14:3d}: chunks {m.sum():5d} mean {dur[m].mean():.3f} ms max {dur[m].max():.3f} ms")
