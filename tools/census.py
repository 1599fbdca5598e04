#!/usr/bin/env python3
"""Scheduler census of the render kernel on the bench workload (GPU): lanes per NODE/PRIM/SHADE step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes
w, h, spp = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1920, 1080, 16)))
spec = (scenes.c4_blob63k if os.environ.get("CENSUS_WORKLOAD") == "c4v" else scenes.c4_shipped)(width=w, height=h, spp=spp)
g = api.GpuScene(spec, n_threads=8)
res, st = g.render(want=("color",), collect_stats=1)
rays = max(1, st.rays_closest + st.rays_shadow)      # 0 with the timers-only build (-DRT_TIMERS): only the time split is meaningful
print(f"rays {rays} closest {st.rays_closest} shadow {st.rays_shadow} shaded {st.shaded_hits} ms {st.ms_render:.2f}")
print(f"NODE : {st.sched_node_steps} wave-steps, {st.nodes_visited} lane-steps, {st.nodes_visited / max(1, st.sched_node_steps):.1f} lanes/step, {st.nodes_visited / rays:.2f} per ray")
pt = st.tri_tests + st.sphere_tests + st.plane_tests
print(f"PRIM : {st.sched_prim_steps} wave-steps, {pt} lane-steps, {pt / max(1, st.sched_prim_steps):.1f} lanes/step, {pt / rays:.2f} per ray")
print(f"SHADE: {st.sched_shade_steps} wave-steps, {rays} lane-steps, {rays / max(1, st.sched_shade_steps):.1f} lanes/step")
print(f"wave-steps per 64 rays: node {64 * st.sched_node_steps / rays:.1f} prim {64 * st.sched_prim_steps / rays:.1f} shade {64 * st.sched_shade_steps / rays:.1f}")
steps = st.sched_node_steps + st.sched_prim_steps + st.sched_shade_steps
print(f"live lanes per trip: {st.sched_live_lanes / max(1, steps):.1f} of 64")
print(f"node visits served from LDS: {st.top_node_hits / max(1, st.nodes_visited):.3f}")
tot = st.cyc_vote + st.cyc_node + st.cyc_prim + st.cyc_shade + st.cyc_refill
if tot:
    steps = max(1, steps)
    for n, c, k in (("vote", st.cyc_vote, steps), ("node", st.cyc_node, st.sched_node_steps), ("prim", st.cyc_prim, st.sched_prim_steps), ("shade", st.cyc_shade, st.sched_shade_steps), ("refill", st.cyc_refill, st.sched_shade_steps)):
        print(f"time in {n:6s}: {100 * c / tot:5.1f} %   {c / max(1, k):8.0f} ticks per trip")
names = ("hit", "light term", "shadow set-up", "scatter", "finish")
for k, n in enumerate(names):
    print(f"SHADE {n:14s}: {st.shade_lanes[k]} lanes in {st.shade_trips[k]} trips = {st.shade_lanes[k] / max(1, st.shade_trips[k]):.1f} lanes/trip, in {st.shade_trips[k] / max(1, st.sched_shade_steps):.3f} of the SHADE trips")
det = list(st.cyc_shade_detail)
if sum(det):
    for n, c in zip(("loads", "hit", "light", "set-up", "scatter", "finish", "begin query"), det):
        print(f"SHADE time in {n:12s}: {100 * c / max(1, sum(det)):5.1f} %")
