#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric on MI355X: Mrays/s and frame time of the per-pixel ray<->scene
hot path on the 63,432-triangle scene at 1920x1080, 100 spp (config C4), N = 1/2/4/8 GPUs.

One step = one frame: every rank renders its interleaved 8x8 tiles from its HBM-resident scene copy and quantises them
to the 8-bit frame in the same pass; the packed 8-bit tiles are gathered on rank 0 over RCCL and un-tiled in one launch
(raytracer_amd/dist.py).  A ray = one BVH query (primary/bounce closest-hit + shadow any-hit), counted on the device by the
stats variant of the kernel in an untimed pass (counts are deterministic for a seed).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)
    python bench.py --gpus N --single-process [--devices 0,0,..]   the same metric through rt_init_devices + ONE blocking
                                                            rt_render per frame (what the Haskell host binds; ids may repeat)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`; at N = 1 the line also
carries `rr` (the opt-in Russian roulette extension BASELINE config 4 names) and `other_workloads` (C2 / C3 / C5 from the
same binary).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NODE_BYTES = 64      # DNode<float>: both child boxes (FP32, conservative) + 2 refs
PRIM_BYTES = 80      # DPrim: 9 doubles + material + kind
MAT_BYTES = 64       # DMaterial fetched per shaded hit
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
# Vector (non-MFMA) peaks of MI355X, MI355X_MICROARCH.md / SURVEY section 7: 157.3 TFLOP/s FP32 (packed), 78.6 TFLOP/s FP64
VALU_PEAK_TFLOPS_F32 = 157.3
VALU_PEAK_TFLOPS_F64 = 78.6
# SURVEY 8(d)'s flop weights of the algorithm: 25 per box test (2 per visited node; FP32 here), 30 + 15 + 6 per triangle
# test that runs to t (the kernel's predicated test always does; FP64), 25 per sphere test, ~60 per shaded hit + 33 for the
# accepted hit's normal / face / point, ~40 per sky lookup (one per path when a sky texture is set)
FLOPS_NODE, FLOPS_TRI, FLOPS_SPHERE, FLOPS_PLANE, FLOPS_SHADED, FLOPS_SKY = 50, 51, 25, 20, 93, 40
# scenes up to this size live in L2 (8 x 4 MB) / Infinity Cache (256 MB): their roofline is the VALU one
CACHE_RESIDENT_BYTES = 64 << 20
RR_START = 2          # the roulette block of the line: rt_params.rr_start_bounce


def make_spec(workload, width=None, height=None, spp=None):
    from raytracer_amd import scenes
    mk = {"c4": scenes.c4_shipped, "c4v": scenes.c4_blob63k, "c3": scenes.c3_blob988, "c2": scenes.c2_spheres, "c5": scenes.c5_multi}[workload]
    kw = {k: v for k, v in (("width", width), ("height", height), ("spp", spp)) if v is not None}
    return mk(**kw)


class Ctx:
    """torch / torch.distributed plumbing of one rank."""

    def __init__(self, n_gpus):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run --nproc-per-node {n_gpus}")
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        # RT_BENCH_FORCE_DIST=1: take the RCCL code path (process group, all-reduce, gather of device tensors, barrier) even with
        # one rank - the one-GPU test box's rehearsal of what the 8-GPU run executes (tests/test_gpu_dist.py)
        self.use_dist = self.world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)

    def sum_ints(self, values):
        t = self.torch.tensor(values, dtype=self.torch.int64, device=self.dev)
        if self.use_dist:
            self.dist.all_reduce(t)
        return [int(x) for x in t.tolist()]

    def max_floats(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.dev)
        if self.use_dist:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    def barrier(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def measure(ctx, spec, steps, warmup, rr=0, count_reference=True):
    """One workload on this rank's share: untimed counting passes, then `warmup` + `steps` timed frames
    (render + resolve + quantise + gather + un-tile).  Returns a dict (timings are max over ranks, counts summed)."""
    import numpy as np
    from raytracer_amd import _ffi as F
    from raytracer_amd import api
    from raytracer_amd import dist as rdist
    torch = ctx.torch
    rank, world, dev = ctx.rank, ctx.world, ctx.dev
    W, H = spec.width, spec.height
    t0 = time.time()
    scene = api.GpuScene(spec, n_threads=8, device=ctx.local_rank)     # BVH build (host mirror) + upload to HBM
    t_setup = time.time() - t0
    n_max = rdist.max_tiles(W, H, world)
    packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device=dev)
    packed8 = torch.zeros((n_max * 64, 3), dtype=torch.uint8, device=dev)      # the rank's tiles as the 8-bit frame: what is gathered
    rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
    bufs = rdist.make_gather_buffers(packed8, world) if (rank == 0 and ctx.use_dist) else None
    stream = torch.cuda.current_stream().cuda_stream
    over = {"rr_start_bounce": rr} if rr else {}

    # ---- untimed counting passes: rays / nodes / primitive tests of this rank's share (deterministic per seed).
    # (1) box_precision=0, the mode that is timed: what the kernel's own traversal (its SAH tree) fetches.  These are the
    #     algorithmic bytes of the algorithm that runs, the numerator of roofline.achieved.
    # (2) box_precision=1 walks the reference's tree with its FP64 slab test, i.e. exactly the reference's node visits
    #     (tests/test_gpu_parity.py::test_ray_counters_match_oracle): SURVEY 8(d)'s per-ray figure, reported beside it.
    ref_work = None
    if count_reference:
        st_ref = scene.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream, collect_stats=1, box_precision=1, **over)
        torch.cuda.synchronize()
        ref_work = (st_ref.nodes_visited, st_ref.tri_tests + st_ref.sphere_tests + st_ref.plane_tests, st_ref.shaded_hits,
                    st_ref.rays_closest + st_ref.rays_shadow)
    st = scene.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream, collect_stats=1, box_precision=0, **over)
    torch.cuda.synchronize()
    # The reference algorithm makes one shadow query per light and hit; the timed mode does not trace those whose light term
    # is zero either way (surface facing away from the light, rays_shadow_culled).  `value` counts only the queries that
    # were TRAVERSED; the reference's count (what the CPU baseline is measured on) is reported beside it.
    mine = [st.rays_closest, st.rays_shadow, st.nodes_visited, st.tri_tests, st.sphere_tests, st.plane_tests,
            st.shaded_hits, st.paths, st.rays_shadow_culled]
    tot = ctx.sum_ints(mine)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        # showPixel's quantisation (RenderWorker.hs:108-109) is per value, so it is applied to the rank's packed tiles in the
        # resolve pass itself and the 8-bit frame is what crosses xGMI: 3 bytes per pixel instead of 24
        scene.render_tiles_device(rank, world, d_rgb8=packed8.data_ptr(), stream=stream, **over)
        if ev is not None:
            ev[1].record()
        rdist.gather_frame(packed8, W, H, rank, world, image=rgb8, gather_buffers=bufs, force_collective=ctx.use_dist)

    for _ in range(warmup):
        step()
    scene.kernel_times()                                             # discard the launches so far
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    ctx.barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        step(events[k])
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    render_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))    # k_render_tiles + k_resolve, events on the launch stream
    k_sum, k_n = scene.kernel_times()                                     # k_render_tiles alone: HIP events around each launch
    kernel_ms = k_sum / max(1, k_n)
    elapsed, kernel_ms_max = ctx.max_floats([elapsed, kernel_ms])
    name = C.create_string_buffer(128)
    F.check(F.core().rt_kernel_name(scene._h, C.byref(api.make_params(spec, **over)), name, 128))
    res = {"spec": spec, "scene": scene, "bvh": (scene.bvh.n_nodes, scene.bvh.n_leaves, scene.bvh.n_prims), "t_setup": t_setup, "tot": tot, "mine": mine, "ref_work": ref_work,
           "ms_per_step": elapsed / steps * 1e3, "kernel_ms": kernel_ms, "kernel_ms_max": kernel_ms_max, "launches": k_n,
           "render_ms": render_ms, "rgb8": rgb8, "kernel": name.value.decode()}
    return res


def roofline_of(m, workload, world):
    """roofline + config extras of one measured workload (rank 0's kernel)."""
    from raytracer_amd import _ffi as F
    spec, mine, ref_work, kernel_ms = m["spec"], m["mine"], m["ref_work"], m["kernel_ms"]
    n_nodes, n_leaves, n_prims = m["bvh"]
    W, H = spec.width, spec.height
    my_rays = max(1, mine[0] + mine[1])
    my_alg_bytes = NODE_BYTES * mine[2] + PRIM_BYTES * (mine[3] + mine[4] + mine[5]) + MAT_BYTES * mine[6]
    alg_gbs = my_alg_bytes / (kernel_ms * 1e-3) / 1e9         # rank 0's kernel: algorithmic bytes per second
    sky_lookups = mine[7] if spec.sky is not None else 0      # every path ends in one background lookup (Trace.hs:22,25)
    flops32 = FLOPS_NODE * mine[2]
    flops64 = FLOPS_TRI * mine[3] + FLOPS_SPHERE * mine[4] + FLOPS_PLANE * mine[5] + FLOPS_SHADED * mine[6] + FLOPS_SKY * sky_lookups
    alg_tflops = (flops32 + flops64) / (kernel_ms * 1e-3) / 1e12
    # measured HBM bytes per launch (rocprofv3 --pmc) from the COMMITTED profile of this workload, never from this run; the
    # record carries the digest of the kernel sources it was taken on (rt_build_id): another digest => stale, and it is not used
    build_id = F.core().rt_build_id().decode()
    traffic, pmc = None, {}
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tf):
        try:
            rec = json.load(open(tf)).get(f"{workload}_{W}x{H}_spp{spec.spp}_n{world}")
        except Exception:
            rec = None
        if rec:
            pmc = {k: rec[k] for k in ("valu_busy_frac", "valu_lane_utilisation", "l2_hit_rate", "source", "build_id", "kernel") if k in rec}
            pmc["stale"] = rec.get("build_id") != build_id or rec.get("kernel", m["kernel"]) != m["kernel"]
            if not pmc["stale"]:
                traffic = rec["hbm_bytes_per_launch"]
            else:
                pmc["stale_hbm_bytes_per_launch"] = rec["hbm_bytes_per_launch"]
    scene_bytes = NODE_BYTES * (n_nodes - n_leaves) + PRIM_BYTES * n_prims
    if spec.sky is not None:
        scene_bytes += spec.sky[1].nbytes
    common = {"traffic": traffic, "traffic_source": "committed profile (profiles/pmc_traffic.json), not this run",
              "kernel": m["kernel"], "build_id": build_id, "kernel_ms": round(kernel_ms, 3), "launches": m["launches"],
              "render_plus_resolve_ms": round(m["render_ms"], 3),
              "nodes_per_ray": round(mine[2] / my_rays, 2), "prim_tests_per_ray": round((mine[3] + mine[4] + mine[5]) / my_rays, 2),
              "pmc": pmc, "scene_bytes": scene_bytes}
    if ref_work is not None:
        common["reference_algorithm"] = {
            "note": "the same rays through the reference's median-split tree (box_precision=1): SURVEY 8(d)'s figure",
            "nodes_per_ray": round(ref_work[0] / max(1, ref_work[3]), 2),
            "prim_tests_per_ray": round(ref_work[1] / max(1, ref_work[3]), 2),
            "bytes_per_ray": round((NODE_BYTES * ref_work[0] + PRIM_BYTES * ref_work[1] + MAT_BYTES * ref_work[2]) / max(1, ref_work[3]), 1)}
    if scene_bytes <= CACHE_RESIDENT_BYTES:
        # cache-resident scene: the memory side is not the bound (measured HBM-side traffic is a few % of the algorithmic
        # bytes), the vector ALUs are (SURVEY 8d names VALU as the co-limit).  achieved = the ALGORITHM's flops per launch
        # (8d's weights x the device counters of the traversal that ran) / kernel time, against the FP32 vector peak.
        roofline = {"bound": "valu", "achieved": round(alg_tflops, 3), "peak": VALU_PEAK_TFLOPS_F32, "unit": "TFLOP/s",
                    "frac": round(alg_tflops / VALU_PEAK_TFLOPS_F32, 5),
                    "algorithmic_flops_per_launch": flops32 + flops64, "fp64_share_of_flops": round(flops64 / max(1, flops32 + flops64), 3),
                    "frac_with_fp64_part_rated_at_78.6": round((flops32 / VALU_PEAK_TFLOPS_F32 + flops64 / VALU_PEAK_TFLOPS_F64) / (kernel_ms * 1e-3) / 1e12, 5),
                    "flops_per_ray": round((flops32 + flops64) / my_rays, 1),
                    "on_chip_bytes": {"algorithmic_bytes_per_launch": my_alg_bytes, "bytes_per_ray": round(my_alg_bytes / my_rays, 1),
                                      "rate_GBs": round(alg_gbs, 1), "served_by": "LDS (top BVH levels), L1, L2; not an HBM figure"},
                    "hbm_frac_measured": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "frac = useful algorithmic flops / vector peak: low because most issued VALU work is not arithmetic of the "
                            "algorithm (selects, votes, address and stack handling, Philox) and lanes idle between states; the SIMDs "
                            "themselves are issue-saturated (pmc.valu_busy_frac). See DESIGN.md section 7"}
    else:
        # the scene exceeds the caches: HBM is the memory-side bound.  achieved = MEASURED HBM bytes per launch (PMC) / kernel
        # time when a counter record of THIS kernel is committed for this workload, else the algorithmic bytes (an upper bound
        # of what must move, served in part by the caches).
        hb = traffic if traffic is not None else my_alg_bytes
        gbs = hb / (kernel_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                    "achieved_is": "measured HBM traffic (FETCH_SIZE x2 + WRITE_SIZE)" if traffic is not None else "algorithmic bytes (no counter record of this kernel)",
                    "algorithmic_bytes_per_launch": my_alg_bytes, "bytes_per_ray": round(my_alg_bytes / my_rays, 1),
                    "valu": {"achieved_TFLOPs": round(alg_tflops, 3), "frac_of_157.3": round(alg_tflops / VALU_PEAK_TFLOPS_F32, 5)}}
    roofline.update(common)
    return roofline


def workload_text(spec):
    lights = f"{len(spec.lights)} light" + ("" if len(spec.lights) == 1 else "s")
    sky_txt = (f"synthetic HDR sky {spec.sky[1].shape[1]}x{spec.sky[1].shape[0]} RGBF (the author's .hdr is not in the reference)"
               if spec.sky is not None else f"no sky texture, background {spec.background['tag']}")
    return (f"{spec.name}: {spec.n_triangles()} triangles, {spec.width}x{spec.height}, {spec.spp} spp, maxBounces {spec.max_bounces}, "
            f"leafThreshold {spec.leaf_threshold}, bvhMaxDepth {spec.bvh_max_depth}, {lights}, {sky_txt}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--workload", default="c4", choices=["c4", "c4v", "c3", "c2", "c5"],
                    help="c4: the metric config as shipped in obli.json (bvhMaxDepth 10, HDR sky); c4v: round 1's variant (bvhMaxDepth 32, solid background)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--others", default="c2,c3,c5", help="N = 1 only: further workloads measured with the same binary into `other_workloads` ('' = none)")
    ap.add_argument("--no-rr", action="store_true", help="skip the Russian-roulette block")
    ap.add_argument("--no-reference-count", action="store_true", help="skip the untimed box_precision=1 counting pass (roofline.reference_algorithm): profiling runs")
    ap.add_argument("--single-process", action="store_true", help="rt_init_devices + ONE rt_render per frame on --gpus devices of this process")
    ap.add_argument("--devices", default=None, help="--single-process: comma-separated device ids (may repeat: virtual devices)")
    args = ap.parse_args()
    if args.single_process:
        return single_process(args)

    ctx = Ctx(args.gpus)
    rank, world = ctx.rank, ctx.world
    spec = make_spec(args.workload, args.width, args.height, args.spp)
    m = measure(ctx, spec, args.steps, args.warmup, count_reference=not args.no_reference_count)
    W, H = spec.width, spec.height
    tot = m["tot"]
    rays_total = tot[0] + tot[1]
    mrays = rays_total / (m["ms_per_step"] * 1e-3) / 1e6

    rr = None
    if not args.no_rr:
        # BASELINE config 4 says "BVH + Russian roulette"; the reference removed it (Config.hs:57), so it is an opt-in extension
        # (DESIGN.md section 9), reported BESIDE the roulette-free headline: the same frame with rt_params.rr_start_bounce = 2
        m["scene"].close()
        m["scene"] = None
        mr = measure(ctx, spec, args.steps, args.warmup, rr=RR_START, count_reference=False)
        rr_rays = mr["tot"][0] + mr["tot"][1]
        rr = {"rr_start_bounce": RR_START, "ms_per_step": round(mr["ms_per_step"], 3), "kernel_ms": round(mr["kernel_ms_max"], 3),
              "rays_per_frame": rr_rays, "Mrays_per_s": round(rr_rays / (mr["ms_per_step"] * 1e-3) / 1e6, 3), "kernel": mr["kernel"],
              "note": "opt-in extension, not the reference's estimator; unbiased for it where its clamp(., 0, 10) is inactive"}
        mr["scene"].close()

    out = None
    if rank == 0:
        # end-to-end extras (untimed for `value`): frame to host + the reference's P3 text body (ImageGenerator.hs:97-115)
        from raytracer_amd import _ffi as F
        t1 = time.perf_counter()
        host_rgb = m["rgb8"].cpu().numpy()
        cap = 12 * W * H + 64
        buf = C.create_string_buffer(cap)
        n = F.host().rth_format_ppm(W, H, host_rgb.ctypes.data_as(F.c_uint8_p), buf, cap)
        t_ppm = time.perf_counter() - t1
        if n <= 0:
            raise SystemExit("rth_format_ppm failed")
        out = {
            "metric": "Mrays/s + frame time, 63k-tri monkey 1080p AA-100, at 1/2/4/8 MI355X",
            "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(m["ms_per_step"], 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_text(spec),
                       "parallelism": f"tiles{world}: one process per GPU, interleaved 8x8 tiles, RCCL gather of the 8-bit tiles",
                       "rays_per_frame": rays_total, "rays_shadow_culled": tot[8], "reference_rays_per_frame": rays_total + tot[8],
                       "paths_per_frame": tot[7],
                       "frame_time_s": round(m["ms_per_step"] * 1e-3, 5), "kernel_ms": round(m["kernel_ms_max"], 3),
                       "setup_s": round(m["t_setup"], 3), "d2h_plus_ppm_text_s": round(t_ppm, 3),
                       "end_to_end_s": round(m["t_setup"] + m["ms_per_step"] * 1e-3 + t_ppm, 3)},
            "roofline": roofline_of(m, args.workload, world),
        }
        if rr is not None:
            out["rr"] = rr
    m["rgb8"] = None
    if m["scene"] is not None:
        m["scene"].close()
    if world == 1 and args.others:
        # the other BASELINE configs on the same binary (DESIGN.md section 7's table), 3 frames each
        others = {}
        for wl in [w for w in args.others.split(",") if w and w != args.workload]:
            sp = make_spec(wl)
            free = ctx.torch.cuda.mem_get_info()[0]
            if sp.width * sp.height * sp.spp * 24 > 0.6 * free:
                others[wl] = {"skipped": "sample buffer does not fit the free HBM"}
                continue
            mo = measure(ctx, sp, 3, 1, count_reference=False)
            r_ = mo["tot"][0] + mo["tot"][1]
            rl = roofline_of(mo, wl, world)
            others[wl] = {"workload": workload_text(sp), "ms_per_step": round(mo["ms_per_step"], 3), "kernel_ms": round(mo["kernel_ms"], 3),
                          "rays_per_frame": r_, "Mrays_per_s": round(r_ / (mo["ms_per_step"] * 1e-3) / 1e6, 3), "setup_s": round(mo["t_setup"], 3),
                          "kernel": mo["kernel"], "roofline": {k: rl[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "pmc", "nodes_per_ray", "prim_tests_per_ray")}}
            mo["scene"].close()
            mo["scene"] = mo["rgb8"] = None
            ctx.torch.cuda.empty_cache()
        out["other_workloads"] = others
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if ctx.use_dist:
        ctx.dist.destroy_process_group()


def single_process(args):
    """The same metric through the path the Haskell host binds: rt_init_devices(n, ids) once, then ONE blocking rt_render per
    frame that fans out over the group inside the library (one host thread + stream per device, peer copies of the 8-bit tiles
    to device 0 over xGMI instead of RCCL: one process owns all devices) and hands back the 8-bit frame in
    host memory - so a step here includes the 6-MB device-to-host copy, which the per-rank path keeps out of `value`."""
    from raytracer_amd import api
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but {len(devices)} device ids")
    spec = make_spec(args.workload, args.width, args.height, args.spp)
    t0 = time.time()
    scene = api.GpuScene(spec, n_threads=8, devices=devices)
    t_setup = time.time() - t0
    _, st = scene.render(want=("rgb8",), collect_stats=1)
    rays = st.rays_closest + st.rays_shadow
    for _ in range(args.warmup):
        scene.render(want=("rgb8",))
    ms_k = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, s1 = scene.render(want=("rgb8",))
        ms_k.append(s1.ms_render)
    elapsed = time.perf_counter() - t0
    ms = elapsed / args.steps * 1e3
    out = {"metric": "Mrays/s + frame time, 63k-tri monkey 1080p AA-100, at 1/2/4/8 MI355X",
           "value": round(rays / (ms * 1e-3) / 1e6, 3), "unit": "Mrays/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": workload_text(spec),
                      "parallelism": f"group{args.gpus}: ONE process, rt_init_devices({devices}) + one blocking rt_render per frame, "
                                     "peer copies (hipMemcpyPeerAsync) of the 8-bit tiles to device 0 instead of RCCL, frame returned in host memory",
                      "rays_per_frame": rays, "slowest_device_kernel_plus_resolve_ms": round(sum(ms_k) / len(ms_k), 3), "setup_s": round(t_setup, 3),
                      "includes": "D2H copy of the 8-bit frame (rt_render hands back host buffers)"}}
    scene.close()
    print(json.dumps(out), flush=True)


def cpu_baseline(spec, threads, target_s=15.0):
    """The oracle (a C++ port of the reference's algorithm, oracle/oracle.cpp) timed on the host cores on a bounded
    sample of the SAME workload: every k-th row of the frame at full spp, k chosen from a short probe so that the
    sample is about `target_s` seconds of CPU work.  Only the orc_render call is timed (no numpy allocation)."""
    from oracle import oracle as O
    threads = max(1, min(threads, os.cpu_count() or 1))
    o = O.OracleScene(spec)

    def run(stride):
        _, cnt = o.render(rows=(stride // 2, spec.height), row_stride=stride, n_threads=threads, counters=True, want_ids=False)
        return cnt["rays_closest"] + cnt["rays_shadow"], o.last_render_s, len(range(stride // 2, spec.height, stride))

    probe = max(1, spec.height // 16)                       # ~16 rows spread over the frame
    rays, dt, rows = run(probe)
    full_s = dt * spec.height / rows
    stride = int(min(probe, max(1, round(full_s / target_s))))
    if stride < probe:
        rays, dt, rows = run(stride)
    o.close()
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{rows} of {spec.height} rows (every {stride}th) x {spec.width} px x {spec.spp} spp = {rays} rays in {dt:.2f} s; "
                      f"full frame extrapolates to {dt * spec.height / rows:.1f} s"}


if __name__ == "__main__":
    main()
