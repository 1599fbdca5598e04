#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric on MI355X: Mrays/s and frame time of the per-pixel ray<->scene
hot path on the 63,432-triangle scene at 1920x1080, 100 spp (config C4), N = 1/2/4/8 GPUs.

One step = one frame: every rank renders its interleaved 8x8 tiles from its HBM-resident scene copy,
then quantises them to the 8-bit frame; the packed 8-bit tiles are gathered on rank 0 over RCCL and un-tiled (raytracer_amd/dist.py).
A ray = one BVH query (primary/bounce closest-hit + shadow any-hit), counted on the device by the
stats variant of the kernel in an untimed pass (counts are deterministic for a seed).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=100)
    ap.add_argument("--workload", default="c4", choices=["c4", "c4v", "c3", "c2", "c5"],
                    help="c4: the metric config as shipped in obli.json (bvhMaxDepth 10, HDR sky); c4v: round 1's variant (bvhMaxDepth 32, solid background)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from raytracer_amd import api, scenes
    from raytracer_amd import dist as rdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_DIST=1: take the RCCL code path (process group, all-reduce, gather of device tensors, barrier) even with
    # one rank - the one-GPU test box's rehearsal of what the 8-GPU run executes (tests/test_gpu_dist.py)
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    mk = {"c4": scenes.c4_shipped, "c4v": scenes.c4_blob63k, "c3": scenes.c3_blob988, "c2": scenes.c2_spheres, "c5": scenes.c5_multi}[args.workload]
    spec = mk(width=args.width, height=args.height, spp=args.spp)
    W, H = spec.width, spec.height
    t0 = time.time()
    scene = api.GpuScene(spec, n_threads=8, device=local_rank)     # BVH build (host mirror) + upload to HBM
    t_setup = time.time() - t0

    n_local = len(rdist.tile_list_host(W, H, rank, world))
    n_max = rdist.max_tiles(W, H, world)
    packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device=dev)
    packed8 = torch.zeros((n_max * 64, 3), dtype=torch.uint8, device=dev)      # the rank's tiles as the 8-bit frame: what is gathered
    rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
    bufs = [torch.empty_like(packed8) for _ in range(world)] if (rank == 0 and use_dist) else None
    stream = torch.cuda.current_stream().cuda_stream

    # ---- untimed counting passes: rays / nodes / primitive tests of this rank's share (deterministic per seed).
    # (1) box_precision=0, the mode that is timed: what the kernel's own traversal (its SAH tree) fetches.  These are the
    #     algorithmic bytes of the algorithm that runs, the numerator of roofline.achieved.
    # (2) box_precision=1 walks the reference's tree with its FP64 slab test, i.e. exactly the reference's node visits:
    #     SURVEY 8(d)'s per-ray figure for the reference algorithm, reported beside it.
    st_ref = scene.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream, collect_stats=1, box_precision=1)
    torch.cuda.synchronize()
    ref_work = (st_ref.nodes_visited, st_ref.tri_tests + st_ref.sphere_tests + st_ref.plane_tests, st_ref.shaded_hits,
                st_ref.rays_closest + st_ref.rays_shadow)
    st = scene.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream, collect_stats=1, box_precision=0)
    torch.cuda.synchronize()
    # The reference algorithm makes one shadow query per light and hit; the timed mode does not trace those whose light term
    # is zero either way (surface facing away from the light, rays_shadow_culled).  `value` counts only the queries that
    # were TRAVERSED; the reference's count (what the CPU baseline is measured on) is reported beside it.
    cnt = torch.tensor([st.rays_closest, st.rays_shadow, st.nodes_visited, st.tri_tests, st.sphere_tests, st.plane_tests,
                        st.shaded_hits, st.paths, st.rays_shadow_culled], dtype=torch.int64, device=dev)
    my_cnt = cnt.clone()
    if use_dist:
        dist.all_reduce(cnt)
    tot = [int(x) for x in cnt.tolist()]
    rays_total = tot[0] + tot[1]
    mine = [int(x) for x in my_cnt.tolist()]
    my_alg_bytes = NODE_BYTES * mine[2] + PRIM_BYTES * (mine[3] + mine[4] + mine[5]) + MAT_BYTES * mine[6]

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        scene.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=stream)
        if ev is not None:
            ev[1].record()
        # showPixel's quantisation (RenderWorker.hs:108-109) is per value, so it is applied to the rank's packed tiles and
        # the 8-bit frame is what crosses xGMI: 3 bytes per pixel instead of 24
        F.check(F.core().rt_quantize_device(C.c_void_p(packed.data_ptr()), C.c_void_p(packed8.data_ptr()), packed.numel(), C.c_void_p(stream)))
        rdist.gather_frame(packed8, W, H, rank, world, image=rgb8, gather_buffers=bufs, force_collective=use_dist)

    from raytracer_amd import _ffi as F
    for _ in range(args.warmup):
        step()
    scene.kernel_times()                                             # discard the launches so far
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    render_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))    # k_render_tiles + k_resolve, events on the launch stream
    k_sum, k_n = scene.kernel_times()                                     # k_render_tiles alone: HIP events around each launch
    kernel_ms = k_sum / max(1, k_n)

    tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = float(tt[0]), float(tt[1])
    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_total / (ms_per_step * 1e-3) / 1e6

    t_ppm = None
    if rank == 0:
        # end-to-end extras (untimed for `value`): frame to host + the reference's P3 text body (ImageGenerator.hs:97-115)
        from raytracer_amd import _ffi as F
        t1 = time.perf_counter()
        host_rgb = rgb8.cpu().numpy()
        cap = 12 * W * H + 64
        buf = C.create_string_buffer(cap)
        n = F.host().rth_format_ppm(W, H, host_rgb.ctypes.data_as(F.c_uint8_p), buf, cap)
        t_ppm = time.perf_counter() - t1
        if n <= 0:
            raise SystemExit("rth_format_ppm failed")
    if rank == 0:
        my_rays = max(1, mine[0] + mine[1])
        alg_gbs = my_alg_bytes / (kernel_ms * 1e-3) / 1e9         # rank 0's kernel: algorithmic bytes per second
        sky_lookups = mine[7] if spec.sky is not None else 0      # every path ends in one background lookup (Trace.hs:22,25)
        flops32 = FLOPS_NODE * mine[2]
        flops64 = FLOPS_TRI * mine[3] + FLOPS_SPHERE * mine[4] + FLOPS_PLANE * mine[5] + FLOPS_SHADED * mine[6] + FLOPS_SKY * sky_lookups
        alg_tflops = (flops32 + flops64) / (kernel_ms * 1e-3) / 1e12
        traffic, pmc = None, {}
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # measured HBM bytes per launch (rocprofv3 --pmc), if committed
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf)).get(f"{args.workload}_{W}x{H}_spp{spec.spp}_n{world}")
                traffic = rec["hbm_bytes_per_launch"] if rec else None
                pmc = {k: rec[k] for k in ("valu_busy_frac", "valu_lane_utilisation", "l2_hit_rate", "source") if rec and k in rec}
            except Exception:
                traffic = None
        scene_bytes = NODE_BYTES * (scene.bvh.n_nodes - scene.bvh.n_leaves) + PRIM_BYTES * scene.bvh.n_prims
        if spec.sky is not None:
            scene_bytes += spec.sky[1].nbytes
        common = {"traffic": traffic, "kernel": "k_render_tiles<float,false,true>", "kernel_ms": round(kernel_ms, 3), "launches": k_n,
                  "render_plus_resolve_ms": round(render_ms, 3),
                  "nodes_per_ray": round(mine[2] / my_rays, 2), "prim_tests_per_ray": round((mine[3] + mine[4] + mine[5]) / my_rays, 2),
                  "reference_algorithm": {
                      "note": "the same rays through the reference's median-split tree (box_precision=1): SURVEY 8(d)'s figure",
                      "nodes_per_ray": round(ref_work[0] / max(1, ref_work[3]), 2),
                      "prim_tests_per_ray": round(ref_work[1] / max(1, ref_work[3]), 2),
                      "bytes_per_ray": round((NODE_BYTES * ref_work[0] + PRIM_BYTES * ref_work[1] + MAT_BYTES * ref_work[2]) / max(1, ref_work[3]), 1)},
                  "pmc": pmc, "scene_bytes": scene_bytes}
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
            # time when a counter record is committed for this workload, else the algorithmic bytes (then an upper bound).
            hb = traffic if traffic is not None else my_alg_bytes
            gbs = hb / (kernel_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                        "achieved_is": "measured HBM traffic (FETCH_SIZE x2 + WRITE_SIZE)" if traffic is not None else "algorithmic bytes (no counter record)",
                        "algorithmic_bytes_per_launch": my_alg_bytes, "bytes_per_ray": round(my_alg_bytes / my_rays, 1),
                        "valu": {"achieved_TFLOPs": round(alg_tflops, 3), "frac_of_157.3": round(alg_tflops / VALU_PEAK_TFLOPS_F32, 5)}}
        roofline.update(common)
        lights = f"{len(spec.lights)} point light" + ("" if len(spec.lights) == 1 else "s")
        sky_txt = (f"synthetic HDR sky {spec.sky[1].shape[1]}x{spec.sky[1].shape[0]} RGBF (the author's .hdr is not in the reference)"
                   if spec.sky is not None else f"no sky texture, background {spec.background['tag']}")
        out = {
            "metric": "Mrays/s + frame time, 63k-tri monkey 1080p AA-100, at 1/2/4/8 MI355X",
            "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{spec.name}: {spec.n_triangles()} triangles, {W}x{H}, {spec.spp} spp, maxBounces {spec.max_bounces}, "
                                   f"leafThreshold {spec.leaf_threshold}, bvhMaxDepth {spec.bvh_max_depth}, {lights}, {sky_txt}",
                       "parallelism": f"tiles{world}", "rays_per_frame": rays_total, "rays_shadow_culled": tot[8], "reference_rays_per_frame": rays_total + tot[8],
                       "paths_per_frame": tot[7],
                       "frame_time_s": round(ms_per_step * 1e-3, 5), "kernel_ms": round(kernel_ms_max, 3),
                       "setup_s": round(t_setup, 3), "d2h_plus_ppm_text_s": round(t_ppm, 3),
                       "end_to_end_s": round(t_setup + ms_per_step * 1e-3 + t_ppm, 3)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, args.cpu_threads)
        print(json.dumps(out), flush=True)
    scene.close()
    if use_dist:
        dist.destroy_process_group()


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
