"""GPU + multi-process: two ranks (two processes sharing the one GPU of the test box) each render their interleaved
tiles with rt_render_tiles_device, the packed buffers are gathered on rank 0 (gloo here, RCCL in bench.py — two ranks
cannot share one device under RCCL) and un-tiled; the frame must be bit-identical to the single-process render."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from raytracer_amd import api, scenes
        from raytracer_amd import dist as rdist
        spec = scenes.c3_blob988(width=211, height=123, spp=6)
        W, H = spec.width, spec.height
        torch.cuda.set_device(0)
        g = api.GpuScene(spec, device=0)
        n_max = rdist.max_tiles(W, H, world)
        packed = torch.zeros((n_max * 64, 3), dtype=torch.float64, device="cuda")
        g.render_tiles_device(rank, world, d_color=packed.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        img = rdist.gather_frame(packed.cpu(), W, H, rank, world)
        if rank == 0:
            full, _ = g.render(want=("color",))
            np.save(out_path, np.stack([img.numpy(), full["color"]]))
        g.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_processes_one_frame(need_gpu, tmp_path):
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    out = str(tmp_path / "frames.npy")
    mp.spawn(worker, args=(2, port, out), nprocs=2, join=True)
    both = np.load(out)
    assert np.array_equal(both[0], both[1])


@pytest.mark.timeout(300)
def test_bench_rccl_code_path_with_one_rank(need_gpu):
    """bench.py's RCCL branch (backend "nccl": process group bound to the device, all-reduce of the counters, barrier, gather of
    the packed 8-bit tiles as device tensors, un-tiling) cannot run with two ranks on one GPU, and no 8-GPU node was available to
    the driver in round 1 - so it is rehearsed with ONE rank under torch.distributed.run, exactly as the driver launches it."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--width", "320", "--height", "180", "--spp", "4", "--no-cpu-baseline", "--others", ""]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["roofline"]["kernel_ms"] > 0
    assert j["rr"]["rr_start_bounce"] == 2 and j["rr"]["rays_per_frame"] < j["config"]["rays_per_frame"]
    assert j["roofline"]["kernel"].startswith("k_render_tiles<float, false, true, false>") and len(j["roofline"]["build_id"]) == 16
