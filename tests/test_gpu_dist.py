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
