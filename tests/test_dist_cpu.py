"""The N>1 path on CPU: tile sharding, padded gather (gloo, world_size 2 and 3) and un-tiling reproduce the
full frame bit-for-bit.  No rendering here — each rank fills its packed buffer with a function of the global
pixel coordinate, exactly what rt_render_tiles_device writes per slot on the GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import dist as rdist


def pixel_value(px, py):
    return np.stack([px * 1.0 + 0.25, py * 2.0 - 0.5, (px * 7919 + py * 104729) % 1000 / 3.0], axis=-1)


def fill_packed(width, height, rank, world, n_max):
    tiles = rdist.tile_list_host(width, height, rank, world)
    packed = np.full((n_max * 64, 3), -777.0)            # padding and out-of-image lanes keep a sentinel
    lane = np.arange(64)
    tx = rdist.tiles_x(width)
    for k, t in enumerate(tiles):
        px = (t % tx) * 8 + (lane & 7)
        py = (t // tx) * 8 + (lane >> 3)
        ok = (px < width) & (py < height)
        packed[k * 64 + lane[ok]] = pixel_value(px[ok], py[ok])
    return torch.from_numpy(packed)


def worker(rank, world, port, width, height, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_max = rdist.max_tiles(width, height, world)
        packed = fill_packed(width, height, rank, world, n_max)
        img = rdist.gather_frame(packed, width, height, rank, world)
        if rank == 0:
            np.save(out_path, img.numpy())
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


def worker_u8(rank, world, port, width, height, out_path):
    """bench.py's exchange: the 8-bit tiles are gathered into preallocated buffers and un-tiled into a preallocated frame."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_max = rdist.max_tiles(width, height, world)
        packed8 = (fill_packed(width, height, rank, world, n_max).abs() % 251).to(torch.uint8)
        frame = torch.full((height, width, 3), 9, dtype=torch.uint8) if rank == 0 else None
        # bench.py's landing buffers: views of one contiguous tensor (a list subclass the collective must accept); on CPU tensors
        # the un-tiling falls back to one scatter per rank
        bufs = rdist.make_gather_buffers(packed8, world) if rank == 0 else None
        for _ in range(2):                                    # the buffers are reused frame after frame
            img = rdist.gather_frame(packed8, width, height, rank, world, image=frame, gather_buffers=bufs)
        if rank == 0:
            assert img is frame
            np.save(out_path, frame.numpy())
        else:
            assert img is None
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,width,height", [(2, 64, 40), (2, 101, 67), (3, 50, 30)])
def test_gather_reassembles_frame(tmp_path, world, width, height):
    out = str(tmp_path / "frame.npy")
    mp.spawn(worker, args=(world, free_port(), width, height, out), nprocs=world, join=True)
    img = np.load(out)
    py, px = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    assert np.array_equal(img, pixel_value(px, py))


@pytest.mark.parametrize("world,width,height", [(2, 101, 67), (3, 50, 30)])
def test_gather_of_the_eight_bit_frame(tmp_path, world, width, height):
    out = str(tmp_path / "frame8.npy")
    mp.spawn(worker_u8, args=(world, free_port(), width, height, out), nprocs=world, join=True)
    py, px = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    want = (torch.from_numpy(pixel_value(px, py)).abs() % 251).to(torch.uint8).numpy()
    assert np.array_equal(np.load(out), want)


def test_world1_untile_and_ownership():
    w, h = 37, 21
    packed = fill_packed(w, h, 0, 1, rdist.max_tiles(w, h, 1))
    img = rdist.gather_frame(packed, w, h, 0, 1).numpy()
    py, px = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    assert np.array_equal(img, pixel_value(px, py))
    # every tile has exactly one owner, neighbours differ (interleaving)
    for world in (2, 4, 8):
        owners = np.array([[rdist.tile_owner(x, y, world) for x in range(10)] for y in range(6)])
        assert ((owners >= 0) & (owners < world)).all()
        assert (owners[:, 1:] != owners[:, :-1]).all() and (owners[1:] != owners[:-1]).all()
