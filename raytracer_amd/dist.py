"""One process per GPU: interleaved-tile sharding of the image and the framebuffer gather.

The path shards naturally: pixels are independent (app/Rendering/RenderWorker.hs:67-69),
the reference already shards rows over its 24 worker threads (app/Rendering/ImageGenerator.hs:71-94).
Here the image is cut into 8x8-pixel tiles, tile (tx, ty) belongs to rank (tx + ty) mod world, every
rank renders its tiles from its own HBM-resident copy of the scene, and ONE exchange step follows:
the packed per-rank tile buffers are gathered on rank 0 (torch.distributed `gather`, i.e. RCCL
send/recv over xGMI with backend "nccl"; gloo on CPU for the tests) and scattered into the
row-major frame.  The RNG is keyed on the global pixel index, so the frame is bit-identical for
every world size.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _ffi as F


def tiles_x(width):
    return (width + 7) // 8


def tiles_y(height):
    return (height + 7) // 8


def tile_skew() -> int:
    """The skew of the tile -> rank map IN FORCE in librtcore (1 unless the experiment knob RT_TILE_SKEW said otherwise when
    the library first read it): asked of the library, so that this module's tile ownership and the device tile lists can
    never disagree, whatever happens to the environment variable after the first call."""
    return int(F.core().rt_tile_skew())


def tile_owner(tx, ty, world):
    return (tx + tile_skew() * ty) % world


def tile_list_host(width, height, rank, world) -> np.ndarray:
    """Pure-host statement of rt_tile_list (tested equal to it): this rank's tile ids, increasing."""
    ty, tx = np.meshgrid(np.arange(tiles_y(height)), np.arange(tiles_x(width)), indexing="ij")
    own = ((tx + tile_skew() * ty) % world) == rank
    return (ty * tiles_x(width) + tx)[own].astype(np.int32)


def max_tiles(width, height, world) -> int:
    return max(len(tile_list_host(width, height, r, world)) for r in range(world))


def untile_into(image: torch.Tensor, packed: torch.Tensor, width, height, rank, world):
    """Scatters one rank's packed buffer (ntiles*64, C) into image (H, W, C).  Device tensors go through
    rt_untile_device (HIP kernel on the current stream); CPU tensors through an index scatter."""
    n_comp = packed.shape[-1]
    if packed.is_cuda:
        stream = torch.cuda.current_stream().cuda_stream
        if packed.dtype == torch.float64:
            F.check(F.core().rt_untile_device(width, height, rank, world, C.c_void_p(packed.data_ptr()),
                                              C.c_void_p(image.data_ptr()), n_comp, C.c_void_p(stream)))
        else:                                  # any other element type goes as bytes (the 8-bit frame: 3 bytes per pixel)
            F.check(F.core().rt_untile_bytes_device(width, height, rank, world, C.c_void_p(packed.data_ptr()),
                                                    C.c_void_p(image.data_ptr()), n_comp * packed.element_size(), C.c_void_p(stream)))
        return image
    tiles = tile_list_host(width, height, rank, world)
    if len(tiles) == 0:
        return image
    lane = np.arange(64)
    px = (tiles[:, None] % tiles_x(width)) * 8 + (lane & 7)[None, :]
    py = (tiles[:, None] // tiles_x(width)) * 8 + (lane >> 3)[None, :]
    ok = (px < width) & (py < height)
    src = packed.reshape(-1, 64, n_comp)[: len(tiles)]
    image[torch.from_numpy(py[ok]), torch.from_numpy(px[ok])] = src[torch.from_numpy(ok)]
    return image


class GatherBuffers(list):
    """Rank 0's landing buffers of the gather: a list of per-rank tensors (what torch.distributed.gather takes) that are views
    of ONE contiguous (world, slots, C) tensor `big`, so that the un-tiling of all ranks is a single launch."""

    def __init__(self, big: torch.Tensor):
        super().__init__(big[r] for r in range(big.shape[0]))
        self.big = big


def make_gather_buffers(packed: torch.Tensor, world) -> GatherBuffers:
    return GatherBuffers(torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device))


def _contiguous_parent(bufs, packed):
    big = getattr(bufs, "big", None)
    if big is None or not big.is_cuda or big.shape[1:] != packed.shape or big.dtype != packed.dtype:
        return None
    return big


def gather_frame(packed: torch.Tensor, width, height, rank, world, image: torch.Tensor | None = None,
                 gather_buffers: list | None = None, force_collective: bool = False):
    """The exchange step.  packed: (max_tiles*64, C) on every rank (padded to the largest tile count so
    the gather is regular).  Returns the (H, W, C) frame on rank 0, None elsewhere."""
    n_comp = packed.shape[-1]
    if world == 1 and not force_collective:      # (force_collective: the one-rank rehearsal of the collective path)
        if image is None:
            image = torch.empty((height, width, n_comp), dtype=packed.dtype, device=packed.device)
        return untile_into(image, packed, width, height, 0, 1)
    if rank == 0:
        bufs = gather_buffers if gather_buffers is not None else [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, gather_list=bufs, dst=0)
        if image is None:
            image = torch.empty((height, width, n_comp), dtype=packed.dtype, device=packed.device)
        big = _contiguous_parent(bufs, packed)
        if big is not None:                      # device buffers from make_gather_buffers: all ranks in one launch
            stream = torch.cuda.current_stream().cuda_stream
            slot_bytes = n_comp * packed.element_size()
            F.check(F.core().rt_untile_all_bytes_device(width, height, world, C.c_void_p(big.data_ptr()), packed.shape[0] * slot_bytes,
                                                        C.c_void_p(image.data_ptr()), slot_bytes, C.c_void_p(stream)))
            return image
        for r in range(world):
            untile_into(image, bufs[r], width, height, r, world)
        return image
    dist.gather(packed, gather_list=None, dst=0)
    return None
