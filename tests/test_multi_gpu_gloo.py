"""The N>1 path on CPU: interleaved band partition, one all-gather (gloo here,
RCCL on GPUs), composition of the picture.  Each rank 'renders' its bands with
the CPU oracle and lays its block out exactly as rt_export_packed_bands does."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import Scene, load_tables, make_params, multi_gpu
from tests import oracle_lib

W, H = 72, 45          # 6 bands, the top one 5 rows tall: uneven split over 2, 3 or 4 ranks


def block_from_packed(packed_full, rank, world, height):
    """numpy statement of the rt_export_packed_bands layout"""
    rows = multi_gpu.block_rows(world, height)
    out = np.zeros((rows, packed_full.shape[1]), dtype=packed_full.dtype)
    for k, b in enumerate(multi_gpu.bands_of_rank(rank, world, height)):
        for line in range(8):
            j = b * 8 + line
            if j < height:
                out[k * 8 + line] = packed_full[height - 1 - j]
    return out


def _worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = oracle_lib.load()
        tables = load_tables()
        flat = Scene.get_scene().flatten()
        cam = o.scene_camera(0, 0.1)
        p = make_params(W, H, spp=2, band_begin=rank, band_stride=world)
        acc, _, seg = o.render(flat, cam, tables, p, threads=1, want_ids=False)
        acc[..., 3][acc[..., 3] == 0] = 1          # untouched rows: avoid 0/0 in the quantiser
        mine = block_from_packed(o.pack(acc), rank, world, H)
        frame = multi_gpu.gather_frame(torch.from_numpy(mine.astype(np.int64)), H)
        segs = torch.tensor([seg], dtype=torch.int64)
        dist.all_reduce(segs)
        np.save(os.path.join(result_dir, "frame%d.npy" % rank), frame.numpy())
        np.save(os.path.join(result_dir, "seg%d.npy" % rank), segs.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_band_partition_all_gather_rebuilds_the_frame(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    o = oracle_lib.load()
    tables = load_tables()
    flat = Scene.get_scene().flatten()
    cam = o.scene_camera(0, 0.1)
    acc, _, seg = o.render(flat, cam, tables, make_params(W, H, spp=2), threads=2, want_ids=False)
    want = o.pack(acc).astype(np.int64)
    for r in range(world):
        got = np.load(tmp_path / ("frame%d.npy" % r))
        assert np.array_equal(got, want), "rank %d composed a different picture" % r
        assert int(np.load(tmp_path / ("seg%d.npy" % r))[0]) == seg


def test_partition_arithmetic():
    for height in (8, 9, 45, 337, 2160):
        for world in (1, 2, 3, 4, 8):
            bands = [b for r in range(world) for b in multi_gpu.bands_of_rank(r, world, height)]
            assert sorted(bands) == list(range(multi_gpu.total_bands(height)))
            idx = multi_gpu.display_row_index(world, height)
            real = idx[idx < height]
            assert sorted(real.tolist()) == list(range(height))       # every display row exactly once
            assert idx.shape == (world, multi_gpu.block_rows(world, height))


def test_compose_single_rank_is_a_vertical_flip():
    blk = torch.arange(16 * 3, dtype=torch.int32).reshape(16, 3)     # 2 bands, height 13: 3 padding rows
    out = multi_gpu.compose(blk.unsqueeze(0), 13)
    assert torch.equal(out, blk[:13].flip(0))


@pytest.mark.parametrize("world,height", [(2, 337), (4, 1080), (8, 2160), (8, 90), (3, 8), (8, 9)])
def test_compose_rebuilds_any_frame(world, height):
    """blocks cut from a random frame the way rt_export_packed_bands lays them out, stacked in
    rank order like the output of all_gather_into_tensor, compose back to the frame"""
    import torch
    rng = np.random.default_rng(world * 10000 + height)
    frame = rng.integers(0, 1 << 24, size=(height, 37)).astype(np.uint32)
    blocks = np.stack([block_from_packed(frame, r, world, height) for r in range(world)])
    gathered = torch.from_numpy(blocks.astype(np.int64)).reshape(world * blocks.shape[1], 37)
    out = multi_gpu.compose(gathered.view(world, blocks.shape[1], 37), height).numpy().astype(np.uint32)
    assert np.array_equal(out, frame)
