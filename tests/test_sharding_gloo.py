"""world_size-2 (and 3) runs of the multi-GPU helpers over gloo on the CPU: tile-sharded
frames gathered and composed equal the unsharded frame; the elapsed-time reduction takes the
maximum."""
import os
import socket

import numpy as np
import pytest

import helpers
from renderer_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, full_path, out_dir):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    full = np.load(full_path)
    h, w = full.shape[:2]
    # what rank `rank` would hold after rendering only its tiles: its own tiles right, the rest untouched (zero)
    mine = np.where((sharding.tile_owner_map(w, h, world) == rank)[..., None], full, 0).astype(np.uint8)
    frame = sharding.gather_frame(torch.from_numpy(mine), dist, world)
    np.save(os.path.join(out_dir, "frame_%d.npy" % rank), frame.numpy())
    t = sharding.max_over_ranks(1.0 + rank, dist)
    np.save(os.path.join(out_dir, "t_%d.npy" % rank), np.array([t]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_gather_over_gloo(tmp_path, oracle, world):
    import torch.multiprocessing as mp
    sc = helpers.soup_scene(200, 150, 400, 3)
    color, _, _ = oracle.render_scene(sc)
    full_path = str(tmp_path / "full.npy")
    np.save(full_path, color)
    mp.spawn(_worker, args=(world, _free_port(), full_path, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.array_equal(np.load(str(tmp_path / ("frame_%d.npy" % r))), color)
        assert np.load(str(tmp_path / ("t_%d.npy" % r)))[0] == float(world)


@pytest.mark.parametrize("wh", [(200, 150), (64, 96), (33, 31)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_pack_and_unpack_tiles(wh, world):
    """Slabs of owned tiles, stacked as the all_gather stacks them, give the image back: sizes that
    are and are not multiples of the tile, ranks with one tile fewer than others, more ranks than tiles."""
    import torch
    w, h = wh
    rng = np.random.RandomState(5)
    full = torch.from_numpy(rng.randint(0, 256, (h, w, 4)).astype(np.uint8))
    own = torch.from_numpy(sharding.tile_owner_map(w, h, world))
    slabs = []
    for r in range(world):
        mine = torch.where((own == r)[..., None], full, torch.zeros_like(full))  # tiles of others: not rendered
        slabs.append(sharding.pack_owned_tiles(mine, r, world))
    assert len({tuple(s.shape) for s in slabs}) == 1
    tiles = ((w + 31) // 32) * ((h + 31) // 32)
    assert slabs[0].shape[0] == (tiles + world - 1) // world
    assert torch.equal(sharding.unpack_tiles(torch.stack(slabs), w, h), full)


@pytest.mark.parametrize("height", [32, 200, 1080, 4320, 31])
@pytest.mark.parametrize("world", [1, 2, 3, 8, 64])
def test_band_rows_cover_the_frame(height, world):
    """Bands are in rank order, disjoint and cover every row; equal when the rows divide evenly (and leave every rank
    a row of tiles), else whole rows of tiles that differ by at most one row of tiles."""
    spans = [sharding.band_rows(height, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == height
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0 and a0 <= a1
    if height % world == 0 and height // world >= 32:
        assert all(y1 - y0 == height // world for y0, y1 in spans)
    else:
        assert all(y0 % 32 == 0 or y0 == height for y0, _ in spans)
        rows = [(y1 - y0 + 31) // 32 for y0, y1 in spans]
        assert max(rows) - min(rows) <= 1


def _band_worker(rank, world, height, width, port, out_dir):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    full = torch.from_numpy(np.random.RandomState(9).randint(0, 256, (height, width, 4)).astype(np.uint8))
    y0, y1 = sharding.band_rows(height, rank, world)
    frame = sharding.gather_bands(full[y0:y1].clone(), dist, world, height)
    ok = bool(torch.equal(frame, full))
    # the exchange straight out of each rank's own image: right inside its band, anything elsewhere
    image = torch.full_like(full, 77)
    image[y0:y1] = full[y0:y1]
    s0, tallest = sharding.band_send_rows(height, rank, world)
    assert 0 <= s0 <= y0 and y1 <= s0 + tallest <= height
    slabs = torch.empty((world, tallest, width, 4), dtype=torch.uint8)
    dist.all_gather_into_tensor(slabs.view(-1), image[s0:s0 + tallest].reshape(-1))
    ok = ok and bool(torch.equal(sharding.compose_bands(slabs, height), full))
    np.save(os.path.join(out_dir, "b_%d.npy" % rank), np.array([ok]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 128), (2, 100), (3, 200), (3, 270)])
def test_band_gather_over_gloo(tmp_path, world, height):
    """Even bands land in the frame directly, uneven ones through padded slabs: every rank ends with the frame."""
    import torch.multiprocessing as mp
    mp.spawn(_band_worker, args=(world, height, 96, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert np.load(str(tmp_path / ("b_%d.npy" % r)))[0]


def test_band_scene_cuts_area_and_scissor():
    from renderer_amd import scenes
    sc = scenes.Scene(300, 201)
    sc.scissor = (10, 50, 200, 100)
    b = sharding.band_scene(sc, 1, 2)  # rows of tiles: 7 -> 4 + 3: band 1 = pixel rows [128, 201)
    assert b.area == (0, 128, 300, 73) and b.scissor == (10, 128, 200, 22)
    assert sc.area == (0, 0, 300, 201)  # the original is untouched
    even = scenes.Scene(300, 200)  # 200 rows divide evenly: two bands of 100 rows, not aligned to tiles
    assert sharding.band_scene(even, 1, 2).area == (0, 100, 300, 100)


def test_owner_map_matches_kernel_rule():
    """tile t = ty * tiles_x + tx belongs to rank t % world; every pixel has exactly one owner."""
    own = sharding.tile_owner_map(100, 70, 3)
    assert own.shape == (70, 100)
    assert own[0, 0] == 0 and own[0, 32] == 1 and own[0, 64] == 2 and own[0, 96] == 0
    assert own[32, 0] == 4 % 3 and own[69, 99] == (2 * 4 + 3) % 3
    parts = [np.full((70, 100, 4), r, dtype=np.uint8) for r in range(3)]
    assert np.array_equal(sharding.compose_tiles(parts, 100, 70)[..., 0], own)
