"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" is RCCL on
ROCm; "gloo" for CPU tests).

The reference is single-GPU (one VkDevice, one queue).  The draw path shards in two ways,
neither of which needs a collective while rendering:
  * views / frames: every rank renders whole frames of its own (no exchange at all);
  * sort-first tiles: rank k renders the 32x32 tiles t with t % world == k of ONE frame
    (rnd_cmd_set_tile_shard); geometry is replicated, and only the finished colour image
    is exchanged -- one all_gather of equal-sized slabs, each holding the tiles one rank owns,
    which are then put back in place.
"""
import numpy as np

TILE = 32


def tile_owner_map(width, height, world):
    """(H, W) int array: rank that owns each pixel under round-robin tile dealing."""
    tiles_x = (width + TILE - 1) // TILE
    ty, tx = np.mgrid[0:height, 0:width] // TILE
    return ((ty * tiles_x + tx) % world).astype(np.int64)


def compose_tiles(parts, width, height):
    """Final image from the per-rank images (each H x W x C, tiles it does not own are
    arbitrary).  Works on numpy arrays and on torch tensors."""
    world = len(parts)
    owner = tile_owner_map(width, height, world)
    if isinstance(parts[0], np.ndarray):
        stack = np.stack(parts)
        return np.take_along_axis(stack, owner[None, :, :, None], axis=0)[0]
    import torch
    stack = torch.stack(list(parts))
    own = torch.from_numpy(owner).to(stack.device)
    idx = own[None, :, :, None].expand(1, height, width, stack.shape[-1])
    return torch.gather(stack, 0, idx)[0]


def pack_owned_tiles(image, rank, world):
    """The tiles rank `rank` owns, cut out of its H x W x C torch image and laid end to end:
    (slots, TILE, TILE, C) with slots = ceil(tiles / world), the same on every rank (a rank with
    one tile fewer pads with zeros).  This slab is all a rank contributes to the exchange: 1/world
    of the image instead of the whole of it."""
    import torch
    h, w, c = image.shape
    tiles_x, tiles_y = (w + TILE - 1) // TILE, (h + TILE - 1) // TILE
    if h != tiles_y * TILE or w != tiles_x * TILE:
        padded = image.new_zeros((tiles_y * TILE, tiles_x * TILE, c))
        padded[:h, :w] = image
        image = padded
    tiles = image.view(tiles_y, TILE, tiles_x, TILE, c).permute(0, 2, 1, 3, 4).reshape(tiles_y * tiles_x, TILE, TILE, c)
    slots = (tiles_y * tiles_x + world - 1) // world
    mine = tiles[rank::world]
    if mine.shape[0] < slots:
        mine = torch.cat([mine, mine.new_zeros((slots - mine.shape[0], TILE, TILE, c))])
    return mine.contiguous()


def unpack_tiles(slabs, width, height):
    """Inverse of pack_owned_tiles over all ranks: slabs is (world, slots, TILE, TILE, C), tile t sits
    at [t % world, t // world]; returns the H x W x C image."""
    world, slots, _, _, c = slabs.shape
    tiles_x, tiles_y = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    tiles = slabs.transpose(0, 1).reshape(world * slots, TILE, TILE, c)[:tiles_y * tiles_x]
    image = tiles.view(tiles_y, tiles_x, TILE, TILE, c).permute(0, 2, 1, 3, 4).reshape(tiles_y * TILE, tiles_x * TILE, c)
    return image[:height, :width].contiguous()


def gather_frame(local_image, dist, world, rank=None):
    """One all_gather of equal-sized slabs, each the tiles one rank owns (its share of the H x W x C
    uint8 torch image on the backend's device), then the tiles are put back in place.  Every rank
    returns the full frame.  Bytes on the wire per rank: image / world sent, image received."""
    import torch
    if rank is None:
        rank = dist.get_rank()
    h, w = local_image.shape[0], local_image.shape[1]
    slab = pack_owned_tiles(local_image, rank, world)
    slabs = torch.empty((world * slab.shape[0],) + tuple(slab.shape[1:]), dtype=slab.dtype, device=slab.device)
    dist.all_gather_into_tensor(slabs, slab)  # rank r's slab lands at [r * slots, (r + 1) * slots)
    return unpack_tiles(slabs.view((world,) + tuple(slab.shape)), w, h)


def max_over_ranks(value, dist, device="cpu"):
    """Largest value over the ranks (the benchmark's elapsed time)."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
