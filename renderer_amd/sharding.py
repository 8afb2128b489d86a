"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" is RCCL on
ROCm; "gloo" for CPU tests).

The reference is single-GPU (one VkDevice, one queue).  The draw path shards in two ways,
neither of which needs a collective while rendering:
  * views / frames: every rank renders whole frames of its own (no exchange at all);
  * sort-first tiles: rank k renders the 32x32 tiles t with t % world == k of ONE frame
    (rnd_cmd_set_tile_shard); geometry is replicated, and only the finished colour image
    is exchanged -- one all_gather of equal-sized slabs, each holding the tiles one rank owns,
    which are then put back in place;
  * sort-first bands: rank k renders a band of rows of ONE frame (equal bands when the rows divide
    evenly, whole rows of tiles otherwise), given to the library as the pass's render area and scissor.  The library then launches only the band's
    tiles and skips every draw whose box of vertices misses the band, so geometry work is
    divided as well as pixel work; the exchange is one all_gather of the bands, which are
    contiguous rows of the image.
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


def band_rows(height, rank, world):
    """Pixel rows [y0, y1) of rank `rank`.  When the frame's rows divide evenly among the ranks (and leave each at
    least a row of tiles) every band is height / world rows: equal bands are what one all_gather places in the
    frame directly, with no padding on the wire and no copy afterwards, and that is worth the row of tiles a band
    then shares with each neighbour (both render their part of it).  Otherwise the rows of tiles are split as
    evenly as whole rows of tiles allow (the first `rows % world` ranks get one more); a rank may then get none
    (y0 == y1)."""
    if height % world == 0 and height // world >= TILE:
        rows = height // world
        return rank * rows, (rank + 1) * rows
    tile_rows = (height + TILE - 1) // TILE
    base, extra = divmod(tile_rows, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return min(first * TILE, height), min((first + count) * TILE, height)


def band_scene(scene, rank, world):
    """A copy of the scene whose render area and scissor are cut down to the band of `rank`."""
    import copy
    y0, y1 = band_rows(scene.height, rank, world)
    sc = copy.copy(scene)
    ax, ay, aw, ah = scene.area if scene.area is not None else (0, 0, scene.width, scene.height)
    sx, sy, sw, sh = scene.scissor
    by0, by1 = max(ay, y0), min(ay + ah, y1)
    sc.area = (ax, by0, aw, max(by1 - by0, 0))
    cy0, cy1 = max(sy, y0), min(sy + sh, y1)
    sc.scissor = (sx, cy0, sw, max(cy1 - cy0, 0))
    return sc


def gather_bands(band, dist, world, height):
    """all_gather of the bands: `band` is this rank's rows (rows x W x C uint8 torch tensor on the
    backend's device).  Every rank returns the full H x W x C frame.  When the rows of tiles divide
    evenly the bands land in the frame directly; otherwise each is padded to the tallest band and the
    frame is put together from the valid rows.  Bytes on the wire per rank: its band sent, the frame received."""
    import torch
    spans = [band_rows(height, r, world) for r in range(world)]
    tallest = max(y1 - y0 for y0, y1 in spans)
    w, c = band.shape[1], band.shape[2]
    if all(y1 - y0 == tallest for y0, y1 in spans):
        frame = torch.empty((height, w, c), dtype=band.dtype, device=band.device)
        dist.all_gather_into_tensor(frame.view(-1), band.contiguous().view(-1))
        return frame
    mine = band
    if band.shape[0] < tallest:
        mine = band.new_zeros((tallest, w, c))
        mine[:band.shape[0]] = band
    slabs = torch.empty((world, tallest, w, c), dtype=band.dtype, device=band.device)
    dist.all_gather_into_tensor(slabs.view(-1), mine.contiguous().view(-1))
    return torch.cat([slabs[r, :y1 - y0] for r, (y0, y1) in enumerate(spans)])


def band_send_rows(height, rank, world):
    """Rows [s0, s0 + tallest) of rank `rank`'s own image that it contributes to an exchange of equal-sized
    slabs straight out of the image: `tallest` rows (the tallest band's height) that contain its band.  A rank
    with a shorter band sends neighbouring rows along, which compose_bands drops."""
    tallest = max(b - a for a, b in (band_rows(height, r, world) for r in range(world)))
    y0, _ = band_rows(height, rank, world)
    return min(y0, height - tallest), tallest


def compose_bands(slabs, height):
    """The frame from the gathered slabs (world, tallest, W, C), slab r being rank r's band_send_rows: a view
    when the bands are all equally tall, else a copy of every slab's valid rows."""
    import torch
    world, tallest = slabs.shape[0], slabs.shape[1]
    spans = [band_rows(height, r, world) for r in range(world)]
    if all(b - a == tallest for a, b in spans):
        return slabs.view(height, slabs.shape[2], slabs.shape[3])
    return torch.cat([slabs[r, a - min(a, height - tallest):][:b - a] for r, (a, b) in enumerate(spans)])


class _DeviceMemory:
    """Device memory that is not torch's, described the way torch.as_tensor understands."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_bytes(ptr, nbytes):
    """A uint8 torch tensor over nbytes of device memory at ptr (no copy; the memory stays the owner's)."""
    import torch
    return torch.as_tensor(_DeviceMemory(ptr, nbytes), device="cuda")


def max_over_ranks(value, dist, device="cpu"):
    """Largest value over the ranks (the benchmark's elapsed time)."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
