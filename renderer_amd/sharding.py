"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" is RCCL on
ROCm; "gloo" for CPU tests).

The reference is single-GPU (one VkDevice, one queue).  The draw path shards in two ways,
neither of which needs a collective while rendering:
  * views / frames: every rank renders whole frames of its own (no exchange at all);
  * sort-first tiles: rank k renders the 32x32 tiles t with t % world == k of ONE frame
    (rnd_cmd_set_tile_shard); geometry is replicated, and only the finished colour image
    is exchanged -- one all_gather of equal-sized slabs, then each pixel is taken from the
    rank that owns its tile.
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


def gather_frame(local_image, dist, world):
    """all_gather of every rank's H x W x C uint8 image (a torch tensor on the backend's
    device) and composition by tile owner.  Every rank returns the full frame."""
    import torch
    parts = [torch.empty_like(local_image) for _ in range(world)]
    dist.all_gather(parts, local_image.contiguous())
    h, w = local_image.shape[0], local_image.shape[1]
    return compose_tiles(parts, w, h)


def max_over_ranks(value, dist, device="cpu"):
    """Largest value over the ranks (the benchmark's elapsed time)."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
