"""Multi-GPU side of the path: how a frame's rays are split over ranks and how the per-rank framebuffers
are combined. One process per GPU, torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" on CPU).

Tracing needs no communication: every rank holds the whole BVH and rays are independent
(rt/bbvh-base/bvh2.cpp:415-501 only reads nodes, index and the scene arrays). The one exchange per frame is the
framebuffer: the reference keeps a running mean with the sample count in .w (libgi/framebuffer.cpp:10-18),
so ranks must combine sums and counts, not means.
"""
import numpy as np


def shard_samples(sppx, world, rank):
    """Sample indices of this rank when a frame's sppx samples per pixel are dealt round-robin."""
    return list(range(rank, sppx, world))


def shard_tiles(width, height, world, rank, tile=64):
    """(x0, y0, x1, y1) of this rank's tiles when the image is cut into tile x tile pixels, dealt round-robin
    in row-major tile order."""
    tiles = [(x, y, min(x + tile, width), min(y + tile, height))
             for y in range(0, height, tile) for x in range(0, width, tile)]
    return tiles[rank::world]


class Framebuffer:
    """framebuffer of libgi/framebuffer.h:49-64: color(x, y) = (mean r, mean g, mean b, sample count)."""

    def __init__(self, width, height, device="cpu"):
        import torch
        self.color = torch.zeros((height, width, 4), dtype=torch.float32, device=device)

    def clear(self):
        self.color.zero_()

    def add(self, radiance, mask=None):
        """framebuffer::add (libgi/framebuffer.cpp:10-18) for one sample per pixel at once: radiance is
        (h, w, 3); pixels where mask is False receive no sample."""
        import torch
        c = self.color
        n = torch.ones_like(c[..., 3]) if mask is None else mask.to(c.dtype)
        new_count = c[..., 3] + n
        safe = torch.where(new_count > 0, new_count, torch.ones_like(new_count))
        acc = radiance * n[..., None]
        c[..., 0:3] = (c[..., 0:3] * c[..., 3:4] + acc) / safe[..., None]
        c[..., 3] = new_count

    def reduce(self, group=None):
        """Combine the framebuffers of all ranks in place (every rank ends with the full frame): one
        all-reduce of (mean * count, count), then back to (mean, count). Works for sample sharding (counts
        add up per pixel) and for tile sharding (a pixel has samples on one rank only) alike."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        buf = self.color.clone()
        buf[..., 0:3] *= buf[..., 3:4]
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        count = buf[..., 3:4]
        self.color[..., 0:3] = torch.where(count > 0, buf[..., 0:3] / torch.clamp(count, min=1.0), torch.zeros_like(buf[..., 0:3]))
        self.color[..., 3:4] = count
        return self

    def to_rgb8(self):
        """framebuffer::png (libgi/framebuffer.cpp:21-30): clamp, gamma 1/2.2, scale to 255, flip vertically."""
        c = self.color[..., 0:3].clamp(0.0, 1.0).pow(1.0 / 2.2) * 255.0
        return np.ascontiguousarray(c.cpu().numpy().astype(np.uint8)[::-1])
