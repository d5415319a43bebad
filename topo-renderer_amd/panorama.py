"""360-degree panorama = 8 fixed 45-degree sectors, sharded by azimuth sector over the GPUs of one node.

The reference has no panorama and no multi-GPU path: each sector is one complete
reference frame.  Rank g of N renders sectors [8g/N, 8(g+1)/N) from a replicated DEM; the strip is kept
sector-major ([sector][row][col][rgba]) so every rank's share is one contiguous block and the only exchange
step is a single all-gather (RCCL over xGMI on the GPU box, gloo in the CPU tests).  The result is
bit-identical for every N by construction.
"""
from __future__ import annotations

N_SECTORS = 8


def sector_range(rank: int, world: int, n_sectors: int = N_SECTORS) -> range:
    if n_sectors % world != 0:
        raise ValueError(f"{n_sectors} sectors cannot be split evenly over {world} ranks")
    per = n_sectors // world
    return range(rank * per, (rank + 1) * per)


def gather_strip(dist, strip, rank: int, world: int):
    """All-gather in place: `strip` is the full sector-major tensor [n_sectors, H, SW, 4]; this rank's sectors are
    already written at their final position.  No-op for world == 1."""
    if world == 1 or dist is None:
        return strip
    rng = sector_range(rank, world, strip.shape[0])
    mine = strip[rng.start:rng.stop]
    dist.all_gather_into_tensor(strip.view(-1), mine.reshape(-1))
    return strip


def to_row_major(strip):
    """[n_sectors, H, SW, C] -> [H, n_sectors*SW, C] (the strip as one image)."""
    n, h, sw = strip.shape[0], strip.shape[1], strip.shape[2]
    rest = tuple(strip.shape[3:])
    perm = (1, 0, 2) + tuple(range(3, strip.ndim))
    if hasattr(strip, "permute"):
        return strip.permute(*perm).reshape((h, n * sw) + rest)
    return strip.transpose(perm).reshape((h, n * sw) + rest)
