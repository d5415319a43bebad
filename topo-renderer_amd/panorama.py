"""360-degree panorama = 8 fixed 45-degree sectors, sharded by azimuth sector over the GPUs of one node.

The reference has no panorama and no multi-GPU path: each sector is one complete
reference frame.  Rank g of N renders sectors [8g/N, 8(g+1)/N) from a replicated DEM and the strip is
assembled on every rank with all-gathers (RCCL over xGMI on the GPU box, gloo in the CPU tests).  The result
is bit-identical for every N by construction.

Exchange and rendering are pipelined: a rank renders its sectors one at a time and the all-gather of sector
slot c (every rank's c-th sector) runs while slot c+1 is being rendered, so for N < 8 the render time hides
under the (longer) exchange.  The strip is therefore stored slot-major,

    strip[c][r] = sector r*per + c        shape [per, world, H, SW, 4]

so that each all-gather writes one contiguous block; `sector(strip, k)` / `to_row_major` give the views.
"""
from __future__ import annotations

N_SECTORS = 8


def sector_range(rank: int, world: int, n_sectors: int = N_SECTORS) -> range:
    if n_sectors % world != 0:
        raise ValueError(f"{n_sectors} sectors cannot be split evenly over {world} ranks")
    per = n_sectors // world
    return range(rank * per, (rank + 1) * per)


def strip_shape(world: int, h: int, sw: int, channels: int = 4, n_sectors: int = N_SECTORS):
    per = len(sector_range(0, world, n_sectors))
    return (per, world, h, sw, channels)


def slot_of(k: int, world: int, n_sectors: int = N_SECTORS):
    """(c, r): where sector k lives in the slot-major strip."""
    per = n_sectors // world
    return k % per, k // per


def sector(strip, k: int):
    c, r = slot_of(k, strip.shape[1], strip.shape[0] * strip.shape[1])
    return strip[c, r]


def gather_slot(dist, strip, c: int, rank: int, world: int, async_op: bool = False):
    """All-gather slot c in place: this rank's sector is already at strip[c, rank].  Returns the work handle when
    async_op (None for world == 1)."""
    if world == 1 or dist is None:
        return None
    out = strip[c]                      # [world, H, SW, 4], contiguous
    return dist.all_gather_into_tensor(out.view(-1), out[rank].reshape(-1), async_op=async_op)


def gather_sector_major(dist, strip, rank: int, world: int, async_op: bool = False):
    """The layout of the C ABI (topo_render_panorama): strip[8][H][SW][C] in sector order, rank g's sectors
    [8g/N, 8(g+1)/N) one contiguous block, ONE in-place all-gather per panorama.  This rank's block must already be in
    place.  Returns the work handle when async_op (None for world == 1)."""
    if world == 1 or dist is None:
        return None
    per = strip.shape[0] // world
    return dist.all_gather_into_tensor(strip.view(-1), strip[rank * per:(rank + 1) * per].reshape(-1), async_op=async_op)


def exchange_slot(dist, strip, slot, rank: int, world: int):
    """One slot of topo_render_panorama's exchange plan (topo_panorama_slots) on the sector-major strip [8][H][SW][C], with
    torch.distributed point-to-point operations in the place of the grouped ncclSend / ncclRecv of the C ABI: this rank's
    band (sector `first + slot.sector`, rows row0 .. row0 + rows) goes to every other rank, theirs come straight into
    place.  Returns the outstanding requests (wait on all of them before reading the strip)."""
    if world == 1 or dist is None:
        return []
    s, row0, rows = slot
    per = strip.shape[0] // world
    ops = []
    for p in range(world):
        if p == rank:
            continue
        ops.append(dist.P2POp(dist.isend, strip[rank * per + s, row0:row0 + rows], p))
        ops.append(dist.P2POp(dist.irecv, strip[p * per + s, row0:row0 + rows], p))
    return dist.batch_isend_irecv(ops)


def to_row_major(strip):
    """[per, world, H, SW, C] -> [H, n_sectors*SW, C] (the strip as one image, sectors left to right)."""
    per, world, h, sw = strip.shape[0], strip.shape[1], strip.shape[2], strip.shape[3]
    rest = tuple(strip.shape[4:])
    # sector k = r*per + c  ->  order (r, c)
    perm = (2, 1, 0, 3) + tuple(range(4, strip.ndim))
    if hasattr(strip, "permute"):
        return strip.permute(*perm).reshape((h, world * per * sw) + rest)
    return strip.transpose(perm).reshape((h, world * per * sw) + rest)
