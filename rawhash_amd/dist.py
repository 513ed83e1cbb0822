"""Multi-GPU helpers (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" in CPU tests).

The path shards by reads -- contiguous blocks per rank, no collective on the data path -- and replicates the index:
rank 0 uploads it, the flattened device blob is broadcast once and adopted by every rank's context.
"""
import ctypes as C
import os

import numpy as np


def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of n reads for `rank` (first n % world ranks get one extra read)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def replicate_index(ctx, opts, index=None, device="cuda", src=0):
    """Broadcast rank `src`'s resident index blob (+ calibrated mid_occ) to every rank and adopt it.

    Returns the torch tensor that owns the replica's memory on non-source ranks (keep it alive as long as ctx)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    hdr_t = torch.zeros(256, dtype=torch.uint8, device=device)
    if rank == src:
        if index is not None:
            ctx.upload(index)
        _, nbytes, hdr = ctx.device_blob()
        meta[0], meta[1] = nbytes, opts.mo.mid_occ
        hdr_t.copy_(torch.frombuffer(bytearray(hdr), dtype=torch.uint8))
    dist.broadcast(meta, src)
    dist.broadcast(hdr_t, src)
    nbytes = int(meta[0].item())
    opts.mo.mid_occ = int(meta[1].item())
    if opts.mo.bw_long < opts.mo.bw:      # what ri_mapopt_update also does (rindex.c:1052)
        opts.mo.bw_long = opts.mo.bw
    blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if rank == src:
        ctx.copy_blob_to(blob.data_ptr())
    # a human-scale blob is 45 GB: broadcast in pieces that stay below 2^31 elements per collective (the count of a collective is
    # an int in places down the stack) and let consecutive pieces pipeline over the xGMI ring
    piece = int(os.environ.get("RH_BCAST_PIECE_BYTES", str(1 << 30)))
    for o in range(0, nbytes, piece):
        dist.broadcast(blob[o:o + piece], src)
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()
    if rank != src:
        ctx.adopt_blob(blob.data_ptr(), nbytes, bytes(hdr_t.cpu().numpy().tobytes()), take_ownership=False)
        return blob
    # the source keeps its own resident index: the staging copy goes back to the device (not into torch's cache: the mapping
    # arenas are sized by what hipMemGetInfo reports free)
    del blob
    if str(device).startswith("cuda"):
        torch.cuda.empty_cache()
    return None


def gather_records(recs, first_read, dst=0):
    """Concatenate per-rank record arrays in read order on rank `dst` (read_idx rebased to the global numbering)."""
    import torch.distributed as dist
    mine = recs.copy()
    mine["read_idx"] += np.uint32(first_read)
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object((first_read, mine.tobytes()), parts, dst=dst)
    if dist.get_rank() != dst:
        return None
    parts.sort(key=lambda p: p[0])
    return np.concatenate([np.frombuffer(b, dtype=recs.dtype) for _, b in parts])
