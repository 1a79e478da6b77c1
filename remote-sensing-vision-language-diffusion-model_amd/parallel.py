"""Data-parallel sharding of independent images: one process per GPU, RCCL over xGMI.

The reference processes one image at a time (infer_dir.py:198-200; just_sampling is only called
with N=1) and has no collective on the hot path.  Images are independent units, so the only
exchange is ONE all-gather of the finished uint8 images per batch.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world, local_rank).  Initialises torch.distributed from RANK/WORLD_SIZE/MASTER_* when
    world > 1 (backend 'nccl' = RCCL on ROCm, 'gloo' on CPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # RSVLD_DIST_BACKEND=gloo: debugging aid (several ranks sharing one GPU)
            backend = os.environ.get("RSVLD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_images, rank, world):
    """Image i is processed by rank i % world (round-robin keeps per-rank load within one image)."""
    return list(range(rank, n_images, world))


def to_uint8(x):
    """[-1,1] fp32 NCHW -> uint8, the quantisation the reference applies at the Stage-1 hand-off
    (utils/tensor2img.py:4-21): clamp, (x+1)/2*255, round."""
    return ((x.clamp(-1, 1) + 1) * 0.5 * 255.0).round().to(torch.uint8)


def gather_images(local_u8, world):
    """All-gather equally-shaped uint8 image batches; returns [world*B, C, H, W] ordered by rank."""
    if world == 1:
        return local_u8
    shape = (world * local_u8.shape[0],) + tuple(local_u8.shape[1:])
    if dist.get_backend() == "gloo" and local_u8.is_cuda:   # gloo gathers host tensors
        out = torch.empty(shape, dtype=local_u8.dtype)
        dist.all_gather_into_tensor(out, local_u8.cpu().contiguous())
        return out.to(local_u8.device)
    out = torch.empty(shape, dtype=local_u8.dtype, device=local_u8.device)   # rank-major concatenation
    dist.all_gather_into_tensor(out, local_u8.contiguous())
    return out


def unshard(gathered, n_images, world):
    """Invert shard_indices: ``gathered`` is rank-major with ``per = ceil(n_images / world)`` rows per rank (short
    ranks pad their tail, gather_images needs equal shapes) -> the n_images real rows in image order."""
    per = gathered.shape[0] // world
    if per * world != gathered.shape[0] or per < -(-n_images // world):
        raise ValueError(f"unshard: {gathered.shape[0]} gathered rows cannot hold {n_images} images over {world} ranks")
    pos = torch.empty(n_images, dtype=torch.long)
    for r in range(world):
        for j, i in enumerate(shard_indices(n_images, r, world)):
            pos[i] = r * per + j                     # image i sits in rank r's j-th row; pad rows are never addressed
    return gathered[pos.to(gathered.device)]


def run_sharded(process, n_images, rank, world):
    """Data-parallel driver of independent images (infer_dir.py:198-200): this rank runs ``process(i) -> uint8 [C,H,W]``
    for its images i = rank, rank + world, ...; ONE all-gather of the finished uint8 images; -> ``[n_images, C, H, W]``
    in image order on every rank.  ``process`` must derive all of its randomness from ``i`` (per-image seeds), so the
    result does not depend on the number of ranks."""
    if n_images < world:                             # checked on EVERY rank before any work or collective: all raise together
        raise ValueError(f"run_sharded: {world} ranks for {n_images} images (a rank without an image cannot join the gather)")
    mine = [process(i) for i in shard_indices(n_images, rank, world)]
    per = -(-n_images // world)
    while len(mine) < per:                           # equal shapes for the collective: pad with a copy, dropped by unshard
        mine.append(mine[-1])
    return unshard(gather_images(torch.stack(mine), world), n_images, world)
