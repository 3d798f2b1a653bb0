"""Read-sharded multi-GPU exchange: one process per GPU, every rank holds
the full graph index and a shard of the reads.  The path has exactly one data-path exchange --
the all-reduce (MAX) of the per-minimiser hit vector -- plus a once-per-job merge of the distinct
read hashes so that |Sp_R| (ILP_index.cpp:641) and the filtered/retained counters are global.

torch.distributed is plumbing here ("nccl" is RCCL on ROCm; "gloo" in the CPU tests); the
functions work on any tensors of the right dtype so the same code runs under both backends.
"""
import torch
import torch.distributed as dist


class DevArray:
    """Zero-copy view of a raw device buffer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, n, typestr="|u1"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def shard_bounds(read_off, world, rank):
    """Contiguous shard of reads [lo, hi) for `rank`, balanced by bases."""
    n = len(read_off) - 1
    total = int(read_off[-1])
    import numpy as np
    cuts = np.searchsorted(read_off, [total * r // world for r in range(world + 1)], side="left")
    cuts[0], cuts[-1] = 0, n
    return int(cuts[rank]), int(cuts[rank + 1])


def _staged():
    """gloo (CPU tests, single-GPU rehearsals of the N > 1 path) moves device tensors through the host."""
    return dist.get_backend() == "gloo"


def allreduce_hits(hit):
    """In-place MAX all-reduce of the uint8 hit vector (one byte per distinct walk minimiser)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if _staged() and hit.is_cuda:
            h = hit.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            hit.copy_(h)
        else:
            dist.all_reduce(hit, op=dist.ReduceOp.MAX)
    return hit


def gather_spectra(mine):
    """mine: int64 tensor of this rank's distinct read hashes.  Returns the list of every rank's
    tensor (own included), via a size exchange + padded all_gather."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [mine]
    world = dist.get_world_size()
    if _staged() and mine.is_cuda:
        dev = mine.device
        return [t.to(dev) for t in gather_spectra(mine.cpu())]
    n = torch.tensor([mine.numel()], dtype=torch.int64, device=mine.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.int64, device=mine.device)
    pad[:mine.numel()] = mine
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [bufs[r][:sizes[r]] for r in range(world)]


def merge_spectrum_into(ctx, device):
    """Make ctx's read-spectrum set the union over ranks (phi_spectrum_export / _import)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    rank = dist.get_rank()
    # the tensors below are made on torch's current stream, the imports run on the context's: unless the
    # caller bound both to one explicit stream they are not ordered, so wait on both sides
    torch.cuda.current_stream().synchronize()
    p, n = ctx.spectrum_export()
    mine = torch.as_tensor(DevArray(p, n, "<i8"), device=device).clone() if n else torch.zeros(0, dtype=torch.int64, device=device)
    parts = gather_spectra(mine)
    torch.cuda.current_stream().synchronize()          # the gathered lists are complete before another stream reads them
    for r, t in enumerate(parts):
        if r != rank and t.numel():
            t = t.contiguous()
            ctx.spectrum_import(t.data_ptr(), t.numel())
    torch.cuda.synchronize()          # the imported buffers must outlive the insert kernels
