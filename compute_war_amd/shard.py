"""Multi-GPU sharding of the block index space.

Blocks are independent units of work (the reference hands whole blocks to any worker thread,
src/hashandcompress/HashAndCompress.cpp:263-269), so rank g of G owns a contiguous range of block indices and
no collective sits on the data path.  The only exchange is the result gather: digests (all_gather) and the
{bytes_in, bytes_out, stored_raw} totals (all_reduce) -- ``torch.distributed`` with backend "nccl" (= RCCL over
xGMI) on GPUs, "gloo" in the CPU tests.  Works on whatever device the tensors live on.
"""
from __future__ import annotations


def shard_range(nblocks: int, rank: int, world: int) -> tuple[int, int]:
    """[first, last) of the blocks rank owns: g*N/G .. (g+1)*N/G (contiguous, sizes differ by at most 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * nblocks) // world, ((rank + 1) * nblocks) // world


def gather_results(digests, totals, world: int, async_op: bool = False):
    """digests: [n_local, digest_bytes] uint8 (equal n_local on every rank); totals: int64[k].
    Returns (all_digests [world*n_local, digest_bytes], totals summed over ranks, work handles)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return digests, totals, []
    out = torch.empty((world * digests.shape[0], digests.shape[1]), dtype=digests.dtype, device=digests.device)
    h1 = dist.all_gather_into_tensor(out, digests.contiguous(), async_op=async_op)
    h2 = dist.all_reduce(totals, op=dist.ReduceOp.SUM, async_op=async_op)
    return out, totals, [h for h in (h1, h2) if h is not None]
