"""Sharding of an evaluation job across the GPUs of one node.

The points x keys batch and the EvalAll sub-domains are independent, so the data path
has no collective: rank r evaluates its slice and keeps the outputs on its GPU.  The
only exchange on this path is Grotto's EvalAll, whose running XOR crosses shard
boundaries: each rank needs the parity of all leaves before its block -- one byte per
rank, all-gathered (torch.distributed; RCCL on GPUs, gloo in the CPU tests).

The slice arithmetic is the library's (include/fss_amd.h: fss_amd_batch_slice,
fss_amd_domain_block, fss_amd_grotto_carries): one definition for C, C++
(fss::gpu::DeviceSet) and these wrappers.  Pure host functions: they need libfss_amd.so
loaded, not a GPU.
"""
import ctypes as C

from . import _lib


def batch_slice(n, rank, world):
    """Contiguous [lo, hi) of an n-element batch for `rank`; sizes differ by at most 1."""
    lo, hi = C.c_int64(), C.c_int64()
    if _lib.lib().fss_amd_batch_slice(int(n), int(rank), int(world), C.byref(lo), C.byref(hi)) != 0:
        raise ValueError("bad rank/world")
    return lo.value, hi.value


def domain_block(in_bits, rank, world):
    """(first_leaf, nleaves) of rank's EvalAll block: world must be a power of two no larger
    than the domain; rank owns the subtree whose path is its log2(world)-bit prefix."""
    first, nleaves = C.c_int64(), C.c_int64()
    if world < 1 or world & (world - 1) or not (0 <= rank < world):
        raise ValueError("world must be a power of two")
    if in_bits <= 62 and world > (1 << in_bits):
        raise ValueError("more ranks than leaves")
    if _lib.lib().fss_amd_domain_block(int(in_bits), int(rank), int(world), C.byref(first), C.byref(nleaves)) != 0:
        raise ValueError("bad in_bits/rank/world")
    return first.value, nleaves.value


def grotto_carries(last_bits):
    """Exclusive XOR scan of the per-shard parities (the last running-XOR byte of each
    shard computed with carry-in 0) -> carry-in of every shard."""
    world = len(last_bits)
    src = (C.c_uint8 * world)(*[int(b) & 1 for b in last_bits])
    dst = (C.c_uint8 * world)()
    if _lib.lib().fss_amd_grotto_carries(src, world, dst) != 0:
        raise ValueError("bad shard list")
    return [int(v) for v in dst]


def grotto_fix_shard(ys_local, dist=None):
    """ys_local: this rank's EvalAll block (uint8 running XOR with carry-in 0).  All-gathers the
    shard parities and applies this rank's carry in place.  With dist=None it is a no-op."""
    import torch
    if dist is None or dist.get_world_size() == 1:
        return ys_local
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = ys_local[-1:].to(torch.uint8).clone()
    if dist.get_backend() == "gloo":
        mine = mine.cpu()  # gloo gathers host tensors only (the CPU tests, and ranks sharing one GPU)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    carry = grotto_carries([int(p.item()) for p in parts])[rank]
    if carry:
        ys_local ^= 1
    return ys_local
