"""Multi-GPU path on CPU: the sharding arithmetic, and a world_size-2 gloo run of the one
exchange the path has (Grotto's cross-shard parity carry), checked against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fss_amd import shard


def test_batch_slices_partition():
    for n in (0, 1, 7, 1 << 20, (1 << 24) + 3):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard.batch_slice(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_domain_blocks():
    for world in (1, 2, 4, 8):
        blocks = [shard.domain_block(28, r, world) for r in range(world)]
        assert sum(n for _, n in blocks) == 1 << 28
        assert all(f % n == 0 for f, n in blocks)
        assert [f for f, _ in blocks] == [r * ((1 << 28) // world) for r in range(world)]
    with pytest.raises(ValueError):
        shard.domain_block(28, 0, 3)
    with pytest.raises(ValueError):
        shard.domain_block(2, 0, 8)


def test_grotto_carries():
    assert shard.grotto_carries([1, 0, 1, 1]) == [0, 1, 1, 0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seeds, cws, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        p = O.Prg(2)
        # every rank evaluates only its block: leaf control bits of its subtree, local running XOR
        full = O.grotto_eval_all(p, n, 1, seeds[1], cws)  # checker
        first, nl = shard.domain_block(n, rank, world)
        # local shard as a rank would produce it with carry-in 0: undo the global prefix parity
        prev = int(full[first - 1]) if first else 0
        local = torch.from_numpy((full[first:first + nl] ^ prev).copy())
        fixed = shard.grotto_fix_shard(local, dist)
        ok = bool((fixed.numpy() == full[first:first + nl]).all())
        lo, hi = shard.batch_slice(1000, rank, world)
        t = torch.tensor([hi - lo], dtype=torch.int64)
        dist.all_reduce(t)
        ret[rank] = ok and int(t.item()) == 1000
    finally:
        dist.destroy_process_group()


def test_two_rank_grotto_exchange_gloo(O, rng):
    n = 10
    seeds = np.stack([rng.i4(), rng.i4()])
    cws = O.grotto_gen(O.Prg(2), n, seeds, 777)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, seeds, cws, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]
