"""N>1 path on CPU: scene sharding + the reporting collectives over gloo, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graspnerf_amd.sharding import scene_shard, max_over_ranks, sum_over_ranks, gather_volumes


def test_scene_shard_partitions_exactly():
    for total in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 8):
            owned = []
            for r in range(world):
                lo, hi = scene_shard(total, r, world)
                assert 0 <= lo <= hi <= total
                owned += list(range(lo, hi))
            assert owned == list(range(total))
    assert scene_shard(256, 3, 8) == (96, 128)            # BASELINE config 4: 32 scenes / GPU
    with pytest.raises(ValueError):
        scene_shard(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = scene_shard(total, rank, world)
        # stand-in for the per-rank forward: volume[b] is a function of the GLOBAL scene id only
        vols = torch.stack([torch.full((1, 2, 2, 2), float(i)) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 1, 2, 2, 2)
        allv = gather_volumes(vols)
        assert allv.shape[0] == total and torch.equal(allv[:, 0, 0, 0, 0], torch.arange(total, dtype=torch.float32))
        assert max_over_ranks(1.0 + rank) == float(world)
        assert sum_over_ranks(hi - lo) == float(total)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('total', [8, 5])
def test_two_rank_gloo(total):
    mp.spawn(_worker, args=(2, _free_port(), total), nprocs=2, join=True)
