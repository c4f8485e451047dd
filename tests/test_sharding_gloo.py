"""CPU, world_size 2 over gloo: the N>1 path of the job (video sharding + one gather of results per round)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dino_tracker_amd import sharding


def _fake_track(v, n, t):
    g = torch.Generator().manual_seed(100 + v)
    return torch.rand(n, t, 2, generator=g) * 800, torch.rand(n, t, generator=g) > 0.5


def _worker(rank, world, port, n_videos, n, t, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def track(v):
            calls.append(v)
            return _fake_track(v, n, t)

        out = sharding.run_sharded(n_videos, n, t, "cpu", track)
        assert calls == sharding.videos_of_rank(n_videos, rank, world)
        if rank == 0:
            assert sorted(out) == list(range(n_videos))
            for v in range(n_videos):
                tr, oc = _fake_track(v, n, t)
                assert torch.equal(out[v][0], tr) and torch.equal(out[v][1], oc)
        else:
            assert out is None
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_video_parallel_gather_world2():
    world, n_videos, n, t = 2, 5, 7, 4  # odd number of videos: the last round has an idle rank
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, n_videos, n, t, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_assignment_matches_davis_layout():
    sizes = [len(sharding.videos_of_rank(30, r, 8)) for r in range(8)]
    assert sizes == [4, 4, 4, 4, 4, 4, 3, 3] and sum(sizes) == 30
    assert sorted(v for r in range(8) for v in sharding.videos_of_rank(30, r, 8)) == list(range(30))
