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


def _qp_worker(rank, world, port, t, hw, c, n, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        state = {}

        def refine(t0, t1):  # "features" of frame f are the constant f + 1
            state["frames"] = (t0, t1)
            return (torch.arange(t0, t1, dtype=torch.float32) + 1)[:, None, None].expand(t1 - t0, hw, c).clone()

        def set_refined(vol):
            state["vol"] = vol.clone()

        def infer(q):  # a stand-in that needs EVERY frame of the gathered volume
            s = state["vol"].sum(dim=(1, 2))
            traj = q[:, None, :2] + s[None, :, None]
            return traj.contiguous(), (q[:, None, 0] + s[None]) % 2 > 1

        queries = torch.stack([torch.arange(n, dtype=torch.float32) * 3, torch.arange(n, dtype=torch.float32) * 5,
                               torch.zeros(n)], dim=1)
        out = sharding.query_parallel(refine, set_refined, infer, t, hw, c, queries, "cpu")
        assert state["frames"] == sharding.split_range(t, rank, world)[:2]
        assert torch.equal(state["vol"].sum(dim=(1, 2)), (torch.arange(t, dtype=torch.float32) + 1) * hw * c)
        if rank == 0:
            traj, occ = out
            s = (torch.arange(t, dtype=torch.float32) + 1) * hw * c
            assert traj.shape == (n, t, 2) and occ.shape == (n, t)
            assert torch.equal(traj, queries[:, None, :2] + s[None, :, None])
            assert torch.equal(occ, (queries[:, None, 0] + s[None]) % 2 > 1)
        else:
            assert out is None
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_query_parallel_world2():
    """SURVEY 8e level 2: frames split for P1 / P2, all-gather, queries split for P3, gather; ragged sizes."""
    world, t, hw, c, n = 2, 5, 3, 4, 7
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 30500 + (os.getpid() % 1000)
    mp.spawn(_qp_worker, args=(world, port, t, hw, c, n, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_split_range():
    assert [sharding.split_range(90, r, 8)[:2] for r in range(8)] == [(0, 12), (12, 24), (24, 36), (36, 48), (48, 60),
                                                                        (60, 72), (72, 84), (84, 90)]
    assert sharding.split_range(3, 3, 4) == (3, 3, 1) and sharding.split_range(5, 1, 2) == (3, 5, 3)


def test_query_parallel_world4_ragged():
    """World size 4 with T not divisible by the world size (T = 10: frames 3 / 3 / 3 / 1) and fewer queries than would fill the
    last rank (N = 9: 3 / 3 / 3 / 0 -- rank 3 has NO query and still takes part in both collectives)."""
    world, t, hw, c, n = 4, 10, 3, 4, 9
    assert [sharding.split_range(t, r, world)[:2] for r in range(world)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sharding.split_range(n, 3, world)[:2] == (9, 9)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 1000)
    mp.spawn(_qp_worker, args=(world, port, t, hw, c, n, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_video_parallel_gather_world4():
    world, n_videos, n, t = 4, 6, 5, 3  # 6 videos over 4 ranks: the second round has two idle ranks
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 32500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, n_videos, n, t, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_video_parallel_gather_world8_davis_batch():
    """north_star's 30-video batch on 8 ranks: 4/4/4/4/4/4/3/3 videos, four rounds, ranks 6 and 7 idle in the last one -- they
    still take part in its gather (no 8-GPU node has ever run this, the gloo form is what pins it)."""
    world, n_videos, n, t = 8, 30, 6, 5
    assert [len(sharding.videos_of_rank(n_videos, r, world)) for r in range(world)] == [4, 4, 4, 4, 4, 4, 3, 3]
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, n_videos, n, t, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_query_parallel_world8_t90():
    """One 90-frame video on 8 ranks: frames 12 x 7 + 6, queries 1024 -> 128 per rank (here 100 -> 13 x 7 + 9)."""
    world, t, hw, c, n = 8, 90, 2, 3, 100
    assert [sharding.split_range(t, r, world)[1] - sharding.split_range(t, r, world)[0] for r in range(world)] == [12] * 7 + [6]
    assert [sharding.split_range(n, r, world)[1] - sharding.split_range(n, r, world)[0] for r in range(world)] == [13] * 7 + [9]
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 34500 + (os.getpid() % 1000)
    mp.spawn(_qp_worker, args=(world, port, t, hw, c, n, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


# ---- round 6: the schedule that does not wait per round (sharding.run_scheduled) --------------------------------------------
DAVIS_LIKE = [50, 80, 43, 90, 66, 104, 25, 75, 70, 82, 40, 69, 100, 60, 84, 59, 90, 34, 50, 96, 72, 52, 78, 33, 68, 91, 45, 63, 80, 71]


def test_lpt_assignment_and_makespans():
    """Equal lengths: LPT = the round-robin counts; ragged (DAVIS-like 25 .. 104 frames): every video exactly once, LPT's
    makespan within Graham's bound of the ideal and strictly below the lock-step schedule's sum of round maxima."""
    assert [len(x) for x in sharding.lpt_assignment([1.0] * 30, 8)] == [4, 4, 4, 4, 4, 4, 3, 3]
    costs = [sharding.video_cost(t) for t in DAVIS_LIKE]
    for world in (1, 2, 4, 8):
        a = sharding.lpt_assignment(costs, world)
        assert sorted(v for lst in a for v in lst) == list(range(30))
        sc = sharding.schedule_costs(costs, world)
        assert abs(sc["lpt"] - max(sum(costs[v] for v in lst) for lst in a)) < 1e-9
        assert sc["ideal"] * (1 - 1e-12) <= sc["lpt"] <= (4 / 3 - 1 / (3 * world)) * sc["ideal"] + 1e-9
        assert sc["lpt"] <= sc["lockstep"] + 1e-9
    sc8 = sharding.schedule_costs(costs, 8)
    assert sc8["lpt"] < 0.85 * sc8["lockstep"], sc8          # the per-round waits cost > 15 % on this batch
    assert sc8["lpt"] < 1.05 * sc8["ideal"], sc8


def _ragged_track(v, n, t):
    g = torch.Generator().manual_seed(500 + v)
    return torch.rand(n, t, 2, generator=g) * 800, torch.rand(n, t, generator=g) > 0.5


def _sched_worker(rank, world, port, lengths, n, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls, clock = [], [0.0]
        costs = [sharding.video_cost(t) for t in lengths]

        def track(v):
            calls.append(v)
            clock[0] += costs[v]          # a virtual clock: this rank's busy time
            return _ragged_track(v, n, lengths[v])

        out = sharding.run_scheduled(lengths, n, "cpu", track)
        table = sharding.lpt_assignment(costs, world)
        assert calls == table[rank]                       # this rank's list, back to back, in LPT order
        if rank == 0:
            assert sorted(out) == list(range(len(lengths)))
            for v, t in enumerate(lengths):
                tr, oc = _ragged_track(v, n, t)
                assert out[v][0].shape == (n, t, 2) and torch.equal(out[v][0], tr) and torch.equal(out[v][1], oc)
        else:
            assert out is None
        ret[rank] = clock[0]
    finally:
        dist.destroy_process_group()


def test_run_scheduled_world8_ragged_davis_batch():
    """30 clips of 25 .. 104 frames on 8 ranks: one gather at the end, results identical to the per-video outputs, and the
    makespan of the run (the largest virtual busy time of a rank) IS max_r sum t -- not the lock-step sum of round maxima."""
    world, n = 8, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 35500 + (os.getpid() % 1000)
    mp.spawn(_sched_worker, args=(world, port, DAVIS_LIKE, n, ret), nprocs=world, join=True)
    busy = dict(ret)
    assert sorted(busy) == list(range(world))
    sc = sharding.schedule_costs([sharding.video_cost(t) for t in DAVIS_LIKE], world)
    assert abs(max(busy.values()) - sc["lpt"]) < 1e-6 and max(busy.values()) < 0.85 * sc["lockstep"]


def test_run_scheduled_world2_fewer_videos_than_ranks_and_single():
    """World 2 with ONE video (rank 1 has an empty list and still takes part in the gather), and the single-process form."""
    world, n = 2, 3
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 36500 + (os.getpid() % 1000)
    mp.spawn(_sched_worker, args=(world, port, [7], n, ret), nprocs=world, join=True)
    assert dict(ret) == {0: sharding.video_cost(7), 1: 0.0}
    out = sharding.run_scheduled([4, 9, 2], n, "cpu", lambda v: _ragged_track(v, n, [4, 9, 2][v]))
    assert sorted(out) == [0, 1, 2] and out[1][0].shape == (n, 9, 2) and torch.equal(out[2][1], _ragged_track(2, n, 2)[1])
