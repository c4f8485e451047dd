"""Multi-GPU layout of the per-video hot path: videos are independent, so rank r of a
one-process-per-GPU job takes videos v = r (mod world) and only the results travel -- one gather of the [N, T, 2]
fp32 trajectories and [N, T] occlusion flags per video to rank 0 over RCCL (xGMI); no collective touches the data path.
`run_sharded` is the lock-step form (one gather per round of `world` videos); `run_scheduled` (round 6) assigns the videos
longest-first by frame count, lets every rank run its list back to back and gathers ONCE at the end -- what a batch of ragged
clips (DAVIS: 25 .. 104 frames) needs.
Level 2 (one video on several GPUs, for the single-video scaling curve): `query_parallel` splits the FRAMES over the
ranks for P1 / P2 (each rank encodes and refines T / world frames), all-gathers the refined volume once (T*HW*C fp32 =
1.1 GB at T = 90, C = 384: ~140 MB per rank over xGMI), splits the QUERIES over the ranks for P3 (every rank needs all
frames: a query is correlated against every frame) and gathers the [N/world, T, 3] results on rank 0.
The volume travels as fp32, not as the 16-bit unit-norm copy SURVEY 8e budgets (0.56 GB): everything that DECIDES a result
(bilinear sampling, the fp32 re-scoring of the arg-max candidates, the window correlations) reads the fp32 master, so a 16-bit
gather would make an 8-GPU run differ from a 1-GPU run; at xGMI rates the extra 0.5 GB is a few milliseconds of a step that
takes every rank > 100 ms.  At 8 ranks a rank encodes 12 / 12 / ... / 6 frames of a 90-frame video (`split_range`): the ViT
runs one pass per rank at that batch (bench.py prints it as config.vit_frames_per_rank).
`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def videos_of_rank(n_videos: int, rank: int, world: int) -> List[int]:
    """Round-robin video assignment: 30 DAVIS videos over 8 GPUs -> 4/4/4/4/4/4/3/3."""
    return list(range(rank, n_videos, world))


def pack_result(traj: torch.Tensor, occ: torch.Tensor) -> torch.Tensor:
    """[N,T,2] f32 + [N,T] bool -> one flat f32 payload (single collective per video)."""
    return torch.cat([traj.reshape(-1).to(torch.float32), occ.reshape(-1).to(torch.float32)])


def unpack_result(payload: torch.Tensor, n: int, t: int) -> Tuple[torch.Tensor, torch.Tensor]:
    traj = payload[: n * t * 2].reshape(n, t, 2)
    occ = payload[n * t * 2: n * t * 3].reshape(n, t) > 0.5
    return traj, occ


def gather_results(traj: Optional[torch.Tensor], occ: Optional[torch.Tensor], n: int, t: int, device,
                   group=None) -> Optional[List[Optional[Tuple[torch.Tensor, torch.Tensor]]]]:
    """Every rank contributes the result of its current video (or None when it has none left in this round);
    rank 0 receives the list indexed by rank, other ranks get None."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    has = traj is not None
    payload = pack_result(traj, occ) if has else torch.zeros(n * t * 3, dtype=torch.float32, device=device)
    payload = torch.cat([payload, torch.tensor([1.0 if has else 0.0], device=payload.device)])
    if world == 1:
        return [unpack_result(payload, n, t) if has else None]
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, bufs, dst=0, group=group)
    if rank != 0:
        return None
    return [unpack_result(b, n, t) if b[-1] > 0.5 else None for b in bufs]


def run_sharded(n_videos: int, n: int, t: int, device, track_fn, group=None) -> Optional[Dict[int, Tuple[torch.Tensor, torch.Tensor]]]:
    """Drive `track_fn(video_index) -> (traj [n,t,2], occ [n,t])` over this rank's videos; rank 0 returns
    {video_index: (traj, occ)} for ALL videos, the others None.  Ranks stay in lock-step per round so that the gather
    is a regular collective even when n_videos is not a multiple of the world size."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = videos_of_rank(n_videos, rank, world)
    rounds = (n_videos + world - 1) // world
    out: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
    for r in range(rounds):
        res = track_fn(mine[r]) if r < len(mine) else (None, None)
        got = gather_results(res[0], res[1], n, t, device, group)
        if got is not None:
            for src, item in enumerate(got):
                if item is not None:
                    out[r * world + src] = item
    return out if rank == 0 else None


# ---- round 6: a schedule that does not wait per round ------------------------------------------
# run_sharded gathers once per ROUND, so every round costs its slowest video: sum_rounds max_r t  >=  max_r sum t.  Nothing needs
# the results before the end of the batch, and real clips are ragged (DAVIS: 25 .. 104 frames; the step is ~linear in T for P1 / P2
# and ~quadratic for P3, whose anchor count grows with T).  run_scheduled: longest-processing-time-first assignment, every rank
# runs its list back to back with NO collective in between, ONE gather of everything at the end.  run_sharded stays as the
# lock-step form (and as the oracle of the tests).
def video_cost(frames: int) -> float:
    """Relative cost of one video of `frames` frames at fixed N: P1 + P2 linear in T (2.2 ms per frame at 854 x 476, ViT-S), P3
    ~ N T (1 + anchors per query) with anchors ~ T (0.78 ms per frame at T = 90) -- measured on the benchmark step, round 5."""
    return float(frames) * (1.0 + 0.0039 * float(frames))


def lpt_assignment(costs: List[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first: videos by decreasing cost (ties: lower index first), each to the least loaded rank (ties:
    lower rank).  Deterministic, so every rank computes the same table without communication.  Graham's bound: makespan <=
    (4/3 - 1/(3 world)) x optimum.  Equal costs reproduce the round-robin counts (30 over 8: 4/4/4/4/4/4/3/3)."""
    order = sorted(range(len(costs)), key=lambda v: (-costs[v], v))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for v in order:
        r = min(range(world), key=lambda i: (load[i], i))
        out[r].append(v)
        load[r] += costs[v]
    return out


def schedule_costs(costs: List[float], world: int) -> Dict[str, float]:
    """Makespans (in cost units) of the three schedules of a batch: ideal = max(mean load, largest video); lpt = max_r sum of the
    rank's list; lockstep = sum over rounds of the round's largest video (v = r mod world, one gather per round)."""
    a = lpt_assignment(costs, world)
    lpt = max((sum(costs[v] for v in lst) for lst in a), default=0.0)
    lock = sum(max(costs[r0:r0 + world]) for r0 in range(0, len(costs), world))
    ideal = max(sum(costs) / world, max(costs, default=0.0))
    return {"ideal": ideal, "lpt": lpt, "lockstep": lock}


def run_scheduled(lengths: List[int], n: int, device, track_fn, group=None, costs: Optional[List[float]] = None,
                  ) -> Optional[Dict[int, Tuple[torch.Tensor, torch.Tensor]]]:
    """A batch of len(lengths) videos (video v has lengths[v] frames, n queries) over the ranks: LPT assignment by `costs`
    (default video_cost(frames)), `track_fn(v) -> (traj [n, T_v, 2], occ [n, T_v])` for this rank's videos back to back, then
    ONE gather of a flat payload per rank (padded to the largest rank's payload: the sizes follow from the table every rank
    holds).  Rank 0 returns {v: (traj, occ)} for all videos, the others None."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    costs = [video_cost(t) for t in lengths] if costs is None else costs
    table = lpt_assignment(costs, world)
    sizes = [sum(n * lengths[v] * 3 for v in lst) for lst in table]
    parts = []
    for v in table[rank]:
        traj, occ = track_fn(v)
        assert traj.shape == (n, lengths[v], 2) and occ.shape == (n, lengths[v]), (v, tuple(traj.shape), tuple(occ.shape))
        parts.append(pack_result(traj, occ))
    payload = torch.zeros(max(sizes + [1]), dtype=torch.float32, device=device)
    if parts:
        mine = torch.cat(parts)
        payload[: mine.numel()] = mine
    if world > 1:
        bufs = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
        dist.gather(payload, bufs, dst=0, group=group)
        if rank != 0:
            return None
    else:
        bufs = [payload]
    out: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
    for r, lst in enumerate(table):
        off = 0
        for v in lst:
            k = n * lengths[v] * 3
            out[v] = unpack_result(bufs[r][off: off + k], n, lengths[v])
            off += k
    return out


def split_range(n: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous equal-size blocks (the last ones may be short or empty): (begin, end, block size)."""
    per = (n + world - 1) // world
    b = min(rank * per, n)
    return b, min(n, b + per), per


def query_parallel(refine_frames_fn, set_refined_fn, infer_fn, t: int, hw: int, c: int, queries: torch.Tensor, device,
                   group=None) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """One video on `world` ranks.
      refine_frames_fn(t0, t1) -> [t1 - t0, hw, c] fp32 refined features of frames t0..t1-1 (P1 + P2 of those frames)
      set_refined_fn(volume [t, hw, c])   installs the gathered volume in this rank's tracker
      infer_fn(queries [n, 3]) -> (traj [n, t, 2] f32, occ [n, t] bool)
    Returns (traj [N, t, 2], occ [N, t]) on rank 0, None elsewhere.  Two collectives: all-gather of the volume, gather
    of the results."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    t0, t1, per = split_range(t, rank, world)
    local = torch.zeros((per, hw, c), dtype=torch.float32, device=device)
    if t1 > t0:
        local[: t1 - t0] = refine_frames_fn(t0, t1)
    if world > 1:
        full = torch.empty((world * per, hw, c), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(full, local, group=group)
    else:
        full = local
    set_refined_fn(full[:t])
    n = queries.shape[0]
    q0, q1, qper = split_range(n, rank, world)
    payload = torch.zeros(qper * t * 3, dtype=torch.float32, device=device)
    if q1 > q0:
        traj, occ = infer_fn(queries[q0:q1].contiguous())
        packed = pack_result(traj, occ)
        payload[: (q1 - q0) * t * 2] = packed[: (q1 - q0) * t * 2]
        payload[qper * t * 2: qper * t * 2 + (q1 - q0) * t] = packed[(q1 - q0) * t * 2:]
    if world == 1:
        return unpack_result(payload, qper, t)[0][:n], unpack_result(payload, qper, t)[1][:n]
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, bufs, dst=0, group=group)
    if rank != 0:
        return None
    parts = [unpack_result(b, qper, t) for b in bufs]
    return torch.cat([p[0] for p in parts])[:n], torch.cat([p[1] for p in parts])[:n]


def query_parallel_step(trk, ex, mi, video: torch.Tensor, queries: torch.Tensor, device, stages=("extract", "refine", "track")):
    """bench.py's binding of `query_parallel` to a Tracker / VitExtractor / ModelInference triple."""
    from .delta_dino import refine_packed_subset
    g = trk.geom

    def refine(t0, t1):
        dino = ex.encode(video[t0:t1]) if "extract" in stages else trk._dino[t0:t1].contiguous()
        return refine_packed_subset(trk.delta_dino, video[t0:t1].contiguous(), dino, g)

    return query_parallel(refine, trk.set_refined_packed, mi.infer, g.T, g.ph * g.pw, g.C, queries, device)
