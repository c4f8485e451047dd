"""Multi-GPU layout of the per-video hot path: videos are independent, so rank r of a
one-process-per-GPU job takes videos v = r (mod world) and only the results travel -- one gather of the [N, T, 2]
fp32 trajectories and [N, T] occlusion flags per video to rank 0 over RCCL (xGMI); no collective touches the data path.
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
