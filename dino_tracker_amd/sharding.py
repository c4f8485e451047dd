"""Multi-GPU layout of the per-video hot path: videos are independent, so rank r of a
one-process-per-GPU job takes videos v = r (mod world) and only the results travel -- one gather of the [N, T, 2]
fp32 trajectories and [N, T] occlusion flags per video to rank 0 over RCCL (xGMI); no collective touches the data path.
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
