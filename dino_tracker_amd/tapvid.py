"""inference_benchmark.py's loop and eval/metrics.py's scoring without leaving the device.

  * `infer_benchmark`: the reference calls ModelInference.infer once per query START FRAME (inference_benchmark.py:36-42);
    all start frames share the refined feature volume, so here they are ONE infer() call (mixed query frames are what
    the batched pipeline handles anyway) and the result is split per frame afterwards.
  * `tapvid_metrics`: compute_tapvid_metrics_for_video (eval/metrics.py:150-223) from device tensors through
    dtk_tapvid_counts -- the five-threshold Jaccard / points-within / occlusion-accuracy numbers, no .npy round trip.
"""
from __future__ import annotations

from typing import Dict, Mapping, Sequence, Tuple

import torch

from . import ops
from ._lib import check, lib

THRESHOLDS = (1, 2, 4, 8, 16)


@torch.no_grad()
def infer_benchmark(model_inference, query_points: Mapping[int, Sequence], device=None,
                    batch_size=None) -> Dict[int, Tuple[torch.Tensor, torch.Tensor]]:
    """query_points {start_frame: [[x, y, start_frame], ...]} (data/tapvid.py:19-41, already at model resolution) ->
    {start_frame: (trajectories [n, T, 2], occlusions [n, T])}, the tensors inference_benchmark.py saves per frame."""
    device = device or model_inference.model.device
    frames = sorted(query_points.keys())
    parts = [torch.as_tensor(query_points[f], dtype=torch.float32, device=device).reshape(-1, 3) for f in frames]
    sizes = [p.shape[0] for p in parts]
    traj, occ = model_inference.infer(torch.cat(parts), batch_size=batch_size)
    out, pos = {}, 0
    for f, n in zip(frames, sizes):
        out[f] = (traj[pos:pos + n], occ[pos:pos + n])
        pos += n
    return out


def metrics_from_counts(counts: Sequence[int]) -> Dict[str, float]:
    """eval/metrics.py:84-146: ratios of the 18 counts, then the means over the five thresholds."""
    ev, occ_eq, vis = counts[0], counts[1], counts[2]

    def ratio(a, b):  # a video without evaluated / visible points: nan, like numpy's 0/0 in eval/metrics.py, not an exception
        return a / b if b else float("nan")

    m = {"occlusion_accuracy": ratio(occ_eq, ev)}
    fr, ja = [], []
    for i, th in enumerate(THRESHOLDS):
        correct, tp, fp = counts[3 + 3 * i], counts[4 + 3 * i], counts[5 + 3 * i]
        m[f"pts_within_{th}"] = ratio(correct, vis)
        m[f"jaccard_{th}"] = ratio(tp, vis + fp)
        fr.append(m[f"pts_within_{th}"])
        ja.append(m[f"jaccard_{th}"])
    m["average_jaccard"] = sum(ja) / len(ja)
    m["average_pts_within_thresh"] = sum(fr) / len(fr)
    return m


@torch.no_grad()
def tapvid_counts(pred_tracks: torch.Tensor, pred_occluded: torch.Tensor, gt_tracks: torch.Tensor,
                  gt_occluded: torch.Tensor, query_frames: torch.Tensor, pred_size, gt_size,
                  query_mode: str = "strided") -> torch.Tensor:
    """18 uint64 counts (dtk.h) as an int64 tensor on the device.  pred_size / gt_size = (w, h) of the rasters the two
    track sets live in; both are scaled to 256 x 256 like eval/metrics.py:204-211."""
    n, t = pred_tracks.shape[:2]
    dev = pred_tracks.device
    counts = torch.zeros(18, dtype=torch.int64, device=dev)
    f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))  # noqa: E731  (numpy multiplies by the float32 value)
    # locals keep the converted tensors alive until the launch is enqueued (a temporary released right after its
    # data_ptr() was taken can be handed out again by the caching allocator for the NEXT temporary)
    pt = pred_tracks.to(torch.float32).contiguous()
    po = pred_occluded.to(torch.uint8).contiguous()
    gt = gt_tracks.to(dev, torch.float32).contiguous()
    go = gt_occluded.to(dev).to(torch.uint8).contiguous()
    qf = query_frames.to(dev).to(torch.int32).contiguous()
    check(lib().dtk_tapvid_counts(ops._p(pt), ops._p(po), ops._p(gt), ops._p(go), ops._p(qf), f32(256 / pred_size[0]),
                                  f32(256 / pred_size[1]), f32(256 / gt_size[0]), f32(256 / gt_size[1]),
                                  int(query_mode == "first"), n, t, ops._p(counts), ops._stream()))
    return counts


@torch.no_grad()
def tapvid_metrics(results: Mapping[int, Tuple[torch.Tensor, torch.Tensor]], video_config: Mapping, pred_size=(854, 476),
                   query_mode: str = "strided") -> Dict[str, float]:
    """compute_tapvid_metrics_for_video (eval/metrics.py:150-223) for `results` = infer_benchmark's output and one video
    entry of the benchmark pickle ({"h", "w", "query_points", "target_points", "occluded"}, data/tapvid.py)."""
    frames = list(video_config["query_points"])
    dev = results[frames[0]][0].device
    pred = torch.cat([results[f][0][..., :2] for f in frames])
    pocc = torch.cat([results[f][1] for f in frames])
    gt = torch.cat([torch.as_tensor(video_config["target_points"][f], dtype=torch.float32) for f in frames]).to(dev)
    gocc = torch.cat([torch.as_tensor(video_config["occluded"][f].astype(bool)) for f in frames]).to(dev)
    qf = torch.cat([torch.full((len(video_config["query_points"][f]),), int(f), dtype=torch.int32) for f in frames])
    counts = tapvid_counts(pred, pocc, gt, gocc, qf, pred_size, (video_config["w"], video_config["h"]), query_mode)
    return metrics_from_counts(counts.cpu().tolist())
