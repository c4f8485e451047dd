"""TEST INFRASTRUCTURE -- stand-in for `torchvision.ops` (torchvision is not installed here; the reference imports
`batched_nms` at preprocessing_dino_bb/compute_dino_bb_nms.py:4).  PARITY UNPINNED for this one op: restated from the
published behaviour of torchvision 0.16 (`torchvision/ops/boxes.py`, `csrc/ops/cpu/nms_kernel.cpp`):

nms(boxes [N,4] (x1,y1,x2,y2), scores [N], iou_threshold): visit boxes in DESCENDING score order; a box that is not yet
suppressed is kept and suppresses every later box whose IoU with it is > iou_threshold, with
IoU = inter / (area_i + area_j - inter), inter = max(0, min(x2) - max(x1)) * max(0, min(y2) - max(y1)) in fp32.
Returns the kept indices in descending score order.
batched_nms(boxes, scores, idxs, iou_threshold): nms within each group of equal idxs (torchvision offsets the boxes of
group g by g * (max coordinate + 1), or loops over the groups when there are more than 4000 boxes; both are the per-group
nms), kept indices of all groups together in descending score order.
"""
import torch


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    x1, y1, x2, y2 = boxes.float().unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    order = torch.argsort(scores, descending=True, stable=True)
    suppressed = torch.zeros(boxes.shape[0], dtype=torch.bool, device=boxes.device)
    keep = []
    for i in order.tolist():
        if suppressed[i]:
            continue
        keep.append(i)
        iw = (torch.minimum(x2[i], x2) - torch.maximum(x1[i], x1)).clamp(min=0)
        ih = (torch.minimum(y2[i], y2) - torch.maximum(y1[i], y1)).clamp(min=0)
        inter = iw * ih
        ovr = inter / (areas[i] + areas - inter)
        suppressed |= ovr > iou_threshold
    return torch.tensor(keep, dtype=torch.int64, device=boxes.device)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for g in torch.unique(idxs):
        members = torch.where(idxs == g)[0]
        keep_mask[members[nms(boxes[members], scores[members], iou_threshold)]] = True
    keep = torch.where(keep_mask)[0]
    return keep[scores[keep].sort(descending=True)[1]]
