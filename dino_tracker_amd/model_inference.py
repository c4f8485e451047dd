"""ModelInference -- the reference's models/model_inference.py:78-216 API, executed as one device-resident
pipeline:

  1. q_n      = bilinear(F[t_n], query_n)                                    dtk_sample_points
  2. traj     = head(relu(cos(q_n, F[t])))           for all n, t            dtk_track        (N*T maps)
  3. S[n,t]   = bilinear(F[t], traj[n,t]);  cs[n,t] = cos(S[n,t_n], S[n,t])  dtk_sample_points + dtk_traj_cos_sims
  4. anchors  A_n = {a : cs[n,a] >= th};  G[n][a,t] = head(relu(cos(S[n,t], F[a])))
                                                                             dtk_build_anchor_sources + dtk_track
  5. occ[n,t] = (lower-median_a |G[n][a,t]-traj[n,a]| > tau_n) or cs[n,t] < th2     dtk_occlusion

No per-query Python loop.  Host synchronisations per infer(): one read-back of the anchor counts (which also is the
zero-anchor check: the reference raises there) plus the ones dtk_track(DTK_TRACK_MFMA) documents (one per call, for the
sizes of its second / third tier); DTK_TRACK_EXACT adds none.
`batch_size` is accepted for signature compatibility; it was a memory knob for the reference's per-call frame
gathers (model_inference.py:45-49,138) and has no effect on results.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .dataset import RangeNormalizer
from .tracker import Tracker


# ---- per-query helpers of the reference module (models/model_inference.py:8-74), kept for API parity ----------------
def generate_trajectory_input(query_point, video, start_t=None, end_t=None):
    """(source_points, source_frame_indices, target_frame_indices, frames_set_t) for Tracker.forward: the query repeated
    for frames [start_t, end_t), frame set = [t_query, start_t .. end_t-1]."""
    start_t = 0 if start_t is None else start_t
    end_t = video.shape[0] if end_t is None else end_t
    rest = end_t - start_t
    dev = video.device
    source_points = query_point.unsqueeze(0).repeat(rest, 1)
    frames_set_t = torch.cat([query_point[2:3].to(dev), torch.arange(start_t, end_t, device=dev).to(query_point.dtype)]).int()
    source_frame_indices = torch.zeros(rest, dtype=torch.long, device=dev)
    target_frame_indices = torch.arange(rest, dtype=torch.long, device=dev) + 1
    return source_points, source_frame_indices, target_frame_indices, frames_set_t


@torch.no_grad()
def generate_trajectory(query_point, video, model, range_normalizer, dst_range=(-1, 1), use_raw_features=False,
                        batch_size=None):
    """rest x 3 (x, y, t) for one query, through Tracker.forward (models/model_inference.py:37-57)."""
    batch_size = video.shape[0] if batch_size is None else batch_size
    out = []
    for start_t in range(0, video.shape[0], batch_size):
        end_t = min(start_t + batch_size, video.shape[0])
        inp = generate_trajectory_input(query_point, video, start_t, end_t)
        coords = range_normalizer.unnormalize(model(inp, use_raw_features=use_raw_features), dims=[0, 1], src=dst_range)
        out.append(torch.cat([coords, inp[-1][1:].to(torch.float32).unsqueeze(1)], dim=1))
    return torch.cat(out, dim=0)


@torch.no_grad()
def generate_trajectories(query_points, video, model, range_normalizer, dst_range=(-1, 1), use_raw_features=False,
                          batch_size=None):
    """N x rest x 3; per-query loop like the reference (ModelInference.compute_trajectories is the batched path)."""
    return torch.stack([generate_trajectory(q, video, model, range_normalizer, dst_range, use_raw_features, batch_size)
                        for q in query_points.to(dtype=torch.float32)])


class ModelInference(torch.nn.Module):
    def __init__(self, model: Tracker, range_normalizer: RangeNormalizer,
                 anchor_cosine_similarity_threshold: float = 0.5, cosine_similarity_threshold: float = 0.5) -> None:
        super().__init__()
        self.model = model
        self.model.eval()
        # model_inference.py:88-89 always recomputes; a volume cached with the CURRENT Delta-DINO weights is identical
        if self.model._refined is None or self.model.refined_is_stale():
            self.model.cache_refined_embeddings()
        self.range_normalizer = range_normalizer
        self.anchor_cosine_similarity_threshold = anchor_cosine_similarity_threshold
        self.cosine_similarity_threshold = cosine_similarity_threshold
        self._anchor_buf = None
        self._idx_cache = {}

    # ---- index plumbing -------------------------------------------------------------------------------------
    def _first_pass_indices(self, N: int, T: int, device):
        key = (N, T, str(device))
        if key not in self._idx_cache:
            t = torch.arange(T, device=device, dtype=torch.int32)
            n = torch.arange(N, device=device, dtype=torch.int32)
            # sources ordered (t, n): all queries against frame t share the feature frame
            src_row = n.repeat(T).contiguous()
            tgt = t.repeat_interleave(N).contiguous()
            out_idx = (n[None, :] * T + t[:, None]).reshape(-1).contiguous()
            t_of_nt = t.repeat(N).contiguous()
            self._idx_cache = {key: (src_row, tgt, out_idx, t_of_nt)}
        return self._idx_cache[key]

    # ---- stages -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def compute_trajectories(self, query_points: torch.Tensor, batch_size=None) -> torch.Tensor:
        """model_inference.py:97-107 -> N x T x 3 (x, y in pixels at model resolution; t)."""
        m = self.model
        g = m.geom
        q = query_points.to(m.device, torch.float32).contiguous()
        N, T = q.shape[0], g.T
        feats = m.features()
        src_row, tgt, out_idx, _ = self._first_pass_indices(N, T, q.device)
        tq = q[:, 2].to(torch.int32).contiguous()
        q_emb = ops.sample_points(g, feats[0], q[:, :2].contiguous(), tq)
        traj = torch.empty((N, T, 3), dtype=torch.float32, device=q.device)
        xy = torch.empty((N * T, 2), dtype=torch.float32, device=q.device)
        m.track_sources(feats, q_emb, src_row, tgt, out_idx, xy, N * T)
        traj[..., :2] = xy.view(N, T, 2)
        traj[..., 2] = torch.arange(T, device=q.device, dtype=torch.float32)[None]
        return traj

    def _sample_along(self, trajectories: torch.Tensor) -> torch.Tensor:
        m = self.model
        N, T = trajectories.shape[:2]
        _, _, _, t_of_nt = self._first_pass_indices(N, T, trajectories.device)
        xy = trajectories[..., :2].reshape(N * T, 2).to(torch.float32).contiguous()
        return ops.sample_points(m.geom, m.features()[0], xy, t_of_nt)  # [N*T][C]

    @torch.no_grad()
    def compute_trajectory_cos_sims(self, trajectories, query_points) -> torch.Tensor:
        """model_inference.py:110-126 -> N x T."""
        N, T = trajectories.shape[:2]
        S = self._sample_along(trajectories)
        tq = query_points[:, 2].to(trajectories.device).to(torch.int32).contiguous()
        self._last_S = S
        return ops.traj_cos_sims(S, tq, N, T)

    def _anchor_stage(self, trajectories, cos_sims, S=None):
        m = self.model
        N, T = trajectories.shape[:2]
        if S is None:
            S = self._sample_along(trajectories)
        buf = ops.build_anchor_sources(cos_sims.contiguous(), self.anchor_cosine_similarity_threshold, self._anchor_buf)
        self._anchor_buf = buf
        # one read-back of (pairs, sources, queries without anchors): sizes the anchor launches exactly and is the
        # zero-anchor check of infer() as well
        self.last_counts = buf.counts.cpu()
        n_src = int(self.last_counts[1])
        green = torch.empty((N * T, T, 2), dtype=torch.float32, device=trajectories.device)
        if n_src > 0:
            m.track_sources(m.features(), S, buf.src_row, buf.tgt, buf.out_idx, green, n_src)
        return buf, green

    @torch.no_grad()
    def compute_anchor_trajectories(self, trajectories: torch.Tensor, cos_sims: torch.Tensor, batch_size=None) -> Dict[int, torch.Tensor]:
        """model_inference.py:156-165 -> {n: A_n x T x 2}.  (Builds the Python dict, hence one host sync.)"""
        buf, green = self._anchor_stage(trajectories, cos_sims)
        off = buf.pair_off.cpu().tolist()
        out = {}
        for n in range(trajectories.shape[0]):
            if off[n + 1] == off[n]:
                raise RuntimeError("stack expects a non-empty TensorList")  # torch.stack([]) at model_inference.py:152
            out[n] = green[off[n]:off[n + 1]]
        return out

    @torch.no_grad()
    def compute_occlusion(self, trajectories: torch.Tensor, trajs_cos_sims: torch.Tensor,
                          anchor_trajectories: Dict[int, torch.Tensor]) -> torch.Tensor:
        """model_inference.py:179-200 from the dict form -> N x T bool."""
        N, T = trajectories.shape[:2]
        dev = trajectories.device
        green = torch.cat([anchor_trajectories[n].reshape(-1, T, 2) for n in range(N)]).to(torch.float32).contiguous()
        sizes = torch.tensor([anchor_trajectories[n].shape[0] for n in range(N)], dtype=torch.int32)
        pair_off = torch.zeros(N + 1, dtype=torch.int32)
        pair_off[1:] = torch.cumsum(sizes, 0)
        flags = trajs_cos_sims >= self.anchor_cosine_similarity_threshold
        pair_frame = torch.nonzero(flags)[:, 1].to(torch.int32).contiguous()
        if pair_frame.numel() != int(pair_off[-1]):
            raise RuntimeError("anchor_trajectories do not match the anchors implied by trajs_cos_sims")
        return ops.occlusion(green, pair_off.to(dev), pair_frame, trajectories[..., :2].to(torch.float32).contiguous(),
                             trajs_cos_sims.contiguous(), self.anchor_cosine_similarity_threshold,
                             self.cosine_similarity_threshold)

    # ---- whole pipeline (model_inference.py:203-216) ------------------------------------------------------------
    @torch.no_grad()
    def infer(self, query_points: torch.Tensor, batch_size=None):
        trajs = self.compute_trajectories(query_points, batch_size)
        N, T = trajs.shape[:2]
        S = self._sample_along(trajs)
        tq = query_points[:, 2].to(trajs.device).to(torch.int32).contiguous()
        cos_sims = ops.traj_cos_sims(S, tq, N, T)
        buf, green = self._anchor_stage(trajs, cos_sims, S)
        traj_xy = trajs[..., :2].contiguous()
        occ = ops.occlusion(green, buf.pair_off, buf.pair_frame, traj_xy, cos_sims,
                            self.anchor_cosine_similarity_threshold, self.cosine_similarity_threshold)
        if int(self.last_counts[2]) > 0:
            raise RuntimeError("stack expects a non-empty TensorList")  # a query without anchors (model_inference.py:152)
        return traj_xy, occ
