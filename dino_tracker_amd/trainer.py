"""Test-time training: the iteration of the reference's trainer (dino_tracker.py:392-448) and its loss terms
(dino_tracker.py:128-352) restated for the device -- static shapes, every (source, target) frame pair of a loss in one
batch, no device -> host read inside the iteration.

The reference's loop is host-bound by construction: per iteration ~300 `.item()` reads (`torch.tensor([idx] * n)` with a
device scalar `idx`, dino_tracker.py:194-201; `f'{source_frame}_{target_frame}'` keys, :181), ~250 small pageable
host -> device copies, data-dependent shapes (boolean-mask indexing before every `randperm(len)`), a `while .any()` re-draw and
`torch.cuda.empty_cache()` (:404).  Run un-modified on this implementation's models that was 4 300 launches and ~540 blocking
copies per iteration (profiles/r03_train_host_profile.txt) around 45 ms of convolution kernels.  Here:

  * random subsets of data-dependent sets (foreground best buddies of a frame pair, mutual nearest neighbours of the refined
    features, pixels of a mask) are drawn as "the k largest of one uniform key per element, keys of non-members at -1": a
    uniformly random k-subset without replacement -- the distribution of `x[randperm(len(x))[:k]]` -- with a flag per slot
    instead of a shorter tensor; flagged-off slots carry weight 0 in the sums, and every loss of the reference is a SUM
    divided by a constant (or a mean, restated as sum / count);
  * the frame-pair selectors stay on the device (`key = s * T + t` indexes tables packed once from the best-buddies file);
  * losses are accumulated as device scalars and read when `log_losses` formats them (every 100th iteration).

`make_trainer(base)` derives the class from the reference's own `DINOTracker` (loaded from the checkout at run time by
overlay/dino_tracker.py): configuration, paths, model / optimizer / scheduler set-up, checkpoints and logging stay the
reference's code; `train`, the loss terms and the batch sampler are this file's.  DTK_TRAINER=reference runs the inherited
loop instead (bit-identical random draws to the reference for equal seeds, the cycle-consistency point sets included: it
defaults DTK_CYC_SAMPLING to "reference"; the parity tests of tests/test_gpu_train.py).

The loss terms are pure functions of explicit selections (`*_terms`), so the CPU tests can feed them the selections the
reference's un-modified methods drew and compare the values (tests/test_trainer_vs_reference.py)."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F

EPS = 1e-8


# ---- loss arithmetic (pure functions) -----------------------------------------------------------------------------------
def contrastive_terms(a: torch.Tensor, b: torch.Tensor, fa: torch.Tensor, fb: torch.Tensor, temp: float):
    """dino_tracker.py:327-343 for P frame pairs at once.  a, b [P, B, C]: embeddings of B best-buddy pairs (a in the source
    frame, b in the target frame); fa, fb [P, n, C]: every cell of the source / target frame.  Returns the per-pair InfoNCE
    terms (source -> target, target -> source), each [P, B]: -log( exp(cos(a, b) / temp) / sum_n exp(cos(a, fb_n) / temp) )."""
    na, nb = a.norm(dim=2), b.norm(dim=2)
    nfa, nfb = fa.norm(dim=2), fb.norm(dim=2)
    bb = (a * b).sum(dim=2) / torch.clamp(na * nb, min=EPS)
    st = torch.bmm(a, fb.transpose(1, 2)) / torch.clamp(na[:, :, None] * nfb[:, None, :], min=EPS)
    ts = torch.bmm(b, fa.transpose(1, 2)) / torch.clamp(nb[:, :, None] * nfa[:, None, :], min=EPS)
    l_st = -torch.log(torch.exp(bb / temp) / torch.exp(st / temp).sum(dim=2))
    l_ts = -torch.log(torch.exp(bb / temp) / torch.exp(ts / temp).sum(dim=2))
    return l_st, l_ts


class _CosLse(torch.autograd.Function):
    """lse[q][i] = log sum_j exp(cos(a[q][i], cell j of frame fidx[q]) / temp) on hand-written kernels, both ways
    (csrc/train.hip: dtk_contrastive_forward / _backward): no [Q, B, n] tensor is handed to torch, the frames are indexed, not
    gathered."""

    @staticmethod
    def forward(ctx, a, fe, fidx, temp):
        from . import ops
        a, fe = a.contiguous(), fe.contiguous()
        fidx = fidx.to(torch.int32).contiguous()
        lse, saved = ops.contrastive_forward(fe, a, fidx, temp)
        ctx.save_for_backward(a, fe, fidx, lse, *saved)
        ctx.temp = temp
        return lse

    @staticmethod
    def backward(ctx, g):
        from . import ops
        a, fe, fidx, lse, nf, na, S = ctx.saved_tensors
        da, dfe = ops.contrastive_backward(fe, a, fidx, ctx.temp, (nf, na, S), lse, g.contiguous())
        return da, dfe, None, None


def contrastive_terms_indexed(a: torch.Tensor, b: torch.Tensor, fe: torch.Tensor, s_sel: torch.Tensor, t_sel: torch.Tensor,
                              temp: float):
    """contrastive_terms for frames given as INDICES into the frame embeddings fe [F, C, h, w] (device path): a, b [P, B, C];
    a's negatives are the cells of frame t_sel[p], b's those of frame s_sel[p].  Same values as
    contrastive_terms(a, b, frame_cells(fe, s_sel), frame_cells(fe, t_sel), temp) up to fp32 rounding:
    -log(exp(bb / temp) / sum_j exp(s_j / temp)) = lse - bb / temp."""
    P = a.shape[0]
    bb = (a * b).sum(dim=2) / torch.clamp(a.norm(dim=2) * b.norm(dim=2), min=EPS)
    lse = _CosLse.apply(torch.cat([a, b]), fe, torch.cat([t_sel, s_sel]), temp)
    return lse[:P] - bb / temp, lse[P:] - bb / temp


def cells_at(frame_embeddings: torch.Tensor, sel: torch.Tensor, cells: torch.Tensor) -> torch.Tensor:
    """[F, C, h, w], sel [P], cells [P, B] -> [P, B, C]: the embeddings of the given cells of frame sel[p] (no per-frame copy)."""
    return frame_embeddings.flatten(2)[sel[:, None], :, cells]


def fused_contrastive(fe: torch.Tensor) -> bool:
    return fe.is_cuda and fe.dtype == torch.float32 and fe.shape[1] % 4 == 0 and os.environ.get("DTK_CL_TERMS", "fused") == "fused"


def frame_cells(frame_embeddings: torch.Tensor, sel: torch.Tensor) -> torch.Tensor:
    """[F, C, h, w], sel [P] -> [P, n, C]: rearrange(frame_embeddings[sel_p], 'c h w -> (h w) c') for every p.
    On the device the selection is a product with the one-hot matrix of `sel` (exact: every output is one input times 1 plus
    zeros): its backward is the transposed product, where index_select's is a scatter kernel that serialises over the few
    distinct frames (1.7 ms per iteration at C = 1024)."""
    f, c = frame_embeddings.shape[:2]
    if frame_embeddings.is_cuda:
        onehot = torch.nn.functional.one_hot(sel, f).to(frame_embeddings.dtype)
        return (onehot @ frame_embeddings.flatten(1)).view(sel.shape[0], c, -1).transpose(1, 2)
    return frame_embeddings.flatten(2).index_select(0, sel).transpose(1, 2)


class _EmbReg(torch.autograd.Function):
    """Both regularisers in one kernel each way (csrc/train.hip: dtk_emb_reg_forward / _backward)."""

    @staticmethod
    def forward(ctx, refined, raw):
        from . import ops
        refined, raw = refined.contiguous(), raw.detach().contiguous()
        out, sums = ops.emb_reg_forward(refined, raw)
        ctx.save_for_backward(refined, raw, sums)
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import ops
        refined, raw, sums = ctx.saved_tensors
        return ops.emb_reg_backward(refined, raw, sums, gout.contiguous()), None


def emb_regularization_terms(refined: torch.Tensor, raw: torch.Tensor):
    """dino_tracker.py:128-139: (mean | |refined| / |raw| - 1 |, mean | cos(refined, raw) - 1 |) over frames and cells."""
    if refined.is_cuda and refined.dtype == torch.float32:
        out = _EmbReg.apply(refined, raw)
        return out[0], out[1]
    nr, nd = refined.norm(dim=1), raw.norm(dim=1)
    cos = (refined * raw).sum(dim=1) / (nr * nd)
    return (nr / nd - 1).abs().mean(), (cos - 1).abs().mean()


def huber(x: torch.Tensor, y: torch.Tensor, delta: float = 1 / 32) -> torch.Tensor:
    """torch.nn.HuberLoss(delta, reduction='none') (dino_tracker.py:30)."""
    return F.huber_loss(x, y, reduction="none", delta=delta)


def weighted_mean(values: torch.Tensor, keep: torch.Tensor) -> torch.Tensor:
    """values [M, K], keep [M] bool: the mean over the kept rows' elements (0 when none is kept)."""
    w = keep.to(values.dtype)
    return (values * w[:, None]).sum() / (w.sum().clamp(min=1) * values.shape[1])


def foreground_at(mask: torch.Tensor, coords: torch.Tensor, resw: int, resh: int) -> torch.Tensor:
    """models/utils.py:53-58 (filter_bb_foreground_pairs): is pixel position `coords` [N, 2] foreground in `mask` [H, W] --
    bilinear `grid_sample` of the mask at 2 (x / resw, y / resh) - 1 (torch's default align_corners=False, zero padding) > 0."""
    if coords.shape[0] == 0:
        return torch.zeros(0, dtype=torch.bool, device=coords.device)
    scale = torch.tensor([resw, resh], dtype=coords.dtype, device=coords.device)
    grid = 2 * (coords[None, None] / scale) - 1
    return F.grid_sample(mask[None, None].float(), grid, align_corners=False)[0, 0, 0] > 0


def pick_subsets(member_sets, counts, generator=None):
    """member_sets: list of bool [P, L] masks over the same L slots; counts: how many to draw from each.  One uniform key per
    slot; per set the `count` largest keys among its members.  Returns (slot indices [P, sum(counts)], ok [P, sum(counts)])."""
    from .train_ops import topk_rows
    P, L = member_sets[0].shape
    dev = member_sets[0].device
    keys = torch.rand(P, L, device=dev, generator=generator)
    neg = torch.full((), -1.0, device=dev)
    idx, ok = [], []
    for members, k in zip(member_sets, counts):
        top = topk_rows(torch.where(members, keys, neg), min(k, L))
        idx.append(top[1])
        ok.append(top[0] >= 0)
    return torch.cat(idx, dim=1), torch.cat(ok, dim=1)


# ---- best buddies of the DINO features, packed once ---------------------------------------------------------------------
class BestBuddyTable:
    """The dict of preprocessing_dino_bb (`{"s_t": {"source_coords", "target_coords", "cos_sims", "r"}}`, loaded by
    dino_tracker.py:72-73) as flat device arrays: slot[s * T + t] -> (offset, count) into coords / r / cos / foreground flag
    (the flag is filter_bb_foreground_pairs of dino_tracker.py:185-189, evaluated once per pair instead of every time the
    pair is drawn)."""

    def __init__(self, pairs: Dict[str, Dict[str, Optional[torch.Tensor]]], fg_masks: torch.Tensor, n_frames: int, resw: int,
                 resh: int, device):
        self.T = n_frames
        off, cnt, slot = [0], [], torch.full((n_frames * n_frames,), -1, dtype=torch.long)
        src, tgt, r, cos, fg = [], [], [], [], []
        by_source = {}
        for name, bb in pairs.items():
            s, t = (int(x) for x in name.split("_"))
            sc = bb.get("source_coords")
            if sc is None or sc.shape[0] == 0 or s >= n_frames or t >= n_frames:
                continue
            by_source.setdefault(s, []).append((t, bb))
        for s in sorted(by_source):
            entries = sorted(by_source[s], key=lambda e: e[0])
            coords_s = torch.cat([bb["source_coords"].to(device, torch.float32) for _, bb in entries])
            fg.append(foreground_at(fg_masks[s].to(device), coords_s, resw, resh))   # one evaluation per source frame
            for t, bb in entries:
                n = bb["source_coords"].shape[0]
                slot[s * n_frames + t] = len(cnt)
                cnt.append(n)
                off.append(off[-1] + n)
                src.append(bb["source_coords"].to(device, torch.float32))
                tgt.append(bb["target_coords"].to(device, torch.float32))
                r.append(bb["r"].to(device, torch.float32).reshape(-1))
                cos.append(bb["cos_sims"].to(device, torch.float32).reshape(-1))
        cat = lambda xs, shape: torch.cat(xs) if xs else torch.zeros(shape, device=device)
        # one trailing dummy entry so that empty slots index something finite
        self.src = torch.cat([cat(src, (0, 2)), torch.zeros(1, 2, device=device)])
        self.tgt = torch.cat([cat(tgt, (0, 2)), torch.zeros(1, 2, device=device)])
        self.r = torch.cat([cat(r, (0,)), torch.ones(1, device=device)])
        self.cos = torch.cat([cat(cos, (0,)), torch.zeros(1, device=device)])
        self.fg = torch.cat([cat(fg, (0,)).bool(), torch.zeros(1, dtype=torch.bool, device=device)])
        self.total = off[-1]
        self.max_count = max(cnt) if cnt else 1
        self.slot = slot.to(device)
        self.off = torch.tensor(off[:-1] + [self.total], dtype=torch.long, device=device)   # [slots + 1]; last = dummy
        self.cnt = torch.tensor(cnt + [0], dtype=torch.long, device=device)
        self.n_slots = len(cnt)

    def window(self, s_frames: torch.Tensor, t_frames: torch.Tensor):
        """Frame numbers [P] (device) -> (flat indices [P, max_count], in-range flag [P, max_count])."""
        slot = self.slot[s_frames.long() * self.T + t_frames.long()]
        slot = torch.where(slot < 0, torch.full_like(slot, self.n_slots), slot)
        ar = torch.arange(self.max_count, device=slot.device)
        inside = ar[None, :] < self.cnt[slot][:, None]
        pos = torch.where(inside, self.off[slot][:, None] + ar[None, :], torch.full_like(ar, self.total)[None, :])
        return pos, inside


# ---- mutual nearest neighbours of the refined features ------------------------------------------------------------------
@torch.no_grad()
def mutual_argmax_prepare(frame_embeddings: torch.Tensor, s_sel: torch.Tensor, t_sel: torch.Tensor, geom=None):
    """Everything of `mutual_argmax` in front of the search itself (device tensors only): the token-major copy, its 16-bit planes
    and the (source row, target frame) lists of both directions.  No host read -- the part a captured iteration keeps in its first
    graph (the search reads one counter back, so it runs between the two graphs)."""
    from . import ops
    Fn, C, h, w = frame_embeddings.shape
    n = h * w
    P = s_sel.shape[0]
    feat, norms = ops.pack_features(frame_embeddings.contiguous())
    method = ops.TRACK_MFMA if C % 32 == 0 else ops.TRACK_EXACT
    f16 = ops.make_feat_f16(geom, feat, norms) if method == ops.TRACK_MFMA else None
    ar = torch.arange(n, device=feat.device)
    rows = torch.cat([s_sel[:, None] * n + ar[None, :], t_sel[:, None] * n + ar[None, :]]).reshape(-1).to(torch.int32)
    tgt = torch.cat([t_sel[:, None].expand(P, n), s_sel[:, None].expand(P, n)]).reshape(-1).to(torch.int32)
    return dict(geom=geom, feat=feat, norms=norms, f16=f16, rows=rows.contiguous(), tgt=tgt.contiguous(), method=method,
                P=P, n=n, Fn=Fn, C=C)


@torch.no_grad()
def mutual_argmax_search(pre) -> torch.Tensor:
    """dtk_argmax_cells over the lists of mutual_argmax_prepare -> cell [2 P n] int32 (fp16 MFMA candidate search + exact fp32
    decision, the N4 kernel path; first maximum on ties; one counter read on the host)."""
    from . import ops
    cell, _ = ops.argmax_cells(pre["geom"], pre["feat"], pre["norms"], pre["f16"], pre["feat"].reshape(pre["Fn"] * pre["n"], pre["C"]),
                               pre["rows"], pre["tgt"], pre["method"])
    return cell


@torch.no_grad()
def mutual_argmax(frame_embeddings: torch.Tensor, s_sel: torch.Tensor, t_sel: torch.Tensor, geom=None):
    """dino_tracker.py:262-283: for P frame pairs (indices into frame_embeddings [F, C, h, w]) the arg-max over the TARGET
    cells of every source cell's cosine, and over the SOURCE cells of every target cell's -> (nn_st [P, n], nn_ts [P, n]).
    On the device: dtk_argmax_cells (mutual_argmax_prepare / _search) -- the n x n affinity matrix is never written.  On a host
    tensor: the affinity matrix and two arg-maxes."""
    if frame_embeddings.is_cuda:
        pre = mutual_argmax_prepare(frame_embeddings, s_sel, t_sel, geom)
        cell = mutual_argmax_search(pre).long().view(2, pre["P"], pre["n"])
        return cell[0], cell[1]
    ff = frame_embeddings.flatten(2).transpose(1, 2)
    sf, tf = ff[s_sel], ff[t_sel]
    aff = torch.bmm(sf, tf.transpose(1, 2)) / torch.clamp(sf.norm(dim=2)[:, :, None] * tf.norm(dim=2)[:, None, :], min=EPS)
    return aff.argmax(dim=2), aff.argmax(dim=1)


# ---- one iteration, eager or as two captured graphs ----------------------------------------------------------------------------
def graphs_enabled(device) -> bool:
    """DTK_TRAIN_GRAPH: "1" (default on a GPU) replays captured iterations, "0" runs every iteration eagerly."""
    return torch.device(device).type == "cuda" and os.environ.get("DTK_TRAIN_GRAPH", "1") != "0"


class GraphedIteration:
    """One iteration of the device-side trainer -- batch, forward, the loss terms, backward, the Adam step -- either eagerly or as TWO
    captured graphs (hipGraph through torch.cuda.CUDAGraph) around the one step that reads the device on the host:

        host    the batch's two frame sets (DinoTrackerSampler.draw_frame_sets), copied into the graphs' input tensors; the Adam
                scalars of this step (train_ops.GraphAdam.refresh: step counts and the scheduler's current learning rates)
        graph A the sampler's device part, the forward pass, the tracking and cycle terms, the random parts of both contrastive
                selections, the operands of the mutual-nearest-neighbour search            (trainer.iteration_front)
        eager   dtk_argmax_cells: its undecidable rows take the exact path, their count is read back (one synchronisation)
        graph B the contrastive terms on the search's result, the regularisers, the total, BACKWARD through both halves, the fused
                Adam step on device-resident scalars                                      (trainer.iteration_back)

    Why: the iteration is ~1 100 launches of which ~1 000 are small library kernels of the selections and loss arithmetic; issued one
    by one they keep the host as busy as the device (docs/TRAINING.md, round 4: per-iteration times jitter 0.03-0.14 s with host
    stalls).  A replay is two launches.

    A graph is specific to (frames in the batch, cycle term on, refiner term on): the union of the two drawn frame sets has 6-8
    frames at config/train.yaml's sizes, and the terms switch on at apply_*_after.  The FIRST iteration of a key runs eagerly (every
    lazily created table and scratch buffer of that shape then exists with real contents), the second is captured, later ones replay.
    All graphs share one memory pool: an iteration leaves nothing in it that the next one reads (parameters, optimizer state,
    BatchNorm statistics and the inputs live outside), so keys may replay in any order.  train_ops' persistent scratch keeps outgrown
    buffers alive once a graph exists (their addresses are baked into the launches).

    Random numbers: the device draws inside a capture go through torch's graph-safe Philox state (fresh offsets every replay); the
    host draws are the sampler's.  `values` of run(): the seven loss values (a fresh device vector)."""

    def __init__(self, trainer, model, optimizer, sampler, enabled=None):
        self.trainer, self.model, self.optimizer, self.sampler = trainer, model, optimizer, sampler
        self.device = next(model.parameters()).device
        self.enabled = graphs_enabled(self.device) if enabled is None else bool(enabled)
        self.entries, self.seen = {}, set()
        self.counts = {"eager": 0, "captured": 0, "replayed": 0}
        if self.enabled and not getattr(optimizer, "_dtk_fused", False):
            self.enabled = False    # torch's own Adam step (DTK_TRAIN_ADAM=torch, options the kernel lacks): its scalars live on the host
        if self.enabled and not getattr(sampler, "fg_valid_trajectories", torch.empty(0)).is_cuda:
            self.enabled = False    # keep_traj_in_cpu: the batch is assembled from host tensors -- nothing to capture
        if self.enabled:
            from .train_ops import GraphAdam
            self.pool = torch.cuda.graph_pool_handle()
            self.stream = torch.cuda.Stream(device=self.device)
            self.adam = GraphAdam(optimizer, self.device)

    def invalidate(self):
        """Drop every captured graph (a tensor whose address they read was replaced, e.g. DinoTrackerSampler.load_next_batch)."""
        self.entries, self.seen = {}, set()

    def key(self, union, i):
        cfg = self.trainer.config
        return (len(union), i >= cfg.get("apply_cyc_after", 0), i >= cfg.get("apply_cl_ref_after", 0))

    # -- eager ------------------------------------------------------------------------------------------------------------
    def run_eager(self, host, union, i):
        from .dataset import stage_to_device
        tr, model = self.trainer, self.model
        self.optimizer.zero_grad(set_to_none=True)
        if self.entries:    # (the last capture's autograd graph and its AccumulateGrad nodes belong to the capture stream: see capture())
            model.frame_embeddings = model.raw_embeddings = model.residual_embeddings = None
        staged = stage_to_device(host, self.device)
        frames_set_t = stage_to_device(torch.tensor(union, dtype=torch.int32), self.device)
        inputs, labels, valid = tr._batch(self.sampler.batch_from_frame_sets(staged, frames_set_t, union))
        loss, values = tr.iteration_losses(model, inputs, labels, valid, i)
        loss.backward()
        self.optimizer.step()
        self.counts["eager"] += 1
        return values

    # -- capture ----------------------------------------------------------------------------------------------------------
    def capture(self, key, host, union, i):
        from . import train_ops
        tr, model = self.trainer, self.model
        train_ops.RETAIN_REPLACED_WORKSPACES = True
        e = {"staged": torch.zeros(host.shape, dtype=torch.long, device=self.device),
             "frames_set_t": torch.zeros(len(union), dtype=torch.int32, device=self.device)}
        self.optimizer.zero_grad(set_to_none=True)
        # The previous (eager) iteration's autograd graph is still alive through the tensors the model keeps for the loss terms; it
        # holds the parameters' AccumulateGrad nodes, which belong to the stream they were created on -- gradients accumulated there
        # would leave the capturing stream.  Without that graph the capture creates its own nodes.
        model.frame_embeddings = model.raw_embeddings = model.residual_embeddings = None
        train_ops._PACKED.clear()
        torch.cuda.synchronize(self.device)
        gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gA, pool=self.pool, stream=self.stream):
            inputs, labels, valid = tr._batch(self.sampler.batch_from_frame_sets(e["staged"], e["frames_set_t"], None))
            st = tr.iteration_front(model, inputs, labels, valid, i)
        pre = st["prepared"][2] if st["prepared"] is not None else None
        e["found"] = torch.zeros(2 * pre["P"] * pre["n"], dtype=torch.int32, device=self.device) if pre is not None else None
        with torch.cuda.graph(gB, pool=self.pool, stream=self.stream):
            loss, values = tr.iteration_back(model, st, e["found"], i)
            loss.backward()
            e["params"], e["grads"] = self.adam.launch()
        e.update(gA=gA, gB=gB, st=st, values=values)
        self.entries[key] = e
        self.counts["captured"] += 1
        return e

    # -- replay -----------------------------------------------------------------------------------------------------------
    def replay(self, e, host, union):
        from .dataset import stage_to_device
        e["staged"].copy_(stage_to_device(host, self.device))
        e["frames_set_t"].copy_(stage_to_device(torch.tensor(union, dtype=torch.int32), self.device))
        self.adam.refresh(e["params"])
        e["gA"].replay()
        if e["found"] is not None:
            e["found"].copy_(self.trainer.refined_bb_search(self.model, e["st"]["prepared"]))
        e["gB"].replay()
        self.optimizer._opt_called = True
        self.counts["replayed"] += 1
        return e["values"].clone()

    def run(self, i):
        host, union = self.sampler.draw_frame_sets()
        if not self.enabled:
            return self.run_eager(host, union, i)
        key = self.key(union, i)
        e = self.entries.get(key)
        if e is None:
            if key not in self.seen:
                self.seen.add(key)
                return self.run_eager(host, union, i)
            e = self.capture(key, host, union, i)       # (a capture records, it does not execute: the replay below is the iteration)
        return self.replay(e, host, union)


# ---- the trainer ----------------------------------------------------------------------------------------------------------
def make_trainer(base):
    """`base`: the reference's DINOTracker class.  Returns the subclass with the device-side iteration."""

    class DINOTracker(base):
        LOSS_NAMES = ("total", "of", "cl_dino_bb", "cl_refiner", "emb_norm_reg", "angle_reg", "cyc")

        # -- set-up ------------------------------------------------------------------------------------------------------
        def prepare_tables(self, model):
            dev = model.video.device
            t, _, h, w = model.video.shape
            self._bb_table = BestBuddyTable(self.dino_bb_pairs, self.fg_masks, t, w, h, dev)
            grid = self.feature_grid(model, dev)
            masks = self.fg_masks.to(dev)
            self._cell_fg = torch.stack([foreground_at(masks[f], grid, w, h) for f in range(t)])      # [T, n]

        @staticmethod
        def feature_grid(model, dev):
            """models/utils.py:87-95: pixel centres (x, y) of the token grid, row-major."""
            _, _, h, w = model.video.shape
            half = model.dino_patch_size // 2
            x = torch.arange(half, w - half + 1, step=model.stride, device=dev).float()
            y = torch.arange(half, h - half + 1, step=model.stride, device=dev).float()
            yy, xx = torch.meshgrid(y, x, indexing="ij")
            return torch.stack([xx.reshape(-1), yy.reshape(-1)], dim=-1)

        # -- batch -------------------------------------------------------------------------------------------------------
        @staticmethod
        def _batch(sample):
            labels = sample["t2_points_normalized"][:, :-1]
            inputs = (sample["t1_points"], sample["source_frame_indices"], sample["target_frame_indices"], sample["frames_set_t"])
            return inputs, labels, sample["valid"]

        def get_inputs_and_labels_device(self, sampler):
            return self._batch(sampler.forward_device())

        # -- loss terms --------------------------------------------------------------------------------------------------
        def split_counts(self, per_pair, fg_ratio):
            n_fg = int(per_pair * fg_ratio)
            return n_fg, per_pair - n_fg

        def dino_bb_selection(self, frames_set_t):
            """Random part of dino_tracker.py:159-212: cl_n_frames (source != target) index pairs into the batch's frame set
            and, per pair, up to cl_points_per_pair of the pair's DINO best buddies, cl_fg_points_ratio of them foreground."""
            n = frames_set_t.shape[0]
            dev = frames_set_t.device
            P = self.config["cl_n_frames"]
            s_sel = torch.randint(n, (P,), device=dev)
            t_sel = (s_sel + 1 + torch.randint(max(n - 1, 1), (P,), device=dev)) % n      # uniform over t != s (:163-164)
            pos, inside = self._bb_table.window(frames_set_t[s_sel], frames_set_t[t_sel])
            fg = self._bb_table.fg[pos] & inside
            n_fg, n_bg = self.split_counts(self.config["cl_points_per_pair"], self.config["cl_fg_points_ratio"])
            col, ok = pick_subsets([fg, inside & ~fg], [n_fg, n_bg])
            return s_sel, t_sel, pos.gather(1, col), ok

        def dino_bb_operands(self, model, s_sel, t_sel, picks, ok):
            """Embeddings of the selected DINO best buddies in their source / target frames (one bilinear-sampling call for
            both: one gradient edge into the frame embeddings) and their weights (dino_tracker.py:220-231) -> a, b [P, B, C], w."""
            tb = self._bb_table
            fe = model.frame_embeddings
            P, B = picks.shape
            col = lambda sel: sel[:, None].expand(P, B).to(torch.float32)[:, :, None]
            pts = torch.cat([torch.cat([tb.src[picks], col(s_sel)], dim=2), torch.cat([tb.tgt[picks], col(t_sel)], dim=2)])
            emb = model.sample_embeddings(fe, model.normalize_points_for_sampling(pts.reshape(2 * P * B, 3))).reshape(2, P, B, -1)
            w_amb = torch.sigmoid(self.config["bb_amb_sig_a"] * (1 - tb.r[picks]) + self.config["bb_amb_sig_b"])
            w_cos = torch.clamp(2 * tb.cos[picks] ** 3, 0)
            return emb[0], emb[1], w_amb * w_cos * ok.to(w_amb.dtype)

        def dino_bb_terms(self, model, s_sel, t_sel, picks, ok):
            """Deterministic part of dino_tracker.py:159-240 for explicit selections: s_sel / t_sel [P] index the batch's
            frame set, picks [P, B] index the flat best-buddy table, ok [P, B] switches slots off."""
            fe = model.frame_embeddings
            a, b, w = self.dino_bb_operands(model, s_sel, t_sel, picks, ok)
            if fused_contrastive(fe):
                l_st, l_ts = contrastive_terms_indexed(a, b, fe, s_sel, t_sel, self.config["cl_temp"])
            else:
                l_st, l_ts = contrastive_terms(a, b, frame_cells(fe, s_sel), frame_cells(fe, t_sel), self.config["cl_temp"])
            div = self.config["cl_div_dino_bb"]
            return ((l_st * w / div).sum() + (l_ts * w / div).sum()) / 2

        def refined_bb_prepare(self, model, frames_set_t):
            """Random part of dino_tracker.py:245-305, first half: cl_n_frames index pairs (source == target allowed, as there)
            and the operands of the mutual-nearest-neighbour search over the current refined features (no host read)."""
            fe = model.frame_embeddings
            n = frames_set_t.shape[0]
            dev = fe.device
            P = self.config["cl_n_frames"]
            s_sel = torch.randint(n, (P,), device=dev)
            t_sel = torch.randint(n, (P,), device=dev)
            pre = None
            if fe.is_cuda:
                pre = mutual_argmax_prepare(fe.detach(), s_sel, t_sel, model.tracker_head.geom(fe.shape[0], fe.shape[1]))
            return s_sel, t_sel, pre

        def refined_bb_search(self, model, prepared):
            """The search between the two halves -> (nn_st [P, n], nn_ts [P, n]) or, on the device, the flat int32 cell list."""
            s_sel, t_sel, pre = prepared
            if pre is not None:
                return mutual_argmax_search(pre)
            return mutual_argmax(model.frame_embeddings.detach(), s_sel, t_sel)

        def refined_bb_finish(self, frames_set_t, prepared, found):
            """Second half: the mutual pairs and up to cl_points_per_pair of them per frame pair, split foreground / background
            by the source frame's mask at the cell centres."""
            s_sel, t_sel, pre = prepared
            if pre is not None:
                cell = found.long().view(2, pre["P"], pre["n"])
                nn_st, nn_ts = cell[0], cell[1]
            else:
                nn_st, nn_ts = found
            cells = torch.arange(nn_st.shape[1], device=nn_st.device)
            mutual = nn_ts.gather(1, nn_st) == cells[None, :]
            fg = self._cell_fg[frames_set_t[s_sel].long()]
            n_fg, n_bg = self.split_counts(self.config["cl_points_per_pair"], self.config["cl_fg_points_ratio"])
            src_cells, ok = pick_subsets([mutual & fg, mutual & ~fg], [n_fg, n_bg])
            return s_sel, t_sel, src_cells, nn_st.gather(1, src_cells), ok

        def refined_bb_selection(self, model, frames_set_t):
            """Random part of dino_tracker.py:245-305: cl_n_frames index pairs (source == target allowed, as there), the
            mutual nearest neighbours of the current refined features, and up to cl_points_per_pair of them per pair split
            foreground / background by the source frame's mask at the cell centres."""
            prepared = self.refined_bb_prepare(model, frames_set_t)
            return self.refined_bb_finish(frames_set_t, prepared, self.refined_bb_search(model, prepared))

        def refined_bb_terms(self, model, s_sel, t_sel, src_cells, tgt_cells, ok):
            """Deterministic part of dino_tracker.py:245-325 for explicit selections (cells of the token grid)."""
            fe = model.frame_embeddings
            if fused_contrastive(fe):
                a, b = cells_at(fe, s_sel, src_cells), cells_at(fe, t_sel, tgt_cells)
                l_st, l_ts = contrastive_terms_indexed(a, b, fe, s_sel, t_sel, self.config["cl_temp"])
            else:
                sf, tf = frame_cells(fe, s_sel), frame_cells(fe, t_sel)
                C = sf.shape[2]
                a = sf.gather(1, src_cells[:, :, None].expand(-1, -1, C))
                b = tf.gather(1, tgt_cells[:, :, None].expand(-1, -1, C))
                l_st, l_ts = contrastive_terms(a, b, sf, tf, self.config["cl_temp"])
            with torch.no_grad():
                aff = (a * b).sum(dim=2) / torch.clamp(a.norm(dim=2) * b.norm(dim=2), min=EPS)
                w = torch.clamp(2 * aff ** 3, 0) * ok.to(aff.dtype)
            return ((l_st * w).sum() + (l_ts * w).sum()) / (2 * self.config["cl_div_ref_bb"])

        def contrastive_losses(self, model, bb_sel, ref_sel):
            """Both contrastive losses in ONE evaluation: the cl_n_frames pairs of the DINO best-buddy loss and those of the
            refined best-buddy loss are the same arithmetic (dino_tracker.py:327-343) with different operands and weights, so
            they run as one batch of 2 cl_n_frames pairs -- half the launches, and the frame embeddings receive one gathered
            gradient (plus the sampling's) instead of four full-size ones.  Returns (cl_dino_bb, cl_refiner); equal to
            dino_bb_terms / refined_bb_terms up to summation order."""
            fe = model.frame_embeddings
            s1, t1, picks, ok1 = bb_sel
            s2, t2, src_cells, tgt_cells, ok2 = ref_sel
            P = s1.shape[0]
            fused = fused_contrastive(fe)
            a1, b1, w1 = self.dino_bb_operands(model, s1, t1, picks, ok1)
            if fused:   # the frames stay where they are: the kernels index them (csrc/train.hip, dtk_contrastive_*)
                a2, b2 = cells_at(fe, s2, src_cells), cells_at(fe, t2, tgt_cells)
            else:
                cells = frame_cells(fe, torch.cat([s1, s2, t1, t2]))          # [4 P, n, C]: one gather of the frames' cells
                sf, tf = cells[:2 * P], cells[2 * P:]
                C = sf.shape[2]
                a2 = sf[P:].gather(1, src_cells[:, :, None].expand(-1, -1, C))
                b2 = tf[P:].gather(1, tgt_cells[:, :, None].expand(-1, -1, C))
            # the two selections may differ in width (a pair has fewer DINO best buddies than cl_points_per_pair): the narrower
            # one is padded with copies of its first slot at weight 0
            B = max(a1.shape[1], a2.shape[1])

            def pad(x, value_from_first=True):
                extra = B - x.shape[1]
                if extra == 0:
                    return x
                fill = x[:, :1].expand(-1, extra, *x.shape[2:]) if value_from_first else x.new_zeros(x.shape[0], extra, *x.shape[2:])
                return torch.cat([x, fill], dim=1)

            a1, b1, a2, b2 = pad(a1), pad(b1), pad(a2), pad(b2)
            w1 = pad(w1, value_from_first=False)
            ok2 = pad(ok2, value_from_first=False)
            if fused:
                l_st, l_ts = contrastive_terms_indexed(torch.cat([a1, a2]), torch.cat([b1, b2]), fe, torch.cat([s1, s2]),
                                                       torch.cat([t1, t2]), self.config["cl_temp"])
            else:
                l_st, l_ts = contrastive_terms(torch.cat([a1, a2]), torch.cat([b1, b2]), sf, tf, self.config["cl_temp"])
            with torch.no_grad():
                aff = (a2 * b2).sum(dim=2) / torch.clamp(a2.norm(dim=2) * b2.norm(dim=2), min=EPS)
                w2 = torch.clamp(2 * aff ** 3, 0) * ok2.to(aff.dtype)
            div = self.config["cl_div_dino_bb"]
            cl_bb = ((l_st[:P] * w1 / div).sum() + (l_ts[:P] * w1 / div).sum()) / 2
            cl_ref = ((l_st[P:] * w2).sum() + (l_ts[P:] * w2).sum()) / (2 * self.config["cl_div_ref_bb"])
            return cl_bb, cl_ref

        def cycle_terms(self, model, frames_set_t):
            """dino_tracker.py:345-352 over the static-shape cycle batch (Tracker.get_cycle_consistency_terms)."""
            c = model.get_cycle_consistency_terms(frames_set_t, self.fg_masks)
            wgt = self.config["cyc_gamma"] ** c["cycle_consistency_dists"]
            st = wgt[:, None] * huber(c["source_target_coords"], c["target_coords"][:, :2])
            ts = wgt[:, None] * huber(c["target_source_coords"], c["source_coords"][:, :2])
            return (weighted_mean(st, c["keep"]) + weighted_mean(ts, c["keep"])) / 2

        def iteration_front(self, model, inputs, labels, valid, i):
            """The iteration up to the mutual-nearest-neighbour search (dino_tracker.py:407-417 and the draws of :159-212 / the
            first half of :245-305): forward pass, tracking term, cycle term, both selections' random parts.  No host read."""
            cfg = self.config
            frames_set_t = inputs[-1]
            coords = model(inputs)
            zero = coords.new_zeros(())
            st = dict(frames_set_t=frames_set_t, tracking=weighted_mean(huber(coords, labels), valid), cyc=zero, prepared=None)
            if i >= cfg.get("apply_cyc_after", 0):
                st["cyc"] = self.cycle_terms(model, frames_set_t)
            st["bb_sel"] = self.dino_bb_selection(frames_set_t)
            if i >= cfg.get("apply_cl_ref_after", 0):
                st["prepared"] = self.refined_bb_prepare(model, frames_set_t)
            return st

        def iteration_back(self, model, st, found, i):
            """The rest (dino_tracker.py:418-426): contrastive terms on the search's result `found`, regularisers, the total ->
            (loss, the seven loss values as a device vector in LOSS_NAMES order)."""
            cfg = self.config
            tracking, cyc = st["tracking"], st["cyc"]
            loss = tracking
            cl_ref = tracking.new_zeros(())
            if i >= cfg.get("apply_cyc_after", 0):
                loss = loss + cfg["lambda_cyc"] * cyc
            if st["prepared"] is not None:
                ref_sel = self.refined_bb_finish(st["frames_set_t"], st["prepared"], found)
                cl_bb, cl_ref = self.contrastive_losses(model, st["bb_sel"], ref_sel)
                loss = loss + cfg["lambda_cl_ref_bb"] * cl_ref
            else:
                cl_bb = self.dino_bb_terms(model, *st["bb_sel"])
            norm_reg, angle_reg = emb_regularization_terms(model.frame_embeddings, model.raw_embeddings)
            loss = loss + cfg["lambda_cl_dino_bb"] * cl_bb + cfg["lambda_emb_norm"] * norm_reg + cfg["lambda_angle"] * angle_reg
            return loss, torch.stack([loss.detach(), tracking.detach(), cl_bb.detach(), cl_ref.detach(), norm_reg.detach(),
                                      angle_reg.detach(), cyc.detach()])

        def iteration_losses(self, model, inputs, labels, valid, i):
            """The seven loss values of one iteration as a device vector in LOSS_NAMES order (dino_tracker.py:407-426)."""
            st = self.iteration_front(model, inputs, labels, valid, i)
            found = self.refined_bb_search(model, st["prepared"]) if st["prepared"] is not None else None
            return self.iteration_back(model, st, found, i)

        # -- the loop (dino_tracker.py:392-448) ---------------------------------------------------------------------------
        def train(self):
            if os.environ.get("DTK_TRAINER", "device") == "reference":
                # the inherited loop promises the reference's sequence of random draws: that includes the cycle-consistency
                # point sets, which Tracker draws on the device unless told otherwise
                os.environ.setdefault("DTK_CYC_SAMPLING", "reference")
                return super().train()
            from tqdm import tqdm
            self.load_fg_masks()
            total_iterations = self.config["total_iterations"]
            checkpoint_interval = self.config["checkpoint_interval"]
            sampler_batch_iterations = self.config.get("sampler_batch_iterations", 100_000)
            self.load_dino_best_buddies()
            train_sampler = self.get_sampler()
            model, optimizer, scheduler = self.train_setup()
            from .train_ops import install_fused_adam
            install_fused_adam(optimizer)                     # the reference's own torch.optim.Adam object, its step on ONE kernel
            self.set_model_train(model)
            self.init_losses()
            self.prepare_tables(model)
            step = GraphedIteration(self, model, optimizer, train_sampler)
            for i in tqdm(range(self.init_iter, total_iterations)):
                values = step.run(i)
                scheduler.step()
                self.update_losses(*values.unbind())          # device scalars: read when log_losses formats them
                if i % 100 == 0:
                    self.log_losses(i, log_interval=100)
                if i == total_iterations - 1 or i % checkpoint_interval == 0:
                    model.save_weights(i)
                if i % sampler_batch_iterations == 0 and i > 0:
                    print("Loading next batch", flush=True)
                    train_sampler.load_next_batch()
                    step.invalidate()                         # (the trajectory tables are new tensors: graphs read the old ones)
            model.save_weights(total_iterations)

    DINOTracker.__qualname__ = "DINOTracker"
    return DINOTracker
