"""N1, CPU: the loss terms of dino_tracker_amd/trainer.py against the UN-MODIFIED methods of the reference's trainer
(dino_tracker.py:128-352) -- the reference methods run with torch's random functions recorded, the recorded draws are
turned into the explicit selections the device-side terms take, and values and gradients must agree.  Plus the properties
of the key-based subset sampling and of the device-side batch sampler.  Skipped where the reference is absent."""
import os
import sys
import types

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="reference not present")

from dino_tracker_amd import trainer as T  # noqa: E402
from dino_tracker_amd.dataset import DinoTrackerSampler, RangeNormalizer  # noqa: E402

CONFIG = {
    "cl_n_frames": 4, "cl_points_per_pair": 40, "cl_fg_points_ratio": 0.7, "cl_temp": 0.1, "cl_div_dino_bb": 700,
    "cl_div_ref_bb": 900, "bb_amb_sig_a": 27, "bb_amb_sig_b": -5.7, "dino_patch_size": 14, "cyc_gamma": 0.8,
    "lambda_cyc": 0.5, "lambda_cl_ref_bb": 5e-5, "lambda_cl_dino_bb": 2.5e-4, "lambda_emb_norm": 1e-4, "lambda_angle": 1e-4,
}
H, W, FRAMES, C = 126, 182, 6, 16


@pytest.fixture()
def world(monkeypatch):
    """The reference trainer module, an instance of its class and of the derived class (no __init__: attributes set by
    hand), a synthetic scene: masks, best buddies of every frame pair, a stand-in model holding the batch's embeddings."""
    ref = ref_harness.load()
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    import models.utils as mu
    monkeypatch.setattr(mu.get_vit_feature_coords_from_mask, "__defaults__", (7, 14, "cpu"))
    import dino_tracker as ref_dt
    assert ref_harness.REFERENCE_ROOT in ref_dt.__file__
    g = torch.Generator().manual_seed(0)
    h, w = (H - 14) // 7 + 1, (W - 14) // 7 + 1
    fg_masks = torch.zeros(FRAMES, H, W)
    for f in range(FRAMES):
        fg_masks[f, 20 + 3 * f:90, 30:120 + 5 * f] = 1.0
    pairs = {}
    for s in range(FRAMES):
        for t in range(FRAMES):
            if s == t:
                continue
            n = int(torch.randint(0, 90, (1,), generator=g)) if (s + t) % 5 else 0
            if n == 0:
                pairs[f"{s}_{t}"] = {"source_coords": None if s % 2 else torch.zeros(0, 2), "target_coords": None,
                                     "cos_sims": None, "r": None}
                continue
            xy = lambda: torch.stack([torch.rand(n, generator=g) * (W - 15) + 7, torch.rand(n, generator=g) * (H - 15) + 7], 1)
            pairs[f"{s}_{t}"] = {"source_coords": xy(), "target_coords": xy(), "cos_sims": torch.rand(n, generator=g) * 1.2 - 0.2,
                                 "r": torch.rand(n, generator=g)}
    frames_set_t = torch.tensor([0, 2, 3, 5], dtype=torch.int32)
    emb = torch.randn(4, C, h, w, generator=g) + 0.3
    raw = emb + 0.1 * torch.randn(4, C, h, w, generator=g)

    def model_like(tracker_cls, e):
        m = types.SimpleNamespace()
        m.video = torch.zeros(FRAMES, 3, H, W)
        m.dino_patch_size, m.stride, m.device = 14, 7, "cpu"
        m.frame_embeddings = e.clone().requires_grad_(True)
        m.raw_embeddings = raw
        m._refined = m._dino = None
        m.normalize_points_for_sampling = types.MethodType(tracker_cls.normalize_points_for_sampling, m)
        m.sample_embeddings = types.MethodType(tracker_cls.sample_embeddings, m)
        return m

    from dino_tracker_amd.tracker import Tracker as OurTracker
    theirs = object.__new__(ref_dt.DINOTracker)
    ours = object.__new__(T.make_trainer(ref_dt.DINOTracker))
    for o in (theirs, ours):
        o.config, o.fg_masks, o.dino_bb_pairs = dict(CONFIG), fg_masks, pairs
        o.of_loss_fn = torch.nn.HuberLoss(delta=1 / 32, reduction="none")
    m_ours = model_like(OurTracker, emb)
    ours.prepare_tables(m_ours)
    return types.SimpleNamespace(ref=ref, ref_dt=ref_dt, theirs=theirs, ours=ours, m_theirs=model_like(ref.tracker.Tracker, emb),
                                 m_ours=m_ours, frames_set_t=frames_set_t, pairs=pairs, fg_masks=fg_masks, h=h, w=w)


class Recorder:
    """Records what torch.randint / torch.randperm return while the reference's methods run."""

    def __init__(self, monkeypatch):
        self.randint, self.randperm = [], []
        ri, rp = torch.randint, torch.randperm

        def randint(*a, **k):
            out = ri(*a, **k)
            self.randint.append(out.clone())
            return out

        def randperm(*a, **k):
            out = rp(*a, **k)
            self.randperm.append(out.clone())
            return out

        monkeypatch.setattr(torch, "randint", randint)
        monkeypatch.setattr(torch, "randperm", randperm)


def _pad(rows, width):
    idx = torch.zeros(len(rows), width, dtype=torch.long)
    ok = torch.zeros(len(rows), width, dtype=torch.bool)
    for i, r in enumerate(rows):
        idx[i, :len(r)] = r
        ok[i, :len(r)] = True
    return idx, ok


def _close(a, b, tol=2e-5):
    assert abs(float(a) - float(b)) <= tol * max(abs(float(b)), 1e-3), (float(a), float(b))


def test_dino_bb_contrastive_loss_matches_reference(world, monkeypatch):
    wd = world
    torch.manual_seed(3)
    rec = Recorder(monkeypatch)
    want = wd.theirs.get_dino_bb_contrastive_loss(wd.m_theirs, wd.frames_set_t)
    want.backward()
    s_sel, t_sel = rec.randint[0], rec.randint[-1]          # the target selectors are re-drawn until none equals its source
    n_fg, n_bg = wd.ours.split_counts(CONFIG["cl_points_per_pair"], CONFIG["cl_fg_points_ratio"])
    tb = wd.ours._bb_table
    rows, perms = [], iter(rec.randperm)
    used_pairs = 0
    for s_i, t_i in zip(s_sel.tolist(), t_sel.tolist()):
        s_f, t_f = int(wd.frames_set_t[s_i]), int(wd.frames_set_t[t_i])
        bb = wd.pairs[f"{s_f}_{t_f}"]
        if bb["source_coords"] is None or bb["source_coords"].shape[0] == 0:
            rows.append(torch.zeros(0, dtype=torch.long))
            continue
        used_pairs += 1
        _, _, fg = wd.ref.models_utils.filter_bb_foreground_pairs(bb["source_coords"], bb["target_coords"], wd.fg_masks[s_f],
                                                                  resw=W, resh=H)
        slot = int(tb.slot[s_f * FRAMES + t_f])
        assert slot >= 0 and int(tb.cnt[slot]) == fg.shape[0]
        base = int(tb.off[slot])
        assert torch.equal(tb.fg[base:base + fg.shape[0]], fg), "foreground flags of the packed table"
        ar = torch.arange(fg.shape[0])
        rows.append(base + torch.cat([ar[fg][next(perms)[:n_fg]], ar[~fg][next(perms)[:n_bg]]]))
    assert used_pairs >= 2, "the scene should exercise populated pairs"
    picks, ok = _pad(rows, n_fg + n_bg)
    got = wd.ours.dino_bb_terms(wd.m_ours, s_sel, t_sel, picks, ok)
    got.backward()
    _close(got, want)
    g_want, g_got = wd.m_theirs.frame_embeddings.grad, wd.m_ours.frame_embeddings.grad
    assert (g_got - g_want).abs().max() <= 2e-5 * g_want.abs().max()


def test_refined_bb_contrastive_loss_matches_reference(world, monkeypatch):
    wd = world
    torch.manual_seed(4)
    rec = Recorder(monkeypatch)
    want = wd.theirs.get_refiner_contrastive_loss(wd.m_theirs, wd.frames_set_t)
    want.backward()
    s_sel, t_sel = rec.randint[0], rec.randint[1]
    n_fg, n_bg = wd.ours.split_counts(CONFIG["cl_points_per_pair"], CONFIG["cl_fg_points_ratio"])
    nn_st, nn_ts = T.mutual_argmax(wd.m_ours.frame_embeddings.detach(), s_sel, t_sel)
    cells = torch.arange(nn_st.shape[1])
    perms = iter(rec.randperm)
    src_rows, tgt_rows = [], []
    for p in range(s_sel.shape[0]):
        mutual = nn_ts[p][nn_st[p]] == cells
        assert int(mutual.sum()) > 0
        fg = wd.ours._cell_fg[int(wd.frames_set_t[s_sel[p]])]
        bb_cells = cells[mutual]
        bb_fg = fg[mutual]
        pf, pb = next(perms), next(perms)
        src = torch.cat([bb_cells[bb_fg][pf[:n_fg]], bb_cells[~bb_fg][pb[:n_bg]]])
        src_rows.append(src)
        tgt_rows.append(nn_st[p][src])
    src_cells, ok = _pad(src_rows, n_fg + n_bg)
    tgt_cells, _ = _pad(tgt_rows, n_fg + n_bg)
    got = wd.ours.refined_bb_terms(wd.m_ours, s_sel, t_sel, src_cells, tgt_cells, ok)
    got.backward()
    _close(got, want)
    g_want, g_got = wd.m_theirs.frame_embeddings.grad, wd.m_ours.frame_embeddings.grad
    assert (g_got - g_want).abs().max() <= 2e-5 * g_want.abs().max()


def test_random_selections_are_members_and_feed_the_terms(world):
    """The random halves: selections only contain members of the sets they were drawn from, without repetition, flagged
    slots are exactly the shortfall; and the terms run on them."""
    wd = world
    torch.manual_seed(5)
    tb = wd.ours._bb_table
    for _ in range(5):
        s_sel, t_sel, picks, ok = wd.ours.dino_bb_selection(wd.frames_set_t)
        assert bool((s_sel != t_sel).all())
        n_fg, n_bg = wd.ours.split_counts(CONFIG["cl_points_per_pair"], CONFIG["cl_fg_points_ratio"])
        for p in range(s_sel.shape[0]):
            s_f, t_f = int(wd.frames_set_t[s_sel[p]]), int(wd.frames_set_t[t_sel[p]])
            slot = int(tb.slot[s_f * FRAMES + t_f])
            cnt = int(tb.cnt[slot]) if slot >= 0 else 0
            base = int(tb.off[slot]) if slot >= 0 else 0
            sel = picks[p][ok[p]]
            assert sel.unique().numel() == sel.numel() and bool(((sel >= base) & (sel < base + cnt)).all())
            fg_total = int(tb.fg[base:base + cnt].sum())
            assert int(ok[p, :n_fg].sum()) == min(n_fg, fg_total) and bool(tb.fg[picks[p, :n_fg][ok[p, :n_fg]]].all())
            assert int(ok[p, n_fg:].sum()) == min(n_bg, cnt - fg_total) and not bool(tb.fg[picks[p, n_fg:][ok[p, n_fg:]]].any())
        assert torch.isfinite(wd.ours.dino_bb_terms(wd.m_ours, s_sel, t_sel, picks, ok))
        sel = wd.ours.refined_bb_selection(wd.m_ours, wd.frames_set_t)
        assert torch.isfinite(wd.ours.refined_bb_terms(wd.m_ours, *sel))


def test_joint_contrastive_evaluation_equals_the_separate_terms(world):
    """trainer.contrastive_losses (both losses as one batch of pairs) against dino_bb_terms + refined_bb_terms on the same
    selections: values and the gradient with respect to the frame embeddings."""
    wd = world
    torch.manual_seed(11)
    bb_sel = wd.ours.dino_bb_selection(wd.frames_set_t)
    ref_sel = wd.ours.refined_bb_selection(wd.m_ours, wd.frames_set_t)
    fe = wd.m_ours.frame_embeddings
    sep_bb, sep_ref = wd.ours.dino_bb_terms(wd.m_ours, *bb_sel), wd.ours.refined_bb_terms(wd.m_ours, *ref_sel)
    (sep_bb + 3 * sep_ref).backward()
    g_sep, fe.grad = fe.grad.clone(), None
    j_bb, j_ref = wd.ours.contrastive_losses(wd.m_ours, bb_sel, ref_sel)
    (j_bb + 3 * j_ref).backward()
    _close(j_bb, sep_bb, 1e-6)
    _close(j_ref, sep_ref, 1e-6)
    assert (fe.grad - g_sep).abs().max() <= 1e-5 * g_sep.abs().max()
    # selections of different widths (fewer best buddies per pair than cl_points_per_pair)
    narrow = (bb_sel[0], bb_sel[1], bb_sel[2][:, :17], bb_sel[3][:, :17])
    sep_n = wd.ours.dino_bb_terms(wd.m_ours, *narrow)
    j_n, j_ref2 = wd.ours.contrastive_losses(wd.m_ours, narrow, ref_sel)
    _close(j_n, sep_n, 1e-6)
    _close(j_ref2, sep_ref, 1e-6)
    wide_ref = tuple(x[:, :9] if x.dim() == 2 else x for x in ref_sel)
    j_bb3, j_ref3 = wd.ours.contrastive_losses(wd.m_ours, bb_sel, wide_ref)
    _close(j_bb3, sep_bb, 1e-6)
    _close(j_ref3, wd.ours.refined_bb_terms(wd.m_ours, *wide_ref), 1e-6)


def test_regularisation_tracking_and_cycle_terms_match_reference(world):
    wd = world
    want_n = wd.theirs.get_emb_norm_regularization_loss(wd.m_theirs)
    want_a = wd.theirs.get_emb_angle_regularization_loss(wd.m_theirs)
    got_n, got_a = T.emb_regularization_terms(wd.m_ours.frame_embeddings, wd.m_ours.raw_embeddings)
    _close(got_n, want_n, 1e-6)
    _close(got_a, want_a, 1e-6)
    g = torch.Generator().manual_seed(6)
    x, y = torch.randn(50, 2, generator=g) * 0.05, torch.randn(50, 2, generator=g) * 0.05
    _close(T.weighted_mean(T.huber(x, y), torch.ones(50, dtype=torch.bool)), wd.theirs.of_loss_fn(x, y).mean(), 1e-6)
    keep = torch.rand(50, generator=g) > 0.4
    preds = {"source_coords": torch.randn(50, 3, generator=g), "target_coords": torch.randn(50, 3, generator=g),
             "source_target_coords": torch.randn(50, 2, generator=g), "target_source_coords": torch.randn(50, 2, generator=g),
             "cycle_consistency_dists": torch.rand(50, generator=g) * 4}
    kept = {k: v[keep] for k, v in preds.items()}
    m_t = types.SimpleNamespace(get_cycle_consistent_preds=lambda frames, masks: kept)
    m_o = types.SimpleNamespace(get_cycle_consistency_terms=lambda frames, masks: dict(preds, keep=keep))
    _close(wd.ours.cycle_terms(m_o, wd.frames_set_t), wd.theirs.get_cycle_consistency_loss(m_t, (None, wd.frames_set_t)), 1e-6)


def test_key_sampling_is_uniform_without_replacement():
    """pick_subsets: every member equally likely, non-members never, no repetition (40 000 seeded draws of 3 out of 7)."""
    g = torch.Generator().manual_seed(7)
    members = torch.tensor([[1, 0, 1, 1, 0, 1, 1, 1, 0, 1]], dtype=torch.bool).repeat(40000, 1)
    idx, ok = T.pick_subsets([members], [3], generator=g)
    assert bool(ok.all()) and bool(members.gather(1, idx).all())
    assert bool((idx.sort(dim=1).values.diff(dim=1) != 0).all())
    freq = torch.bincount(idx.reshape(-1), minlength=10).float() / 40000
    assert bool((freq[~members[0]] == 0).all())
    assert (freq[members[0]] - 3 / 7).abs().max() < 0.01
    idx, ok = T.pick_subsets([members[:4] & (torch.arange(10) < 3)], [5], generator=g)     # 2 members, 5 asked for
    assert ok.sum(dim=1).tolist() == [2, 2, 2, 2]


def test_device_side_batch_sampler_properties():
    """DinoTrackerSampler.forward_device: points sit on their trajectories at tracked frames of the drawn frame sets, the two
    times differ, the indices address `frames_set_t`, labels are the normalised target points; a set with too few eligible
    trajectories comes back flagged."""
    g = torch.Generator().manual_seed(8)
    t, n = 12, 300

    def trajectories(n_valid):
        tr = torch.rand(n, t, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
        gone = torch.rand(n, t, generator=g) < 0.5
        gone[n_valid:] = True
        tr[gone] = float("nan")
        return tr

    fg, bg = trajectories(300), trajectories(300)
    rn = RangeNormalizer(shapes=(W, H, t))
    sampler = DinoTrackerSampler(batch_size=64, range_normalizer=rn, dst_range=(-1, 1), fg_trajectories=fg, bg_trajectories=bg,
                                 fg_traj_ratio=0.5, num_frames=4)
    torch.manual_seed(9)
    for _ in range(4):
        s = sampler.forward_device()
        frames = s["frames_set_t"].long()
        assert frames.tolist() == s["frames_set_t_host"] == sorted(set(frames.tolist())) and 4 <= len(frames) <= 8
        assert bool(s["valid"].all())
        t1f, t2f = frames[s["source_frame_indices"]], frames[s["target_frame_indices"]]
        assert bool((t1f != t2f).all())
        assert torch.equal(t2f.float(), s["target_times"])
        t2 = rn.unnormalize(s["t2_points_normalized"], src=(-1, 1))
        for half, traj in ((slice(0, 32), sampler.fg_valid_trajectories), (slice(32, 64), sampler.bg_valid_trajectories)):
            for b in range(half.start, half.stop):
                on1 = (traj[:, t1f[b]] - s["t1_points"][b, :2]).abs().sum(dim=1) == 0
                on2 = (traj[:, t2f[b]] - t2[b, :2]).abs().sum(dim=1) < 1e-3
                assert bool((on1 & on2).any()), "a sampled pair is one trajectory at two of its tracked frames"
    # a foreground set in which no trajectory is tracked twice
    sampler2 = DinoTrackerSampler(batch_size=64, range_normalizer=rn, dst_range=(-1, 1), fg_trajectories=fg, bg_trajectories=bg,
                                  fg_traj_ratio=0.5, num_frames=4)
    sampler2.fg_can_sample[:] = False
    sampler2.invalidate_host_tables()                       # (changed in place: the host copy the re-draw decides on is stale)
    with pytest.warns(RuntimeWarning, match="no set of 4 frames"):   # the reference would loop forever; here: said out loud, rows invalid
        s = sampler2.forward_device()
    assert not bool(s["valid"][:32].any()) and bool(s["valid"][32:].all()) and bool(torch.isfinite(s["t1_points"]).all())


def test_cells_at_equals_gather_of_frame_cells_and_lse_identity():
    """Host-side pieces of the fused contrastive path (round 4): trainer.cells_at reads the embeddings of given cells of
    indexed frames without the per-frame copy of frame_cells + gather; and the identity the kernels rest on --
    -log(exp(bb / t) / sum_j exp(s_j / t)) = logsumexp_j(s_j / t) - bb / t -- holds for trainer.contrastive_terms in float64."""
    g = torch.Generator().manual_seed(5)
    fe = torch.randn(5, 12, 4, 6, generator=g, dtype=torch.float64)
    sel = torch.tensor([3, 0, 3])
    cells = torch.randint(0, 24, (3, 7), generator=g)
    got = T.cells_at(fe, sel, cells)
    want = T.frame_cells(fe, sel).gather(1, cells[:, :, None].expand(-1, -1, 12))
    assert got.shape == (3, 7, 12) and torch.equal(got, want)
    a, b = torch.randn(3, 7, 12, generator=g, dtype=torch.float64), torch.randn(3, 7, 12, generator=g, dtype=torch.float64)
    fa, fb = T.frame_cells(fe, sel), T.frame_cells(fe, torch.tensor([1, 2, 4]))
    l_st, _ = T.contrastive_terms(a, b, fa, fb, 0.1)
    cos = lambda x, y: (x @ y.transpose(1, 2)) / torch.clamp(x.norm(dim=2)[:, :, None] * y.norm(dim=2)[:, None, :], min=T.EPS)
    bb = (a * b).sum(2) / torch.clamp(a.norm(dim=2) * b.norm(dim=2), min=T.EPS)
    assert torch.allclose(l_st, torch.logsumexp(cos(a, fb) / 0.1, dim=2) - bb / 0.1, rtol=1e-12, atol=1e-12)
    assert not T.fused_contrastive(fe)                      # a host tensor never takes the kernel path


def test_topk_rows_is_topk_for_long_rows():
    """train_ops.topk_rows (the k largest per row in levels of 2048-wide pieces, so that no call takes ATen's multi-block radix path --
    its memset-initialised semaphores do not survive a captured graph on the target stack): values and index consistency against
    torch.topk for rows up to 4e5 elements, more rows than one call takes, and the direct / fallback branches."""
    import torch
    from dino_tracker_amd.train_ops import topk_rows
    torch.manual_seed(0)
    for rows, n, k in [(2, 107604, 16), (4, 406504, 179), (3, 5000, 7), (2, 4097, 2048), (1, 100, 10), (2, 9000, 1024), (300, 2400, 5)]:
        x = torch.rand(rows, n)
        a = torch.topk(x, k, dim=1)
        v, i = topk_rows(x, k)
        assert torch.equal(a.values, v), (rows, n, k)
        assert torch.equal(x.gather(1, i), v)
        assert all(len(set(r.tolist())) == k for r in i)
    # members at -1 (the trainer's "key per element, non-members at -1" subsets): fewer members than k leaves -1 values at the end
    keys = torch.full((2, 50000), -1.0)
    keys[0, [5, 40000, 49999]] = torch.tensor([0.3, 0.9, 0.1])
    v, i = topk_rows(keys, 8)
    assert i[0, :3].tolist() == [40000, 5, 49999] and bool((v[0, 3:] == -1).all()) and bool((v[1] == -1).all())
