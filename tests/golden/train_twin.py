"""TEST INFRASTRUCTURE -- the reference's train.py WITHOUT the reference checkout (BASELINE.json config 5 at reduced size).

The driver's GPU box has no reference, so the un-modified-script tests of tests/test_gpu_train.py skip there.  This twin restates
what `train.py` + `dino_tracker.DINOTracker` do around this implementation's models, IN THE REFERENCE'S ORDER OF RANDOM DRAWS,
so that a run from the same seed consumes torch's (host) generator exactly like the un-modified script did when
tests/golden/make_golden.py recorded `ref_train_synth.npz` on the reference's own PyTorch code:

  * control plane (dino_tracker.py:21-126): config, paths (utils.py:10-29), foreground masks (preprocessing/
    split_trajectories_to_fg_bg.py:38-52), best buddies, sampler, model, Adam over the two parameter groups, the LambdaLR
    schedule of optimization/schedulers.py:4-8 (head group: gamma ** (epoch // apply_every); Delta-DINO group: 1), the start
    checkpoint and `init_scheduler`;
  * one iteration (dino_tracker.py:392-448): batch -> forward -> Huber tracking term -> cycle-consistency term -> refiner
    contrastive term -> DINO best-buddy contrastive term -> the two regularisers -> backward -> Adam -> scheduler.  The random
    parts are drawn here in the reference's sequence (`*_reference_order`: randint, randint [re-drawn while a pair has source ==
    target, :161-164], then per frame pair randperm over its foreground and over its background candidates, :197-206 / :298-299;
    pairs without candidates draw nothing) and handed as EXPLICIT selections to the loss terms of dino_tracker_amd/trainer.py
    (`refined_bb_terms`, `dino_bb_terms`: the functions the device-side trainer evaluates with its own selections); the batch
    comes from `DinoTrackerSampler.forward` and the cycle-consistency point sets from `Tracker.get_cycle_consistent_preds` with
    `cyc_sampling = "reference"`, both of which keep the reference's draw order;
  * like tests/golden/train_driver.py, every draw comes from torch's CPU generator (the shims below), so that the device run and
    the CPU golden run see the same indices; the reference's log column "of" is the total (its `loss += ...` adds to
    `tracking_loss` in place, :409-426), reproduced here.

    python train_twin.py --config C --data-path D --seed S [--log losses.json]
"""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import yaml
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

_randperm, _randint, _multinomial = torch.randperm, torch.randint, torch.Tensor.multinomial


def _host_draws():
    """torch.randperm / randint / Tensor.multinomial on the CPU generator whatever device is asked for (as train_driver.py)."""
    def randperm(n, *a, device=None, **k):
        out = _randperm(n, *a, **k)
        return out if device is None else out.to(device)

    def randint(*a, device=None, **k):
        out = _randint(*a, **k)
        return out if device is None else out.to(device)

    def multinomial(self, *a, **k):
        return _multinomial(self.cpu(), *a, **k).to(self.device)

    torch.randperm, torch.randint, torch.Tensor.multinomial = randperm, randint, multinomial


def load_masks(path, h_resize, w_resize=854):  # split_trajectories_to_fg_bg.py:38-52
    files = sorted(list(Path(path).glob("*.jpg")) + list(Path(path).glob("*.png")))
    masks = torch.from_numpy(np.stack([np.array(Image.open(f).convert("L")) for f in files])).unsqueeze(1)
    return torch.nn.functional.interpolate(masks, size=(h_resize, w_resize), mode="nearest")[:, 0]


def load_video(folder, resize):  # data/data_utils.py:79-104 (ToTensor = HWC uint8 -> CHW float / 255)
    files = sorted(list(Path(folder).glob("*.jpg")) + list(Path(folder).glob("*.png")))
    resh, resw = resize
    return torch.stack([torch.from_numpy(np.asarray(Image.open(str(f)).resize((resw, resh), Image.LANCZOS))).permute(2, 0, 1).float().div(255)
                        for f in files])


def last_ckpt_iter(folder):  # models/utils.py:61-68
    return max([-1] + [int(f.split("_")[-1].split(".")[0]) for f in os.listdir(folder) if os.path.isfile(os.path.join(folder, f))])


def _pad(rows, device):
    width = max([1] + [len(r) for r in rows])
    idx = torch.zeros(len(rows), width, dtype=torch.long)
    ok = torch.zeros(len(rows), width, dtype=torch.bool)
    for i, r in enumerate(rows):
        idx[i, :len(r)] = r
        ok[i, :len(r)] = True
    return idx.to(device), ok.to(device)


def refined_bb_selection_reference_order(tr, model, frames_set_t):
    """The draws of get_refined_bb_contrastive_loss (dino_tracker.py:245-305) -> (s_sel, t_sel, src_cells, tgt_cells, ok)."""
    from dino_tracker_amd import trainer as T
    fe = model.frame_embeddings
    dev, n, P = fe.device, frames_set_t.shape[0], tr.config["cl_n_frames"]
    s_sel = torch.randint(n, (P,), device=dev)
    t_sel = torch.randint(n, (P,), device=dev)
    geom = model.tracker_head.geom(fe.shape[0], fe.shape[1]) if fe.is_cuda else None
    nn_st, nn_ts = T.mutual_argmax(fe.detach(), s_sel, t_sel, geom)
    nn_st_h, nn_ts_h = nn_st.cpu(), nn_ts.cpu()
    n_fg, n_bg = tr.split_counts(tr.config["cl_points_per_pair"], tr.config["cl_fg_points_ratio"])
    cells = torch.arange(nn_st_h.shape[1])
    cell_fg = tr._cell_fg.cpu()
    frames_h = frames_set_t.cpu()
    src_rows, tgt_rows = [], []
    for p in range(P):
        mutual = nn_ts_h[p][nn_st_h[p]] == cells
        if int(mutual.sum()) == 0:      # :281-282: the pair is skipped before anything is drawn
            src_rows.append(torch.zeros(0, dtype=torch.long))
            tgt_rows.append(torch.zeros(0, dtype=torch.long))
            continue
        mc = cells[mutual]
        fg = cell_fg[int(frames_h[int(s_sel[p])])][mc]
        fg_c, bg_c = mc[fg], mc[~fg]
        fsel = torch.randperm(fg_c.shape[0])[:n_fg]
        bsel = torch.randperm(bg_c.shape[0])[:n_bg]
        src = torch.cat([fg_c[fsel], bg_c[bsel]])
        src_rows.append(src)
        tgt_rows.append(nn_st_h[p][src])
    src_cells, ok = _pad(src_rows, dev)
    tgt_cells, _ = _pad(tgt_rows, dev)
    return s_sel, t_sel, src_cells, tgt_cells, ok


def dino_bb_selection_reference_order(tr, frames_set_t, n_frames):
    """The draws of get_dino_bb_contrastive_loss (dino_tracker.py:159-212) -> (s_sel, t_sel, picks into the packed table, ok)."""
    dev, n, P = frames_set_t.device, frames_set_t.shape[0], tr.config["cl_n_frames"]
    s_sel = torch.randint(n, (P,), device=dev)
    t_sel = torch.randint(n, (P,), device=dev)
    while bool((s_sel == t_sel).any()):
        t_sel = torch.randint(n, (P,), device=dev)
    n_fg, n_bg = tr.split_counts(tr.config["cl_points_per_pair"], tr.config["cl_fg_points_ratio"])
    tb = tr._bb_table
    slot_h, off_h, cnt_h, fg_h = tb.slot.cpu(), tb.off.cpu(), tb.cnt.cpu(), tb.fg.cpu()
    frames_h = frames_set_t.cpu()
    rows = []
    for s_i, t_i in zip(s_sel.tolist(), t_sel.tolist()):
        slot = int(slot_h[int(frames_h[s_i]) * n_frames + int(frames_h[t_i])])
        if slot < 0:                    # no best buddies for this pair (:181-183): skipped before anything is drawn
            rows.append(torch.zeros(0, dtype=torch.long))
            continue
        base, cnt = int(off_h[slot]), int(cnt_h[slot])
        fg = fg_h[base:base + cnt]
        ar = torch.arange(cnt)
        fg_i, bg_i = ar[fg], ar[~fg]
        fsel = torch.randperm(fg_i.shape[0])[:n_fg]
        bsel = torch.randperm(bg_i.shape[0])[:n_bg]
        rows.append(base + torch.cat([fg_i[fsel], bg_i[bsel]]))
    picks, ok = _pad(rows, dev)
    return s_sel, t_sel, picks, ok


class Twin:
    def __init__(self, config_path, data_path, device):
        from dino_tracker_amd.dataset import RangeNormalizer
        from dino_tracker_amd.utils import add_config_paths
        with open(config_path) as fh:
            self.config = yaml.safe_load(fh.read())
        self.device = device
        self.paths = add_config_paths(data_path, {})
        self.ckpt_folder = self.paths["ckpt_folder"]
        os.makedirs(self.ckpt_folder, exist_ok=True)
        files = sorted(list(Path(self.paths["video_folder"]).glob("*.jpg")) + list(Path(self.paths["video_folder"]).glob("*.png")))
        self.range_normalizer = RangeNormalizer(shapes=(self.config["video_resw"], self.config["video_resh"], len(files))).to(device)
        self.of_loss_fn = torch.nn.HuberLoss(delta=1 / 32, reduction="none")

    def setup(self):
        from dino_tracker_amd import trainer as T
        from dino_tracker_amd.dataset import DinoTrackerSampler
        if os.environ.get("DTK_TWIN_MODEL") == "reference":
            # CPU cross-check of THIS FILE where the checkout exists (tests/test_train_vs_reference.py): the reference's own
            # Tracker (PYTHONPATH = oracle/shims : reference) under the twin's control plane, draw order and loss terms
            if not torch.cuda.is_available():
                torch.Tensor.cuda = lambda self_, *a, **k: self_
            import data.dataset as ref_ds
            if ref_ds.RangeNormalizer.__init__.__defaults__ == ("cuda",):
                ref_ds.RangeNormalizer.__init__.__defaults__ = ("cpu",)
            import models.utils as ref_mu
            ref_mu.get_vit_feature_coords_from_mask.__defaults__ = (7, 14, "cpu")
            from models.tracker import Tracker
        else:
            from dino_tracker_amd.tracker import Tracker
        cfg, dev = self.config, self.device
        self.fg_masks = load_masks(self.paths["masks_path"], h_resize=cfg["video_resh"]).to(dev)
        self.dino_bb_pairs = torch.load(os.path.join(self.paths["dino_bb_dir"], "dino_best_buddies_filtered.pt"), map_location=dev)
        trj_dev = torch.device("cpu") if cfg["keep_traj_in_cpu"] else dev
        fg_tr = torch.load(self.paths["fg_trajectories_file"], map_location=trj_dev)
        bg_tr = torch.load(self.paths["bg_trajectories_file"], map_location=trj_dev)
        self.sampler = DinoTrackerSampler(fg_trajectories=fg_tr, bg_trajectories=bg_tr, fg_traj_ratio=cfg["fg_traj_ratio"],
                                          batch_size=cfg["train_batch_size"], range_normalizer=self.range_normalizer,
                                          dst_range=(-1, 1), num_frames=cfg["batch_n_frames"], keep_in_cpu=cfg["keep_traj_in_cpu"])
        video = load_video(self.paths["video_folder"], (cfg["video_resh"], cfg["video_resw"])).to(dev)
        model = Tracker(video=video, device=dev, dino_embed_path=self.paths["dino_embed_video_path"],
                        dino_patch_size=cfg["dino_patch_size"], stride=cfg["stride"], ckpt_path=self.ckpt_folder,
                        cyc_n_frames=cfg["cyc_n_frames"], cyc_batch_size_per_frame=cfg["cyc_batch_size_per_frame"],
                        cyc_fg_points_ratio=cfg["cyc_fg_points_ratio"], cyc_thresh=cfg["cyc_thresh"]).to(dev)
        model.cyc_sampling = "reference"
        self.init_iter = last_ckpt_iter(self.ckpt_folder)
        if self.init_iter > 0:
            model.load_weights(self.init_iter)
        opt = torch.optim.Adam([{"params": model.delta_dino.parameters(), "lr": cfg["lr_delta_dino"]},
                                {"params": model.tracker_head.parameters(), "lr": cfg["lr_cnn_refiner"]}])
        gamma, every = cfg["scheduler_gamma"], cfg["apply_scheduler_every"]
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=[lambda e: gamma ** (e // every), lambda e: 1])
        for _ in range(max(self.init_iter, 0)):
            sched.step()
        model.train()
        # the loss terms: this implementation's (dino_tracker_amd/trainer.py), on explicit selections
        self.terms = object.__new__(T.make_trainer(object))
        self.terms.config, self.terms.fg_masks, self.terms.dino_bb_pairs = cfg, self.fg_masks, self.dino_bb_pairs
        self.terms.prepare_tables(model)
        self.T = T
        return model, opt, sched

    def iteration(self, model, i):
        cfg, T = self.config, self.T
        sample = self.sampler()
        labels = sample["t2_points_normalized"][:, :-1]
        inputs = (sample["t1_points"], sample["source_frame_indices"], sample["target_frame_indices"], sample["frames_set_t"])
        frames_set_t = inputs[-1]
        coords = model(inputs)
        loss = self.of_loss_fn(coords, labels).mean()
        cyc = cl_ref = torch.zeros((), device=coords.device)
        if i >= cfg.get("apply_cyc_after", 0):          # dino_tracker.py:332-341
            c = model.get_cycle_consistent_preds(frames_set_t, self.fg_masks)
            w = cfg["cyc_gamma"] ** c["cycle_consistency_dists"]
            st = w[:, None] * self.of_loss_fn(c["source_target_coords"], c["target_coords"][:, :2])
            ts = w[:, None] * self.of_loss_fn(c["target_source_coords"], c["source_coords"][:, :2])
            cyc = (st.mean() + ts.mean()) / 2
            loss = loss + cfg["lambda_cyc"] * cyc
        if i >= cfg.get("apply_cl_ref_after", 0):
            cl_ref = self.terms.refined_bb_terms(model, *refined_bb_selection_reference_order(self.terms, model, frames_set_t))
            loss = loss + cfg["lambda_cl_ref_bb"] * cl_ref
        cl_bb = self.terms.dino_bb_terms(model, *dino_bb_selection_reference_order(self.terms, frames_set_t, model.video.shape[0]))
        norm_reg, angle_reg = T.emb_regularization_terms(model.frame_embeddings, model.raw_embeddings)
        loss = loss + cfg["lambda_cl_dino_bb"] * cl_bb + cfg["lambda_emb_norm"] * norm_reg + cfg["lambda_angle"] * angle_reg
        return loss, [float(loss), float(loss), float(cl_bb), float(cl_ref), float(norm_reg), float(angle_reg), float(cyc)]

    def train(self):
        model, opt, sched = self.setup()
        cfg = self.config
        total = cfg["total_iterations"]
        losses = []
        for i in range(self.init_iter, total):
            opt.zero_grad()
            loss, vals = self.iteration(model, i)
            loss.backward()
            opt.step()
            sched.step()
            losses.append(vals)
            if i == total - 1 or i % cfg["checkpoint_interval"] == 0:
                model.save_weights(i)
        model.save_weights(total)
        return losses


    def compare_device_terms(self, draws):
        """One batch and one model state, two evaluations of the seven loss values: this file's (the reference's draw order) and
        `draws` evaluations of the device-side trainer's `iteration_losses` (dino_tracker_amd/trainer.py: its own key-based
        selections, its static-shape cycle batch).  The tracking term and the regularisers see identical inputs; the
        stochastic terms are sample means of the same quantity."""
        model, _, _ = self.setup()
        i = self.init_iter
        sample = self.sampler()
        labels = sample["t2_points_normalized"][:, :-1]
        inputs = (sample["t1_points"], sample["source_frame_indices"], sample["target_frame_indices"], sample["frames_set_t"])
        state = torch.get_rng_state()

        def reference_order():
            torch.set_rng_state(state)
            # (iteration() draws its own batch: hand it this one)
            keep = self.sampler
            self.sampler = lambda: sample
            try:
                with torch.no_grad():
                    return self.iteration(model, i)[1]
            finally:
                self.sampler = keep

        ref_vals = reference_order()
        valid = torch.ones(labels.shape[0], dtype=torch.bool, device=labels.device)
        model.cyc_sampling = "device"
        dev_vals = []
        for _ in range(draws):
            with torch.no_grad():
                dev_vals.append(self.terms.iteration_losses(model, inputs, labels, valid, i)[1].tolist())
        return ref_vals, dev_vals


def main(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--data-path", required=True)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--log", default=os.environ.get("DTK_TRAIN_LOG"))
    ap.add_argument("--device", default="cuda:0" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--compare-device-terms", type=int, default=0, metavar="K",
                    help="instead of training: the seven loss values of the first iteration in the reference's draw order and K "
                         "evaluations of the device-side trainer's iteration on the same batch and weights")
    args = ap.parse_args(argv)
    _host_draws()
    torch.manual_seed(args.seed)       # models/utils.py:98-104 (fix_random_seeds)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(args.seed)
    np.random.seed(args.seed)
    if args.compare_device_terms:
        ref_vals, dev_vals = Twin(args.config, args.data_path, args.device).compare_device_terms(args.compare_device_terms)
        with open(args.log, "w") as fh:
            json.dump({"names": ["total", "of", "cl_dino_bb", "cl_refiner", "emb_norm_reg", "angle_reg", "cyc"],
                       "reference_order": ref_vals, "device_trainer": dev_vals}, fh)
        print("twin ok (terms)")
        return
    losses = Twin(args.config, args.data_path, args.device).train()
    if args.log:
        with open(args.log, "w") as fh:
            json.dump({"names": ["total", "of", "cl_dino_bb", "cl_refiner", "emb_norm_reg", "angle_reg", "cyc"], "losses": losses}, fh)
    print("twin ok", losses[-1][0])


if __name__ == "__main__":
    main(sys.argv[1:])
