"""Test-time training WITHOUT the reference checkout:  python -m dino_tracker_amd.train --config C --data-path D [--seed S]

`overlay/dino_tracker.py` derives the trainer from the reference's own `DINOTracker` class, so its un-modified train.py keeps the
reference's control plane.  Where no checkout exists (the GPU box, `bench.py --train`, a deployment that ships this package only)
this module supplies the same control plane restated -- the parts of dino_tracker.py the device-side trainer leaves to its base
class: configuration and paths (:21-56), trajectories and sampler (:58-83), model, Adam over the two parameter groups and the
LambdaLR of optimization/schedulers.py:4-8 (:85-122), the running-loss bookkeeping (:355-389) -- under
`trainer.make_trainer`, whose `train()` is the loop of :392-448 with every iteration on the device (trainer.GraphedIteration).
Same on-disk layout in and out (utils.add_config_paths), same checkpoint files, same log line.

Differences from the reference's `train.py` (:1-26): `--seed` seeds torch / numpy like models/utils.py:98-104; no wandb / tqdm
set-up; the optional `DTK_TRAIN_LOG` JSON file (per-iteration loss values and a synchronised clock at both ends of the loop) is this
module's own, for tests and `bench.py --train`."""
from __future__ import annotations

import argparse
import json
import logging
import os
import time
from pathlib import Path

import numpy as np
import torch
import yaml

from .dataset import DinoTrackerSampler, RangeNormalizer
from .trainer import make_trainer
from .utils import add_config_paths


def load_masks(masks_path, h_resize=476, w_resize=854) -> torch.Tensor:
    """preprocessing/split_trajectories_to_fg_bg.py:38-52: the mask files as greyscale, nearest-resized -> [T, h, w] uint8."""
    from PIL import Image
    files = sorted(list(Path(masks_path).glob("*.jpg")) + list(Path(masks_path).glob("*.png")))
    masks = torch.from_numpy(np.stack([np.array(Image.open(f).convert("L")) for f in files])).unsqueeze(1)
    h_resize = masks.shape[2] if h_resize is None else h_resize
    w_resize = masks.shape[3] if w_resize is None else w_resize
    return torch.nn.functional.interpolate(masks, size=(h_resize, w_resize), mode="nearest")[:, 0]


def load_video(video_folder, resize=None) -> torch.Tensor:
    """data/data_utils.py:79-104: frames in file order, LANCZOS-resized to `resize` = (h, w), ToTensor (HWC uint8 -> CHW / 255)."""
    from PIL import Image
    files = sorted(list(Path(video_folder).glob("*.jpg")) + list(Path(video_folder).glob("*.png")))
    frames = []
    for f in files:
        img = Image.open(str(f))
        if resize is not None:
            img = img.resize((resize[1], resize[0]), Image.LANCZOS)
        frames.append(torch.from_numpy(np.asarray(img)).permute(2, 0, 1).float().div(255))
    return torch.stack(frames)


def last_ckpt_iter(folder) -> int:
    """models/utils.py:61-68."""
    return max([-1] + [int(f.split("_")[-1].split(".")[0]) for f in os.listdir(folder) if os.path.isfile(os.path.join(folder, f))])


class StandaloneBase:
    """The control plane `trainer.make_trainer`'s class expects from its base (dino_tracker.py:21-126, :355-389)."""

    def __init__(self, args):
        self.device = torch.device(getattr(args, "device", None) or ("cuda:0" if torch.cuda.is_available() else "cpu"))
        with open(args.config) as fh:
            self.config = yaml.safe_load(fh.read())
        p = add_config_paths(args.data_path, {})
        self.video_path, self.fg_masks_path, self.dino_embed_path = p["video_folder"], p["masks_path"], p["dino_embed_video_path"]
        self.fg_trajectories_path, self.bg_trajectories_path = p["fg_trajectories_file"], p["bg_trajectories_file"]
        self.dino_bb_path = os.path.join(p["dino_bb_dir"], "dino_best_buddies_filtered.pt")
        self.ckpt_folder = p["ckpt_folder"]
        os.makedirs(self.ckpt_folder, exist_ok=True)
        n_frames = len(list(Path(self.video_path).glob("*.jpg")) + list(Path(self.video_path).glob("*.png")))
        self.range_normalizer = RangeNormalizer(shapes=(self.config["video_resw"], self.config["video_resh"], n_frames)).to(self.device)
        self.of_loss_fn = torch.nn.HuberLoss(delta=1 / 32, reduction="none")

    def load_fg_masks(self):
        self.fg_masks = load_masks(self.fg_masks_path, h_resize=self.config["video_resh"]).to(self.device)

    def load_dino_best_buddies(self):
        self.dino_bb_pairs = torch.load(self.dino_bb_path, map_location=self.device)

    def get_sampler(self):
        cfg = self.config
        where = torch.device("cpu") if cfg["keep_traj_in_cpu"] else self.device
        fg = torch.load(self.fg_trajectories_path, map_location=where)
        bg = torch.load(self.bg_trajectories_path, map_location=where)
        return DinoTrackerSampler(fg_trajectories=fg, bg_trajectories=bg, fg_traj_ratio=cfg["fg_traj_ratio"],
                                  batch_size=cfg["train_batch_size"], range_normalizer=self.range_normalizer, dst_range=(-1, 1),
                                  num_frames=cfg["batch_n_frames"], keep_in_cpu=cfg["keep_traj_in_cpu"])

    def get_model(self):
        from .tracker import Tracker
        cfg = self.config
        video = load_video(self.video_path, resize=(cfg["video_resh"], cfg["video_resw"])).to(self.device)
        model = Tracker(video=video, device=self.device, dino_embed_path=self.dino_embed_path, dino_patch_size=cfg["dino_patch_size"],
                        stride=cfg["stride"], ckpt_path=self.ckpt_folder, cyc_n_frames=cfg["cyc_n_frames"],
                        cyc_batch_size_per_frame=cfg["cyc_batch_size_per_frame"], cyc_fg_points_ratio=cfg["cyc_fg_points_ratio"],
                        cyc_thresh=cfg["cyc_thresh"]).to(self.device)
        self.init_iter = last_ckpt_iter(self.ckpt_folder)
        if self.init_iter > 0:
            model.load_weights(self.init_iter)
        return model

    def train_setup(self):
        cfg = self.config
        model = self.get_model()
        optimizer = torch.optim.Adam([{"params": model.delta_dino.parameters(), "lr": cfg["lr_delta_dino"]},
                                      {"params": model.tracker_head.parameters(), "lr": cfg["lr_cnn_refiner"]}])
        gamma, every = cfg["scheduler_gamma"], cfg["apply_scheduler_every"]
        # optimization/schedulers.py:4-8: the FIRST group's rate decays by gamma every `apply_every` steps, the second stays
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=[lambda e: gamma ** (e // every), lambda e: 1])
        for _ in range(max(self.init_iter, 0)):
            scheduler.step()
        print("------- INIT ITER", self.init_iter)
        return model, optimizer, scheduler

    def set_model_train(self, model):
        model.train()

    # -- running losses (dino_tracker.py:355-389) ---------------------------------------------------------------------------------
    NAMES = ("total", "of", "cl_dino_bb", "cl_refiner", "emb_norm_reg", "angle_reg", "cyc")

    def init_losses(self):
        self.running = [0.0] * len(self.NAMES)

    def update_losses(self, *values):
        self.running = [r + v for r, v in zip(self.running, values)]
        log = getattr(self, "_loss_log", None)
        if log is not None:
            log.append(torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in values]))

    def log_losses(self, i, log_interval=100):
        r = dict(zip(self.NAMES, (float(x) / log_interval for x in self.running)))
        line = (f"loss_of: {r['of']:.4f}, loss_cl_dino_bb: {r['cl_dino_bb']:.4f}, loss_emb_norm_reg: {r['emb_norm_reg']:.4f}, "
                f"loss_angle_reg: {r['angle_reg']:.4f}")
        if i >= self.config.get("apply_cl_ref_after", 0):
            line += f", loss_cl_refiner: {r['cl_refiner']:.4f}"
        if i >= self.config.get("apply_cyc_after", 0):
            line += f", loss_cyc: {r['cyc']:.4f}"
        logging.info(line + f", loss_total: {r['total']:.4f}")
        self.init_losses()


def standalone_trainer(args):
    """An instance of the device-side trainer over the restated control plane (`args`: .config, .data_path, optional .device)."""
    return make_trainer(StandaloneBase)(args)


def fix_random_seeds(seed: int) -> None:
    """models/utils.py:98-104."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--config", default="./config/train.yaml")
    ap.add_argument("--data-path", default="./dataset/libby", dest="data_path")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--device", default=None)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    fix_random_seeds(args.seed)
    tr = standalone_trainer(args)
    log_path = os.environ.get("DTK_TRAIN_LOG")
    if log_path:
        tr._loss_log = []
    t0 = time.time()
    tr.train()
    if tr.device.type == "cuda":
        torch.cuda.synchronize()
    if log_path:
        with open(log_path, "w") as fh:
            json.dump({"names": list(StandaloneBase.NAMES), "losses": [v.tolist() for v in tr._loss_log],
                       "wall_s": time.time() - t0}, fh)


if __name__ == "__main__":
    main()
