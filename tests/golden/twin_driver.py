"""TEST INFRASTRUCTURE -- the call sequence of the reference's inference_grid.py / inference_benchmark.py WITHOUT the reference
checkout: what `dino_tracker.DINOTracker.__init__ / get_model` (dino_tracker.py:21-115) and the two scripts' `run`
(inference_grid.py:12-41, inference_benchmark.py:13-46) do, restated, against the module paths the scripts import --
`models.tracker`, `models.model_inference`, `data.dataset`, `utils` -- which resolve to overlay/ (= this implementation).
The driver's box has no reference, so the un-modified-script tests skip there; this twin exercises the same API surface
(constructor kwargs, load_weights, .video.shape, ModelInference, infer, the on-disk layout of utils.py:10-29) on inputs
that can be rebuilt anywhere (ref_scripts_data.build_synth_data_dir), and its outputs are compared with what the
UN-MODIFIED scripts wrote on the reference's own PyTorch code (tests/golden/ref_scripts_synth.npz, make_golden.py).

    python twin_driver.py grid <data> <interval>          python twin_driver.py benchmark <data> <pickle> <video_id>
"""
import os
import pickle
import sys
from pathlib import Path

import numpy as np
import torch
from PIL import Image

# config/train.yaml values the two scripts read (config/train.yaml:6-7,58-63,28-33)
CONFIG = dict(video_resw=854, video_resh=476, dino_patch_size=14, stride=7, anchor_cosine_similarity_threshold=0.7,
              cosine_similarity_threshold=0.6, cyc_n_frames=4, cyc_batch_size_per_frame=256, cyc_fg_points_ratio=0.7, cyc_thresh=4)


def load_video(folder, resize):  # data/data_utils.py:79-104 (ToTensor = HWC uint8 -> CHW float / 255)
    files = sorted(list(Path(folder).glob("*.jpg")) + list(Path(folder).glob("*.png")))
    resh, resw = resize
    frames = [torch.from_numpy(np.asarray(Image.open(str(f)).resize((resw, resh), Image.LANCZOS))).permute(2, 0, 1).float().div(255)
              for f in files]
    return torch.stack(frames)


def last_ckpt_iter(folder):  # models/utils.py:61-68
    its = [-1] + [int(f.split("_")[-1].split(".")[0]) for f in os.listdir(folder) if os.path.isfile(os.path.join(folder, f))]
    return max(its)


def build(data_path, device):
    """DINOTracker.__init__ + get_model (dino_tracker.py:21-28, 85-108)."""
    from data.dataset import RangeNormalizer
    from models.tracker import Tracker
    from utils import add_config_paths
    paths = add_config_paths(data_path, {})
    files = sorted(list(Path(paths["video_folder"]).glob("*.jpg")) + list(Path(paths["video_folder"]).glob("*.png")))
    orig_h, orig_w = Image.open(files[0]).size[::-1]
    rn = RangeNormalizer(shapes=(CONFIG["video_resw"], CONFIG["video_resh"], len(files))).to(device)
    os.makedirs(paths["ckpt_folder"], exist_ok=True)
    video = load_video(paths["video_folder"], (CONFIG["video_resh"], CONFIG["video_resw"])).to(device)
    model = Tracker(video=video, device=device, dino_embed_path=paths["dino_embed_video_path"],
                    dino_patch_size=CONFIG["dino_patch_size"], stride=CONFIG["stride"], ckpt_path=paths["ckpt_folder"],
                    cyc_n_frames=CONFIG["cyc_n_frames"], cyc_batch_size_per_frame=CONFIG["cyc_batch_size_per_frame"],
                    cyc_fg_points_ratio=CONFIG["cyc_fg_points_ratio"], cyc_thresh=CONFIG["cyc_thresh"]).to(device)
    it = last_ckpt_iter(paths["ckpt_folder"])
    if it > 0:
        model.load_weights(it)
    return paths, rn, model, (orig_h, orig_w)


@torch.no_grad()
def main(argv):
    from models.model_inference import ModelInference
    import models.tracker
    overlay = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "overlay")
    assert os.path.abspath(models.tracker.__file__).startswith(overlay), models.tracker.__file__
    device = "cuda:0"
    mode, data_path = argv[0], argv[1]
    paths, rn, model, (orig_h, orig_w) = build(data_path, device)
    mi = ModelInference(model=model, range_normalizer=rn,
                        anchor_cosine_similarity_threshold=CONFIG["anchor_cosine_similarity_threshold"],
                        cosine_similarity_threshold=CONFIG["cosine_similarity_threshold"])
    mh, mw = model.video.shape[-2], model.video.shape[-1]
    if mode == "grid":  # inference_grid.py:31-41; get_grid_query_points: data/data_utils.py:167-197
        interval = int(argv[2])
        gy, gx = torch.meshgrid(torch.arange(0, orig_h, interval, device=device), torch.arange(0, orig_w, interval, device=device),
                                indexing="ij")
        pts = torch.stack([gx, gy], dim=-1).reshape(-1, 2)
        q = torch.cat([pts, torch.zeros_like(pts[:, :1])], dim=1) * torch.tensor([mw / orig_w, mh / orig_h, 1.0], device=device)
        traj, occ = mi.infer(q, batch_size=None)
        for d in (paths["grid_trajectories_dir"], paths["grid_occlusions_dir"]):
            os.makedirs(d, exist_ok=True)
        np.save(os.path.join(paths["grid_trajectories_dir"], "grid_trajectories.npy"), traj[..., :2].cpu().detach().numpy())
        np.save(os.path.join(paths["grid_occlusions_dir"], "grid_occlusions.npy"), occ.cpu().detach().numpy())
    else:  # inference_benchmark.py:32-46; data/tapvid.py:19-41
        bench = pickle.load(open(argv[2], "rb"))
        vc = [v for v in bench["videos"] if v["video_idx"] == int(argv[3])][0]
        fx, fy = mw / vc["w"], mh / vc["h"]
        for d in (paths["trajectories_dir"], paths["occlusions_dir"]):
            os.makedirs(d, exist_ok=True)
        for f in sorted(vc["query_points"].keys()):
            q = torch.tensor([[fx * p[0], fy * p[1], f] for p in vc["query_points"][f]], dtype=torch.float32, device=device)
            traj, occ = mi.infer(query_points=q, batch_size=None)
            np.save(os.path.join(paths["trajectories_dir"], f"trajectories_{f}.npy"), traj[..., :2].cpu().detach().numpy())
            np.save(os.path.join(paths["occlusions_dir"], f"occlusion_preds_{f}.npy"), occ.cpu().detach().numpy())
    print("twin ok")


if __name__ == "__main__":
    main(sys.argv[1:])
