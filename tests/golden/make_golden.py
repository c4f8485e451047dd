"""Writes tests/golden/*.npz by running the UN-MODIFIED reference (/root/reference) on CPU through
oracle/ref_harness.py.  Run in the build container only:  python tests/golden/make_golden.py

Inputs are regenerated from seeds by dino_tracker_amd.synth (bit-identical on any machine), so the fixtures hold
only the reference's OUTPUTS plus the few intermediate tensors the parity tests need.
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness  # noqa: E402
from dino_tracker_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (H, W, T, C, queries(nx,ny,t), head_seed, benign, feat_seed, delta: None|seed)
    "p3_small": dict(H=140, W=210, T=8, C=32, q=[(4, 3, 0), (2, 2, 5)], head_seed=3, benign=True, feat_seed=10, delta=None),
    "p3_small_wild": dict(H=140, W=210, T=8, C=32, q=[(4, 3, 2)], head_seed=5, benign=False, feat_seed=11, delta=None),
    "p3_full": dict(H=476, W=854, T=4, C=384, q=[(3, 2, 1)], head_seed=3, benign=True, feat_seed=12, delta=None),
    "p23_small": dict(H=140, W=210, T=5, C=32, q=[(3, 2, 0)], head_seed=3, benign=True, feat_seed=13, delta=4),
}


def build_inputs(cfg):
    ph, pw = 1 + (cfg["H"] - 14) // 7, 1 + (cfg["W"] - 14) // 7
    video = synth.synth_video(cfg["T"], cfg["H"], cfg["W"], seed=cfg["feat_seed"] + 100)
    dino = synth.synth_features(cfg["T"], cfg["C"], ph, pw, seed=cfg["feat_seed"])
    head = synth.synth_head_weights(cfg["head_seed"], cfg["benign"])
    queries = torch.cat([synth.grid_queries(nx, ny, cfg["H"], cfg["W"], t, margin=20.0) for nx, ny, t in cfg["q"]])
    delta = synth.synth_delta_dino_weights(cfg["C"], cfg["delta"]) if cfg["delta"] is not None else None
    return video, dino, head, queries, delta


def run_reference(cfg):
    R = ref_harness.load()
    video, dino, head, queries, delta = build_inputs(cfg)
    tmp = tempfile.mkdtemp()
    emb_path = os.path.join(tmp, "dino_embed_video.pt")
    torch.save(dino, emb_path)
    trk = R.tracker.Tracker(video=video, ckpt_path=tmp, dino_embed_path=emb_path, dino_patch_size=14, stride=7,
                            device="cpu")
    # the reference hard-codes a 1024-wide DeltaDINO (delta_dino.py:9); build it at the embedding width instead
    trk.delta_dino = R.delta_dino.DeltaDINO(channels=[3, 64, 128, 256, cfg["C"]], vit_stride=7)
    if delta is not None:
        trk.delta_dino.load_state_dict(delta)
    trk.tracker_head.load_state_dict(head)
    rn = R.dataset.RangeNormalizer(shapes=(cfg["W"], cfg["H"], cfg["T"]))
    mi = R.model_inference.ModelInference(trk, rn, anchor_cosine_similarity_threshold=0.7,
                                          cosine_similarity_threshold=0.6)
    with torch.no_grad():
        trajs = mi.compute_trajectories(queries, None)
        cs = mi.compute_trajectory_cos_sims(trajs, queries)
        anchors = mi.compute_anchor_trajectories(trajs, cs, None)
        occ = mi.compute_occlusion(trajs, cs, anchors)
        # single-call forward fixtures (Tracker.forward API): track query 0 into every frame
        inp = R.model_inference.generate_trajectory_input(queries[0], video)
        fwd = trk(inp)
        # TrackerHead alone on crafted maps
        hm = torch.zeros(4, 1, dino.shape[-2], dino.shape[-1])
        g = torch.Generator().manual_seed(99)
        hm[1] = torch.rand(1, dino.shape[-2], dino.shape[-1], generator=g)
        hm[2, 0, 0, 0] = 0.9
        hm[3, 0, -1, -1] = 0.7
        hm[3, 0, 3, 4] = 0.7
        head_out = trk.tracker_head(hm.clone())
    out = dict(traj=trajs[..., :2].numpy(), occ=occ.numpy(), cos_sims=cs.numpy(),
               n_anchors=np.array([anchors[i].shape[0] for i in range(len(anchors))]),
               anchor0=anchors[0].numpy(), fwd_q0=fwd.numpy(), head_maps=hm.numpy(), head_out=head_out.numpy(),
               queries=queries.numpy())
    if delta is not None:
        out["refined"] = trk.refined_features.detach().numpy()
    return out


# ---- P1: the reference's own VitExtractor / get_dino_features_video around a DINOv2-API stub --------------------------
P1_CASE = dict(H=98, W=126, T=2, seed=74, model="dinov2_vits14")


def p1_video():
    return synth.synth_video(P1_CASE["T"], P1_CASE["H"], P1_CASE["W"], seed=P1_CASE["seed"])


def p1_weights(layerscale):
    return synth.make_vit_weights(P1_CASE["model"], seed=2, layerscale=layerscale)


def run_reference_p1():
    """models/extractor.py:23-150 + utils.py:33-72 un-modified; torch.hub.load is answered by oracle/shims/dinov2_stub
    (upstream is un-vendored and needs the network) loaded with the seeded weights."""
    R = ref_harness.load()
    import dinov2_stub
    video = p1_video()
    name = P1_CASE["model"]
    out = {}
    orig = torch.hub.load
    try:
        for tag, ls in (("ls1", 1.0), ("ls01", 0.1)):
            sd = p1_weights(ls)
            torch.hub.load = lambda repo, model_name, *a, **k: dinov2_stub.build(model_name, sd)
            with torch.no_grad():
                out[f"tokens_{tag}_l11"] = R.utils.get_dino_features_video(
                    video if tag == "ls1" else video[:1], model_name=name, facet="tokens", stride=7, layer=None,
                    device="cpu").numpy()  # layer=None: the last block (utils.py:51)
                if tag == "ls1":
                    out["tokens_ls1_l3"] = R.utils.get_dino_features_video(
                        video[:1], model_name=name, facet="tokens", stride=7, layer=3, device="cpu").numpy()
                    out["keys_l3"] = R.utils.get_dino_features_video(
                        video[:1], model_name=name, facet="keys", stride=7, layer=3, device="cpu").numpy()
                    ex = R.extractor.VitExtractor(model_name=name, stride=7, device="cpu").eval()
                    m = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)
                    sdev = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
                    x = (video[:1] - m) / sdev
                    out["feature_with_cls_l2_l5"] = ex.get_feature_from_input(x, layers=[2, 5]).numpy()  # mean over layers
                    out["qkv_l1"] = ex.get_qkv_feature_from_input(x)[1].numpy()
                    out["keys_self_sim_l1"] = ex.get_keys_self_sim_from_input(x, layer_num=1).numpy()
    finally:
        torch.hub.load = orig
    return out


# ---- configs 1 and 3: the reference's SCRIPTS, un-modified, on CPU ------------------------------------------------------
def run_reference_scripts(only_cfg3=False, synthetic=False):
    """inference_grid.py (config 1) and inference_benchmark.py + eval/metrics.py (configs 3-4) of the reference, run as
    __main__ through runpy with the CPU shims; their .npy outputs and the TAP-Vid metrics are the golden values the
    `-m gpu` test compares the launcher-run HIP outputs with (tests/test_gpu_reference_scripts.py).
    synthetic=True: the same scripts on the synthetic-video data directories (ref_scripts_data.CFG1S / CFG3S), whose inputs
    can be rebuilt WITHOUT the reference checkout -> ref_scripts_synth.npz, the golden of the twin tests."""
    import runpy
    import ref_scripts_data as D
    R = ref_harness.load()
    if synthetic:
        cfg1, cfg3 = D.CFG1S, D.CFG3S
        build = lambda dst, ref, cfg: D.build_synth_data_dir(dst, cfg)  # noqa: E731
    else:
        cfg1, cfg3, build = D.CFG1, D.CFG3, D.build_data_dir
    ref = ref_harness.REFERENCE_ROOT
    out = {}
    tmp = tempfile.mkdtemp()
    argv0 = sys.argv
    cwd = os.getcwd()
    try:
        os.chdir(ref)
        if only_cfg3:  # keep the (slow) config-1 outputs of the existing fixture
            old = np.load(os.path.join(OUT, "ref_scripts.npz"))
            out["cfg1_traj"], out["cfg1_occ"] = old["cfg1_traj"], old["cfg1_occ"]
        else:
            d1 = build(os.path.join(tmp, "cfg1"), ref, cfg1)
            sys.argv = ["inference_grid.py", "--config", os.path.join(ref, "config", "train.yaml"), "--data-path", d1,
                        "--interval", str(cfg1["interval"])]
            runpy.run_path(os.path.join(ref, "inference_grid.py"), run_name="__main__")
            out["cfg1_traj"] = np.load(os.path.join(d1, "grid_trajectories", "grid_trajectories.npy"))
            out["cfg1_occ"] = np.load(os.path.join(d1, "grid_occlusions", "grid_occlusions.npy"))
        d3 = build(os.path.join(tmp, "cfg3"), ref, cfg3)
        pkl = os.path.join(tmp, "tapvid_synth.pkl")
        bench = D.build_tapvid_pickle(pkl, cfg3)
        sys.argv = ["inference_benchmark.py", "--config", os.path.join(ref, "config", "train.yaml"), "--data-path", d3,
                    "--benchmark-pickle-path", pkl, "--video-id", str(cfg3["video_idx"])]
        runpy.run_path(os.path.join(ref, "inference_benchmark.py"), run_name="__main__")
        for f in cfg3["query_frames"]:
            out[f"cfg3_traj_{f}"] = np.load(os.path.join(d3, "trajectories", f"trajectories_{f}.npy"))
            out[f"cfg3_occ_{f}"] = np.load(os.path.join(d3, "occlusions", f"occlusion_preds_{f}.npy"))
        import eval.metrics as EM
        m = EM.compute_tapvid_metrics_for_video(os.path.join(d3, "trajectories"), os.path.join(d3, "occlusions"), bench,
                                                cfg3["video_idx"], pred_video_sizes=[854, 476])
        out["cfg3_metric_names"] = np.array(sorted(m))
        out["cfg3_metric_values"] = np.array([m[k] for k in sorted(m)], dtype=np.float64)
    finally:
        sys.argv = argv0
        os.chdir(cwd)
    return out


def summarise_training(ckpt_dir, start_iter, end_iter):
    """What tests/test_gpu_train.py compares of a finished run: the head's four tensors in full; per Delta-DINO tensor the
    norm of the final value and of the update since the start checkpoint, 64 fixed entries of the update, and the BatchNorm
    running statistics in full (the full state dict is 28 MB)."""
    import torch
    out = {}
    for k, v in torch.load(os.path.join(ckpt_dir, f"tracker_head_{end_iter}.pt"), map_location="cpu").items():
        out["head." + k] = v.numpy()
    a = torch.load(os.path.join(ckpt_dir, f"delta_dino_{start_iter}.pt"), map_location="cpu")
    b = torch.load(os.path.join(ckpt_dir, f"delta_dino_{end_iter}.pt"), map_location="cpu")
    g = np.random.default_rng(0)
    for k in b:
        if k.endswith("filt") or k.endswith("num_batches_tracked"):
            continue
        fin, upd = b[k].double().flatten().numpy(), (b[k].double() - a[k].double()).flatten().numpy()
        if "running" in k:
            out["delta." + k] = fin
            continue
        idx = g.integers(0, fin.size, 64)
        out["delta." + k] = np.concatenate([[np.linalg.norm(fin), np.linalg.norm(upd)], upd[idx]])
    return out


def run_reference_training(synthetic=False):
    """train.py of the reference (BASELINE.json config 5 at reduced size), un-modified, on CPU: three iterations from the
    seeded start checkpoint with every loss term active, through tests/golden/train_driver.py (host-side random draws,
    per-iteration loss capture).  synthetic: on train_data's generated video + masks (a data directory that can be rebuilt
    WITHOUT the reference checkout -- the fixture of the reference-free twin, tests/golden/train_twin.py) instead of the
    reference's horsejump frames."""
    import json
    import subprocess
    import train_data as TD
    ref = ref_harness.REFERENCE_ROOT
    tmp = tempfile.mkdtemp()
    d, cfg = TD.build(os.path.join(tmp, "train"), ref, synthetic_video=synthetic)
    log = os.path.join(tmp, "losses.json")
    env = dict(os.environ, DTK_TRAIN_LOG=log,
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "oracle", "shims"), ref, ROOT]))
    r = subprocess.run([sys.executable, os.path.join(OUT, "train_driver.py"), os.path.join(ref, "train.py"), "--config", cfg,
                        "--data-path", d, "--seed", "2"], env=env, cwd=ref, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    with open(log) as fh:
        rec = json.load(fh)
    out = summarise_training(os.path.join(d, "models", "dino_tracker"), TD.CFG["start_iter"], TD.CFG["total_iterations"])
    out["losses"] = np.array(rec["losses"], dtype=np.float64)
    out["loss_names"] = np.array(rec["names"])
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES) + ["p1_small"]
    if "ref_train" in names:
        names.remove("ref_train")
        res = run_reference_training()
        path = os.path.join(OUT, "ref_train.npz")
        np.savez_compressed(path, **res)
        print("ref_train", res["losses"], os.path.getsize(path) // 1024, "KiB")
        if not names:
            sys.exit(0)
    if "ref_train_synth" in names:
        names.remove("ref_train_synth")
        res = run_reference_training(synthetic=True)
        path = os.path.join(OUT, "ref_train_synth.npz")
        np.savez_compressed(path, **res)
        print("ref_train_synth", res["losses"], os.path.getsize(path) // 1024, "KiB")
        if not names:
            sys.exit(0)
    if "ref_scripts" in names or "ref_scripts_cfg3" in names:  # (not "ref_scripts_synth": handled below)
        only3 = "ref_scripts_cfg3" in names
        names = [n for n in names if not n.startswith("ref_scripts") or n == "ref_scripts_synth"]
        res = run_reference_scripts(only_cfg3=only3)
        path = os.path.join(OUT, "ref_scripts.npz")
        np.savez_compressed(path, **res)
        print("ref_scripts", {k: v.shape for k, v in res.items()}, dict(zip(res["cfg3_metric_names"], res["cfg3_metric_values"])),
              os.path.getsize(path) // 1024, "KiB")
    if "ref_scripts_synth" in names:
        names.remove("ref_scripts_synth")
        res = run_reference_scripts(synthetic=True)
        path = os.path.join(OUT, "ref_scripts_synth.npz")
        np.savez_compressed(path, **res)
        print("ref_scripts_synth", {k: v.shape for k, v in res.items()},
              dict(zip(res["cfg3_metric_names"], res["cfg3_metric_values"])), os.path.getsize(path) // 1024, "KiB")
    if "p1_small" in names:
        names.remove("p1_small")
        res = run_reference_p1()
        path = os.path.join(OUT, "p1_small.npz")
        np.savez_compressed(path, **res)
        print("p1_small", {k: v.shape for k, v in res.items()}, os.path.getsize(path) // 1024, "KiB")
    for name in names:
        res = run_reference(CASES[name])
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **res)
        print(name, {k: v.shape for k, v in res.items()}, "n_anchors", res["n_anchors"].tolist(),
              "occ frac", float(res["occ"].mean()), os.path.getsize(path) // 1024, "KiB")
