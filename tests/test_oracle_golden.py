"""CPU: the oracle (oracle/ref_algo.py) reproduces the golden fixtures written from the UN-MODIFIED reference
(tests/golden/make_golden.py).  Tolerances: 1e-3 px on positions (north_star), identical occlusion flags."""
import os

import numpy as np
import pytest
import torch

import make_golden as MG
from oracle import ref_algo as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"))


@pytest.mark.parametrize("name", list(MG.CASES))
def test_infer_matches_reference_golden(name):
    cfg = MG.CASES[name]
    g = load(name)
    video, dino, head, queries, delta = MG.build_inputs(cfg)
    assert np.array_equal(queries.numpy(), g["queries"])
    feats = dino if delta is None else A.refine_features(video, dino, delta)
    if delta is not None:
        assert np.abs(feats.numpy() - g["refined"]).max() < 1e-5
    traj, occ, cs, greens = A.infer(feats, queries, head, cfg["H"], cfg["W"], return_aux=True)
    assert np.abs(traj.numpy() - g["traj"]).max() < 1e-3
    assert np.array_equal(occ.numpy(), g["occ"])
    assert np.abs(cs.numpy() - g["cos_sims"]).max() < 1e-5
    assert [x.shape[0] for x in greens] == g["n_anchors"].tolist()
    assert np.abs(greens[0].numpy() - g["anchor0"]).max() < 1e-3


@pytest.mark.parametrize("name", ["p3_small", "p3_small_wild", "p3_full"])
def test_head_matches_reference_golden(name):
    cfg = MG.CASES[name]
    g = load(name)
    head = MG.build_inputs(cfg)[2]
    out = A.tracker_head(torch.relu(torch.from_numpy(g["head_maps"][:, 0])), head, cfg["H"], cfg["W"])
    assert np.abs(out.numpy() - g["head_out"]).max() < 1e-6
    # all-zero map: argmax index 0, peak pixel (7,7)
    _, k, _, _ = A.tracker_head(torch.zeros(1, *g["head_maps"].shape[-2:]), head, cfg["H"], cfg["W"], return_aux=True)
    assert int(k[0]) == 0


@pytest.mark.parametrize("name", ["p3_small", "p3_full"])
def test_forward_api_golden(name):
    """Tracker.forward for (query 0 -> every frame) returns normalised coords; same numbers as traj[0]."""
    cfg = MG.CASES[name]
    g = load(name)
    fx = (g["fwd_q0"][:, 0] + 1) / 2 * (cfg["W"] - 1)
    fy = (g["fwd_q0"][:, 1] + 1) / 2 * (cfg["H"] - 1)
    assert np.abs(fx - g["traj"][0, :, 0]).max() < 1e-3 and np.abs(fy - g["traj"][0, :, 1]).max() < 1e-3


def test_sampling_literal_vs_algorithmic():
    torch.manual_seed(0)
    emb = torch.randn(5, 8, 19, 29)
    pts = torch.rand(40, 3) * torch.tensor([230.0, 160.0, 0.0]) - torch.tensor([10.0, 10.0, 0.0])
    pts[:, 2] = torch.randint(0, 5, (40,)).float()
    lit = A.sample_embeddings_literal(emb, A.points_to_grid_coords(pts, 140, 210))
    alg = A.sample_bilinear(emb, pts[:, :2], pts[:, 2], 140, 210)
    assert (lit - alg).abs().max() < 2e-5


def test_vit_oracle_matches_reference_extractor_golden():
    """p1_small.npz: the UN-MODIFIED reference VitExtractor / get_dino_features_video (models/extractor.py:23-150,
    utils.py:33-72) around the DINOv2-API stub.  Pins rows a1-a4 of the oracle: position-encoding interpolation, patch
    embedding at stride 7, hooked block outputs, layer mean incl. CLS, qkv record, key facet, key self-similarity."""
    g = load("p1_small")
    video = MG.p1_video()
    name = MG.P1_CASE["model"]
    for tag, ls in (("ls1", 1.0), ("ls01", 0.1)):
        sd = MG.p1_weights(ls)
        gold = g[f"tokens_{tag}_l11"]
        for t in range(gold.shape[0]):
            got = A.vit_tokens(video[t:t + 1], sd, name, layer=None)
            assert np.abs(got.numpy() - gold[t]).max() < 2e-5 * max(1.0, np.abs(gold[t]).max())
    sd = MG.p1_weights(1.0)
    assert np.abs(A.vit_tokens(video[:1], sd, name, layer=3).numpy() - g["tokens_ls1_l3"][0]).max() < 2e-5
    cls = torch.stack([A.vit_all_tokens(video[:1], sd, name, layer=l) for l in (2, 5)]).mean(0)
    assert np.abs(cls.numpy() - g["feature_with_cls_l2_l5"]).max() < 2e-5
    qkv = A.vit_qkv(video[:1], sd, name, layer=1)
    assert np.abs(qkv.numpy() - g["qkv_l1"]).max() < 2e-5
    k3 = A.vit_qkv(video[:1], sd, name, layer=3)[0, 1:, 384:768].reshape(13, 17, 384).permute(2, 0, 1)
    assert np.abs(k3.numpy() - g["keys_l3"][0]).max() < 2e-5
