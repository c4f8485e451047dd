"""The oracle's DEVICE form (oracle/ref_algo.py `_plain`: unfold + matmul convolutions, explicit BatchNorm / BlurPool / attention --
what runs when the restatement is given CUDA tensors, so that no vendor convolution / attention library decides a result) against its
CPU form (the one pinned on the un-modified reference), here on the CPU by forcing the switch.  tests/test_gpu_fullsize.py repeats the
comparison on the GPU box; this file keeps the algebra pinned where no GPU exists."""
import pytest
import torch

from dino_tracker_amd import synth
from oracle import ref_algo as A


@pytest.fixture
def plain(monkeypatch):
    def on():
        monkeypatch.setattr(A, "_plain", lambda x: True)

    def off():
        monkeypatch.setattr(A, "_plain", lambda x: False)
    return on, off


def test_infer_plain_form_matches(plain):
    on, off = plain
    H, W, T, C = 140, 210, 5, 48
    feats = synth.synth_features(T, C, 19, 29, seed=11)
    head = synth.synth_head_weights(3)
    queries = torch.cat([synth.grid_queries(3, 3, H, W, 0, margin=20.0), synth.grid_queries(2, 2, H, W, 2, margin=20.0)])
    off()
    rt, ro, rcs, rg = A.infer(feats, queries, head, H, W, return_aux=True)
    on()
    pt, po, pcs, pg = A.infer(feats, queries, head, H, W, return_aux=True)
    assert (pt - rt).abs().max() < 2e-4 and torch.equal(po, ro) and (pcs - rcs).abs().max() < 5e-6
    assert all((a - b).abs().max() < 2e-4 for a, b in zip(pg, rg))   # (a few fp32 ulps of a coordinate ~ 200)
    # the table form of the anchor stage (src_row) == materialised sources
    src = torch.randn(7, C)
    tgt = torch.tensor([0, 3, 3, 1, 4, 0, 2])
    row = torch.tensor([6, 0, 0, 5, 2, 2, 1])
    off()
    assert torch.equal(A.track(src, feats, tgt, head, H, W, src_row=row), A.track(src[row], feats, tgt, head, H, W))


def test_delta_dino_and_vit_plain_form_match(plain):
    on, off = plain
    H, W, T, C = 98, 126, 2, 384
    video = synth.synth_video(T, H, W, seed=5)
    delta = synth.synth_delta_dino_weights(C, seed=4)
    sd = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1)
    off()
    dino = torch.stack([A.vit_tokens(video[t:t + 1], sd, "dinov2_vits14", layer=2) for t in range(T)])
    ref = A.refine_features(video, dino, delta)
    on()
    dino_p = torch.stack([A.vit_tokens(video[t:t + 1], sd, "dinov2_vits14", layer=2) for t in range(T)])
    ref_p = A.refine_features(video, dino, delta)
    assert ((dino_p - dino).norm() / dino.norm()) < 2e-6
    assert ((ref_p - ref).norm() / ref.norm()) < 2e-6
    assert (ref_p - ref).abs().max() < 1e-4 * ref.abs().max()
