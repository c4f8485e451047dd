"""-m gpu: parity on the HEADLINE configuration (BASELINE.json metric: 854 x 480 x 90, 1024 queries) -- every one of the 92 160
positions and flags, not a sample (VERDICT r4 "missing" #1; the reference's contract is all N queries,
models/model_inference.py:203-216).

What makes it affordable: the oracle (oracle/ref_algo.py, plain torch) takes its device from its inputs, so the 8.4 M correlation
maps of the full configuration are evaluated ON THE GPU in fp32 (no TF32 exists on gfx950; matmul precision pinned to "highest")
in seconds.  The device form of the oracle is pinned against its CPU form -- the one pinned on the un-modified reference -- first."""
import json
import os
import sys

import pytest
import torch

from dino_tracker_amd import synth
from oracle import ref_algo as A

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 476, 854


@pytest.fixture(autouse=True)
def _fp32_matmul():
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    torch.set_float32_matmul_precision("highest")


def _cuda(d):
    return {k: v.cuda() for k, v in d.items()}


def test_oracle_device_form_matches_cpu_form():
    """GPU-torch against CPU-torch: the same restatement on `cuda` tensors and on host tensors -- P3 (infer incl. anchors and
    occlusion) at the full 67 x 121 grid, C = 384; Delta-DINO and the ViT encoder on full-resolution frames."""
    T, C = 8, 384
    feats = synth.synth_features(T, C, 67, 121, seed=41)
    head = synth.synth_head_weights(3)
    queries = torch.cat([synth.grid_queries(4, 3, H, W, 0), synth.grid_queries(2, 2, H, W, 3)])
    rt, ro, rcs, rg = A.infer(feats, queries, head, H, W, return_aux=True)
    gt, go, gcs, gg = A.infer(feats.cuda(), queries.cuda(), _cuda(head), H, W, return_aux=True)
    assert (gt.cpu() - rt).abs().max() < 5e-4, (gt.cpu() - rt).abs().max()      # a few fp32 ulps of a coordinate ~ 800 (ulp 6.1e-5; measured: up to 5)
    assert (gcs.cpu() - rcs).abs().max() < 5e-6
    assert torch.equal(go.cpu(), ro)
    # anchor trajectories (16 queries x 8 anchors x 8 frames): to a few fp32 ulps, except where a map's two best cells tie within
    # fp32 rounding and the two evaluation orders pick differently -- each such position is ARBITRATED (float64 map, fp32 band only)
    flagged = 0
    for i, (a, b) in enumerate(zip(gg, rg)):
        anchors = torch.nonzero(rcs[i] >= 0.7)[:, 0]
        d = (a.cpu() - b).norm(dim=-1)                                   # [A, T]
        for ai, t in torch.nonzero(d > 5e-4).tolist():
            flagged += 1
            src_xyt = torch.cat([rt[i, t], torch.tensor([float(t)])])    # the source is the embedding at traj[i, t] in frame t
            r = A.tie_arbiter(feats, src_xyt, int(anchors[ai]), a[ai, t].cpu(), head, H, W, A.fp32_dot_band(C))
            assert r["ok"] and r["gap64"] <= r["delta"], (i, ai, t, float(d[ai, t]), r)
    print("anchor tracks cuda vs cpu:", sum(int(b.shape[0] * b.shape[1]) for b in rg), "positions,", flagged, "arbitrated")
    assert flagged <= 64
    # P1 / P2 on two full-resolution frames
    video = synth.synth_video(2, H, W, seed=2000)
    sd = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1)
    delta = synth.synth_delta_dino_weights(C, seed=4)
    dino = torch.stack([A.vit_tokens(video[t:t + 1], sd, "dinov2_vits14") for t in range(2)])
    dino_g = torch.stack([A.vit_tokens(video[t:t + 1].cuda(), _cuda(sd), "dinov2_vits14") for t in range(2)])
    rel1 = float((dino_g.cpu() - dino).norm() / dino.norm())
    ref = A.refine_features(video, dino, delta)
    ref_g = A.refine_features(video.cuda(), dino.cuda(), _cuda(delta))
    rel2 = float((ref_g.cpu() - ref).norm() / ref.norm())
    print("oracle cuda vs cpu: ViT rel", rel1, "refine rel", rel2)
    assert rel1 < 5e-6 and rel2 < 2e-6, (rel1, rel2)    # fp32 summation order only (the fp16-operand device path sits at 1.3e-4)


def test_headline_configuration_every_query():
    """854 x 476 x 90, 1024 grid queries, benchmark weights -- the configuration BENCH quotes:
      (b) from the VIDEO: HIP ViT -> HIP Delta-DINO -> HIP infer against oracle ViT -> oracle refine -> oracle infer (all on the
          GPU in fp32): every position within 1e-3 px or arbitrated by the float64 tie arbiter with its TIGHT band (fp32 rounding
          + the measured deviation of the two cosines in question), flags identical for every query without an arbitrated point;
      (a) on IDENTICAL features (the device's refined volume through the oracle): every position within 1e-3 px unless it is an
          fp32 near-tie (band: fp32 rounding alone, 2.3e-6), flags identical likewise."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import e2e_error
    r = e2e_error.run(H, W, 90, 32, oracle_device="cuda")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "e2e_error_fullsize_476x854x90_1024q.json"), "w") as fh:
        json.dump(r, fh, indent=1)
    px, arb, same = r["px_err_vs_oracle_on_same_video"], r["arbitrated"], r["px_err_vs_oracle_on_same_features"]
    print("headline configuration, every query:", json.dumps(px), "beyond 1e-3 px:", r["points_beyond_1e-3px"],
          "same features:", json.dumps({k: v for k, v in same.items() if k != "arbitrated"}),
          "flags", r["occ_mismatch_same_video"], r["occ_mismatch_same_features"], "oracle s", r["oracle_seconds"])
    assert r["argmax_margin"]["points"] == 92160 and r["occ_total"] == 92160
    assert r["feature_rel_err_P1"] < 3e-4 and r["feature_rel_err_refined"] < 3e-4
    # (a) identical features
    assert same["arbitration_failures"] == 0, same["arbitrated"]
    assert all(a["gap64"] <= a["delta"] <= 3e-6 and a["dist_px"] <= 1e-3 for a in same["arbitrated"]), same["arbitrated"]
    assert len(same["arbitrated"]) <= 16, len(same["arbitrated"])
    assert same["p99"] <= 1e-3
    assert r["occ_mismatch_same_features_queries_without_a_tie"] == 0
    # (b) from the video
    assert r["arbitration_failures"] == 0, arb
    assert all(a["gap64"] <= a["delta"] and a["dist_px"] <= 1e-3 for a in arb), arb
    assert len(arb) <= 92, len(arb)                 # (T = 16 / 256 queries: 1 of 4 096; the budget here is 1 per 1 000)
    assert px["p99"] <= 1e-3, px
    assert r["occ_mismatch_same_video_queries_without_a_tie"] == 0
