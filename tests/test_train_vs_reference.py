"""N1 (test-time training), CPU: the differentiable statement of the path (dino_tracker_amd/train_ops.py) and the
trajectory samplers against the UN-MODIFIED reference modules imported from /root/reference -- values, gradients with
respect to every parameter and input, BatchNorm running statistics, and seeded sampler draws.  Skipped where the reference
is absent (the GPU box); tests/test_gpu_train.py covers the device there against committed goldens."""
import os
import re
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="reference not present")

from dino_tracker_amd import train_ops  # noqa: E402
from dino_tracker_amd.dataset import DinoTrackerSampler, RangeNormalizer  # noqa: E402
from dino_tracker_amd.networks import DeltaDINO, TrackerHead  # noqa: E402


def _randomise(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.2) + (1.0 if "bn" in name else 0.0))
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)


def test_delta_dino_train_mode_matches_reference():
    """Forward value, gradients of all 16 parameter tensors and of the input, running statistics after two steps.  The
    gradient of this chain (four conv + batch-statistics BatchNorm + ReLU stages) is ill-conditioned in float32 -- two
    float32 evaluations with different summation orders differ by ~1e-3 -- so the arbiter is the REFERENCE run in float64:
    this implementation (float32, convs as unfold + GEMM) must be as close to it as the reference's own float32 run is."""
    import copy
    ref = ref_harness.load()
    C, H, W = 32, 70, 98
    theirs = ref.delta_dino.DeltaDINO(channels=[3, 64, 128, 256, C], vit_stride=7)
    _randomise(theirs, 0)
    ours = DeltaDINO(channels=[3, 64, 128, 256, C], vit_stride=7)
    ours.load_state_dict(theirs.state_dict())
    exact = copy.deepcopy(theirs).double()
    for m in (theirs, ours, exact):
        m.train()
    g = torch.Generator().manual_seed(1)
    h, w = (H - 14) // 7 + 1, (W - 14) // 7 + 1
    vit = torch.zeros(3, C, h, w)
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp(min=1e-30))
    for step in range(2):
        x = torch.rand(3, 3, H, W, generator=g)
        cot = torch.randn(3, C, h, w, generator=g)
        res = []
        for m, dt in ((exact, torch.float64), (theirs, torch.float32), (ours, torch.float32)):
            xi = x.to(dt).requires_grad_()
            y = m(xi, vit.to(dt))
            assert y.shape == (3, C, h, w)
            m.zero_grad()
            (y * cot.to(dt)).sum().backward()
            res.append((y.detach(), xi.grad, dict((n, p.grad) for n, p in m.named_parameters())))
        (y_e, dx_e, gp_e), (y_t, dx_t, gp_t), (y_o, dx_o, gp_o) = res
        assert rel(y_o, y_e) <= 3 * rel(y_t, y_e) + 1e-6
        assert rel(dx_o, dx_e) <= 3 * rel(dx_t, dx_e) + 1e-6
        for n in gp_e:
            scale = gp_e[n].abs().max()
            if re.fullmatch(r"layers\.(0|4|8|12)\.bias", n):
                # a conv bias in front of a train-mode BatchNorm has gradient exactly 0 (the batch mean is subtracted):
                # both float32 runs hold rounding noise far below the weight gradient's magnitude
                wscale = gp_e[n.replace("bias", "weight")].abs().max()
                assert scale <= 1e-9 * wscale and gp_t[n].abs().max() <= 1e-3 * wscale and gp_o[n].abs().max() <= 1e-3 * wscale, n
                continue
            e_o = float((gp_o[n].double() - gp_e[n]).abs().max() / scale)
            e_t = float((gp_t[n].double() - gp_e[n]).abs().max() / scale)
            assert e_o <= 3 * e_t + 1e-5, (n, e_o, e_t)
    sd_t, sd_o = theirs.state_dict(), ours.state_dict()
    assert list(sd_t.keys()) == list(sd_o.keys())
    for k in sd_t:
        if "running" in k or "num_batches" in k:
            assert torch.allclose(sd_t[k].float(), sd_o[k].float(), rtol=1e-5, atol=1e-6), k


def test_align_matrices_equal_grid_sample():
    ref = ref_harness.load()
    g = torch.Generator().manual_seed(2)
    for (hc, wc, h, w) in ((60, 107, 67, 121), (9, 13, 9, 13), (16, 107, 17, 121)):
        cnn = torch.randn(2, 5, hc, wc, generator=g)
        want = ref.models_utils.align_cnn_vit_features(torch.zeros(2, 5, h, w), cnn, cnn_stride=8, vit_stride=7)
        got = train_ops.align_cnn_to_vit(cnn, h, w, 7, 14, 8)
        assert (got - want).abs().max() < 5e-6


def test_sampling_correlation_head_match_reference_with_gradients():
    """Tracker.get_point_predictions (tracker.py:171-180) of the reference, assembled from its own methods, against
    train_ops: predictions and gradients with respect to the frame embeddings and the four head tensors -- including maps
    that take the zero-mass fallback branch of the soft arg-max."""
    ref = ref_harness.load()
    H, W, C, n, B = 126, 210, 24, 3, 40
    h, w = (H - 14) // 7 + 1, (W - 14) // 7 + 1
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)  # the heads' default initialisation draws from the global generator
    head_t = ref.tracker_head.TrackerHead(patch_size=14, step_h=7, step_w=7, video_h=H, video_w=W)
    head_o = TrackerHead(patch_size=14, step_h=7, step_w=7, video_h=H, video_w=W)
    head_o.load_state_dict(head_t.state_dict())
    head_o.train()
    head_t.train()

    class Stub:  # the attributes the reference's Tracker methods read
        video = torch.zeros(n, 3, H, W)
        dino_patch_size, stride, device = 14, 7, "cpu"
        tracker_head = head_t
        cmap_relu = torch.nn.ReLU()
    T = ref.tracker.Tracker
    for name in ("normalize_points_for_sampling", "sample_embeddings", "get_corr_maps_for_frame_set",
                 "get_point_predictions_from_embeddings", "get_point_predictions"):
        setattr(Stub, name, getattr(T, name))
    stub = Stub()

    for scale in (1.0, 300.0):  # second round: logits that peak AWAY from the correlation peak -> fallback branch
        with torch.no_grad():
            head_t.cnn_refiner[2].bias.fill_(0.0)
        emb_t = (torch.randn(n, C, h, w, generator=g)).requires_grad_()
        emb_o = emb_t.detach().clone().requires_grad_()
        pts = torch.stack([torch.rand(B, generator=g) * (W - 1), torch.rand(B, generator=g) * (H - 1),
                           torch.zeros(B)], dim=1)
        src = torch.randint(n, (B,), generator=g)
        tgt = torch.randint(n, (B,), generator=g)
        if scale > 1:
            with torch.no_grad():
                for hd in (head_t, head_o):
                    # (the normalisation W / sum W is scale invariant: large logits need taps of mixed sign)
                    # hidden channel 0 = the map itself; its output filter = scale x (cell - right neighbour) + neighbour:
                    # a steep horizontal-gradient detector whose maximum usually lies outside the peak's disk
                    hd.cnn_refiner[0].weight[0].zero_()
                    hd.cnn_refiner[0].weight[0, 0, 1, 1] = 1.0
                    hd.cnn_refiner[0].bias[0] = 0.0
                    hd.cnn_refiner[2].weight[0, 0].zero_()
                    hd.cnn_refiner[2].weight[0, 0, 1, 1] = scale
                    hd.cnn_refiner[2].weight[0, 0, 1, 2] = 1.0 - scale
        inp = (pts, src, tgt, torch.arange(n))
        out_t = stub.get_point_predictions(inp, emb_t)
        nrm = stub.normalize_points_for_sampling(pts)
        s_o = train_ops.sample_bilinear(emb_o, torch.cat([nrm[:, :2], src[:, None].float()], dim=1))
        out_o = train_ops.head_forward(head_o, torch.relu(train_ops.cosine_maps(s_o, emb_o, tgt))[:, None])
        assert (out_o - out_t).abs().max() < 2e-5
        cot = torch.randn(out_t.shape, generator=g)
        head_t.zero_grad()
        head_o.zero_grad()
        (out_t * cot).sum().backward()
        (out_o * cot).sum().backward()
        assert (emb_o.grad - emb_t.grad).abs().max() <= 2e-3 * emb_t.grad.abs().max()
        for (nm, p_t), (_, p_o) in zip(head_t.named_parameters(), head_o.named_parameters()):
            assert (p_o.grad - p_t.grad).abs().max() <= 2e-3 * p_t.grad.abs().max() + 1e-6, nm
    # the second configuration did exercise the fallback
    with torch.no_grad():
        cost = torch.relu(train_ops.cosine_maps(s_o, emb_o, tgt))[:, None]
        p = torch.softmax(train_ops.head_logits(head_o, cost).reshape(B, -1), dim=1).reshape(B, h, w)
        peak = cost[:, 0].reshape(B, -1).argmax(dim=1)
        ys = torch.arange(h) * 7.0 + 7
        xs = torch.arange(w) * 7.0 + 7
        d = torch.sqrt((ys[None, :, None] - (peak // w * 7.0 + 7)[:, None, None]) ** 2 +
                       (xs[None, None, :] - (peak % w * 7.0 + 7)[:, None, None]) ** 2)
        assert ((p * (d <= 35)).sum(dim=(1, 2)) < 1e-8).any()


def test_samplers_draw_what_the_reference_draws():
    ref = ref_harness.load()
    g = torch.Generator().manual_seed(4)
    T, W, H = 12, 210, 126

    def trajectories(n):
        tr = torch.rand(n, T, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
        tr[torch.rand(n, T, generator=g) < 0.45] = float("nan")
        return tr
    fg, bg = trajectories(300), trajectories(500)
    rn_t = ref.dataset.RangeNormalizer(shapes=(W, H, T))
    rn_o = RangeNormalizer(shapes=(W, H, T))
    kw = dict(batch_size=64, dst_range=(-1, 1), fg_traj_ratio=0.5, num_frames=4, keep_in_cpu=False)
    theirs = ref.dataset.DinoTrackerSampler(range_normalizer=rn_t, fg_trajectories=fg.clone(), bg_trajectories=bg.clone(), **kw)
    ours = DinoTrackerSampler(range_normalizer=rn_o, fg_trajectories=fg.clone(), bg_trajectories=bg.clone(), **kw)
    for seed in range(5):
        torch.manual_seed(seed)
        a = theirs()
        torch.manual_seed(seed)
        b = ours()
        assert a.keys() == b.keys()
        for k in a:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
            assert torch.equal(a[k], b[k]), k
        assert b["frames_set_t"].numel() <= 8 and b["t1_points"].shape == (64, 3)
        assert not torch.isnan(b["t1_points"]).any() and not torch.isnan(b["t2_points_normalized"]).any()


def test_restated_train_yaml_matches_reference():
    """tests/golden/train_data.TRAIN_YAML (what the reference-free data directory is configured from) == config/train.yaml."""
    import yaml
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import train_data as TD
    with open(os.path.join(ref_harness.REFERENCE_ROOT, "config", "train.yaml")) as fh:
        assert yaml.safe_load(fh.read()) == TD.TRAIN_YAML


@pytest.mark.skipif(not os.environ.get("DTK_SLOW_CPU_TESTS"), reason="~4 min of reference training on CPU: set DTK_SLOW_CPU_TESTS=1 "
                    "(measured in round 4: all 21 losses within 2.6e-6 of tests/golden/ref_train_synth.npz)")
def test_train_twin_pinned_against_reference_on_cpu(tmp_path):
    """tests/golden/train_twin.py (the reference-free twin of train.py that the driver's GPU box runs) around the REFERENCE's own
    Tracker on CPU: its control plane, its order of random draws and the loss terms it evaluates reproduce the golden run of the
    un-modified train.py (ref_train_synth.npz) -- so what the GPU test adds to the comparison is this implementation's models
    and kernels, nothing else."""
    import json
    import subprocess
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import train_data as TD
    d, cfg = TD.build(str(tmp_path / "train"), None, synthetic_video=True)
    log = str(tmp_path / "l.json")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "oracle", "shims"), ref_harness.REFERENCE_ROOT, root]),
               DTK_TWIN_MODEL="reference")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "train_twin.py"), "--config", cfg, "--data-path", d,
                        "--seed", "2", "--log", log, "--device", "cpu"], capture_output=True, text=True, env=env,
                       cwd=ref_harness.REFERENCE_ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.array(json.load(open(log))["losses"])
    gold = np.load(os.path.join(root, "tests", "golden", "ref_train_synth.npz"))["losses"]
    assert (np.abs(got - gold) / np.maximum(np.abs(gold), 1e-6)).max() < 2e-5
