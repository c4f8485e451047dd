"""N1, CPU, no reference needed: the differentiable pieces of dino_tracker_amd/train_ops.py checked against torch's own
operators and by finite differences in float64 (torch.autograd.gradcheck) -- the host forms the device path shares its
arithmetic with (conv as unfold + GEMM, blur-pool as strided views, the alignment matrices, four-corner sampling, grouped
cosine maps, the head as two matrix products and a shift-add)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dino_tracker_amd import train_ops  # noqa: E402
from dino_tracker_amd.networks import TrackerHead  # noqa: E402


def test_conv2d_gemm_equals_conv2d_and_its_gradients():
    g = torch.Generator().manual_seed(0)
    for cin, cout, k, pad, dil, mode in ((3, 5, 5, 2, 1, "reflect"), (4, 6, 5, 4, 2, "reflect"), (1, 16, 3, 1, 1, "zeros")):
        x = torch.randn(2, cin, 11, 13, dtype=torch.float64, generator=g, requires_grad=True)
        w = torch.randn(cout, cin, k, k, dtype=torch.float64, generator=g, requires_grad=True)
        b = torch.randn(cout, dtype=torch.float64, generator=g, requires_grad=True)
        got = train_ops.conv2d_gemm(x, w, b, pad, dil, mode)
        xp = F.pad(x, (pad,) * 4, mode=mode) if mode != "zeros" else x
        want = F.conv2d(xp, w, b, padding=0 if mode != "zeros" else pad, dilation=dil)
        assert torch.allclose(got, want, atol=1e-12)
        assert torch.autograd.gradcheck(lambda a, c, d: train_ops.conv2d_gemm(a, c, d, pad, dil, mode), (x, w, b), atol=1e-7)
        # the custom Function of the device path (persistent scratch, unfold recomputed in the backward) is plain torch too
        got2 = train_ops._ConvGemm.apply(x, w, pad, dil, mode) + b[None, :, None, None]
        assert torch.allclose(got2, want, atol=1e-12)
        assert torch.autograd.gradcheck(lambda a, c: train_ops._ConvGemm.apply(a, c, pad, dil, mode), (x, w), atol=1e-7)
    train_ops.release_scratch()


def test_blurpool_views_equal_the_depthwise_convolution():
    g = torch.Generator().manual_seed(1)
    a = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float64)
    for shape in ((2, 3, 9, 12), (1, 2, 4, 4), (1, 1, 7, 5)):
        x = torch.randn(shape, dtype=torch.float64, generator=g, requires_grad=True)
        filt = (a[:, None] * a[None, :] / 64.0)[None, None].repeat(shape[1], 1, 1, 1)
        want = F.conv2d(F.pad(x, (1, 2, 1, 2), mode="reflect"), filt, stride=2, groups=shape[1])
        assert torch.allclose(train_ops.blurpool(x, filt), want, atol=1e-13)
        assert torch.autograd.gradcheck(lambda t: train_ops.blurpool(t, filt), (x,), atol=1e-7)


def test_alignment_matrices_are_bilinear_border_clamped_rows():
    for n_vit, n_cnn in ((67, 60), (121, 107), (9, 9), (17, 16)):
        m = train_ops.align_matrix(n_vit, n_cnn, 7, 14, 8, "cpu", torch.float64)
        assert torch.allclose(m.sum(dim=1), torch.ones(n_vit, dtype=torch.float64), atol=1e-6)
        assert (m >= 0).all() and ((m > 0).sum(dim=1) <= 2).all()
        pos = ((torch.arange(n_vit, dtype=torch.float64) * 7 + 7) - 0.5) / 8  # CNN index of ViT centre i
        centre = (m * torch.arange(n_cnn, dtype=torch.float64)[None]).sum(dim=1)
        assert torch.allclose(centre, pos.clamp(0, n_cnn - 1), atol=1e-4)


def test_sampling_and_cosine_maps_gradcheck():
    g = torch.Generator().manual_seed(2)
    emb = torch.randn(3, 4, 5, 6, dtype=torch.float64, generator=g, requires_grad=True)
    pts = torch.stack([torch.rand(7, dtype=torch.float64, generator=g) * 2.4 - 1.2,   # some outside [-1, 1]: border clamp
                       torch.rand(7, dtype=torch.float64, generator=g) * 2.4 - 1.2,
                       torch.randint(3, (7,), generator=g).double()], dim=1)
    want = F.grid_sample(emb.permute(1, 0, 2, 3)[None], torch.cat([pts[:, :2], pts[:, 2:] - 1.0], dim=1)[None, None, :, None],
                         align_corners=True, padding_mode="border")[0, :, 0, :, 0].t()
    assert torch.allclose(train_ops.sample_bilinear(emb, pts), want, atol=1e-12)
    assert torch.autograd.gradcheck(lambda e: train_ops.sample_bilinear(e, pts), (emb,), atol=1e-7)
    src = torch.randn(7, 4, dtype=torch.float64, generator=g, requires_grad=True)
    tgt = torch.tensor([0, 2, 2, 1, 0, 2, 1])
    maps = train_ops.cosine_maps(src, emb, tgt)
    for b in range(7):
        want_b = F.cosine_similarity(src[b][:, None, None], emb[tgt[b]], dim=0, eps=1e-8)
        assert torch.allclose(maps[b], want_b, atol=1e-12)
    assert torch.autograd.gradcheck(lambda s_, e_: train_ops.cosine_maps(s_, e_, tgt), (src, emb), atol=1e-7)


def test_head_forward_equals_convs_softmax_softargmax_and_gradcheck():
    torch.manual_seed(3)
    H, W = 70, 98
    head = TrackerHead(patch_size=14, step_h=7, step_w=7, video_h=H, video_w=W).double().train()
    h, w = (H - 14) // 7 + 1, (W - 14) // 7 + 1
    cost = torch.rand(3, 1, h, w, dtype=torch.float64)
    c0, c2 = head.cnn_refiner[0], head.cnn_refiner[2]
    z_want = F.conv2d(torch.relu(F.conv2d(cost, train_ops.normalized_weight(c0.weight), c0.bias, padding=1)),
                      train_ops.normalized_weight(c2.weight), c2.bias, padding=1)
    assert torch.allclose(train_ops.head_logits(head, cost), z_want, atol=1e-12)
    out = train_ops.head_forward(head, cost)
    assert out.shape == (3, 2) and (out.abs() <= 1.0 + 1e-9).all()
    params = [c0.weight, c0.bias, c2.weight, c2.bias]
    assert torch.autograd.gradcheck(lambda *_: train_ops.head_forward(head, cost), params, atol=1e-6, nondet_tol=0)
    # the last bias shifts every logit alike: exactly zero gradient
    (g_b2,) = torch.autograd.grad(train_ops.head_forward(head, cost).sum(), [c2.bias])
    assert g_b2.abs().max() < 1e-12


def test_gradient_sink_equals_ordinary_accumulation():
    """train_ops.attach_grad_sink: consumers that accumulate into the shared buffer (here two bilinear point reads) plus a dense
    consumer give the gradient autograd's ordinary accumulation gives; also with sparse consumers only, and twice in a row (the
    buffer is handed over and reset by the sink node)."""
    torch.manual_seed(0)
    emb = torch.randn(3, 5, 6, 7, dtype=torch.double, requires_grad=True)
    pts = torch.rand(11, 3, dtype=torch.double) * 2 - 1
    pts[:, 2] = torch.randint(0, 3, (11,)).double()
    w = torch.randn(11, 5, dtype=torch.double)
    plain = lambda e: (train_ops.sample_bilinear(e, pts) * w).sum() + (train_ops.sample_bilinear(e, pts.flip(0)) * w).sum()
    (want,) = torch.autograd.grad(plain(emb) + (emb ** 2).sum(), emb)
    (want_sparse,) = torch.autograd.grad(plain(emb), emb)
    for _ in range(2):
        es = train_ops.attach_grad_sink(emb)
        (got,) = torch.autograd.grad(plain(es) + (es ** 2).sum(), emb)
        assert (got - want).abs().max() < 1e-12
        es = train_ops.attach_grad_sink(emb)
        (got,) = torch.autograd.grad(plain(es), emb)
        assert (got - want_sparse).abs().max() < 1e-12
