"""-m gpu: DINOv2 ViT encoder (P1) through the C-ABI vs the fp32 oracle restatement (seeded random weights).
The device path uses bf16 matrix operands (fp32 accumulate / residual stream), the oracle fp32, so parity is stated
at feature level: per-token cosine similarity >= 0.999 and relative Frobenius error <= 2e-2."""
import pytest
import torch

from dino_tracker_amd import synth
from dino_tracker_amd.extractor import VitExtractor
from oracle import ref_algo as A

pytestmark = pytest.mark.gpu


def _check(got, ref, cos_min=0.999, rel_max=2e-2):
    got, ref = got.float().cpu(), ref.float().cpu()
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=-1)
    rel = (got - ref).norm() / ref.norm()
    assert cos.min() >= cos_min, (cos.min().item(), rel.item())
    assert rel <= rel_max, (cos.min().item(), rel.item())
    return cos.min().item(), rel.item()


@pytest.fixture(scope="module")
def vits():
    sd = synth.make_vit_weights("dinov2_vits14", seed=2)
    return sd, VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd)


@pytest.mark.parametrize("layer", [0, 3, 11])
def test_small_frames_all_depths(vits, layer):
    sd, ex = vits
    video = synth.synth_video(3, 140, 210, seed=71)  # 19 x 29 tokens; 3 frames in one batch
    feat = ex.encode(video, layer=layer)
    assert feat.shape == (3, 19 * 29, 384)
    for t in range(3):
        ref = A.vit_tokens(video[t:t + 1], sd, "dinov2_vits14", layer=layer).permute(1, 2, 0).reshape(-1, 384)
        _check(feat[t], ref)


@pytest.mark.parametrize("shape", [(3, 140, 210), (1, 98, 126), (2, 476, 854)])
def test_patch_embedding_is_fp32_grade(vits, shape):
    """layer = -1: tokens = patch embedding + position encoding, no block.  The split-fp16 MFMA kernel (hi + lo halves,
    three products; used when its scratch fits the workspace: the first and third shape) and the f32-input MFMA
    fallback (second shape) must both agree with an fp32 convolution to fp32 rounding level, not to bf16 level."""
    sd, ex = vits
    n, h, w = shape
    video = synth.synth_video(n, h, w, seed=77)
    feat = ex.encode(video, layer=-1)
    for t in range(n):
        ref = A.vit_tokens(video[t:t + 1].double(), {k: v.double() for k, v in sd.items()}, "dinov2_vits14",
                           layer=-1).permute(1, 2, 0).reshape(-1, 384)
        got = feat[t].double().cpu()
        rel = ((got - ref).norm() / ref.norm()).item()
        assert rel <= 2e-6, rel
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_weight_stationary_gemm_matches_tiled_gemm():
    """The K = 384 GEMMs (QKV, projection, fc1) run on the weight-stationary kernel; dtk_vit_model.flags =
    DTK_VIT_TILED_GEMMS sends them to the tiled kernel instead.  Both use bf16 operands, so after all 12 blocks the
    features must agree far more closely than either agrees with the fp32 oracle (differences: accumulation order, erf
    polynomial vs erff).  Run with the benchmark's LayerScale (0.1)."""
    sd = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1)
    ex = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd)
    video = synth.synth_video(2, 140, 210, seed=78)
    a = ex.encode(video).cpu()
    ex.tiled_gemms = True
    b = ex.encode(video).cpu()
    assert not torch.equal(a, b)  # two different kernels did run
    _check(a, b, cos_min=0.9999, rel_max=1e-2)


def test_full_resolution_first_blocks(vits):
    """476 x 854 -> 8107 patch tokens + CLS (not a multiple of the 64-key tile: exercises the masked tail)."""
    sd, ex = vits
    video = synth.synth_video(1, 476, 854, seed=72)
    feat = ex.encode(video, layer=1)
    ref = A.vit_tokens(video, sd, "dinov2_vits14", layer=1).permute(1, 2, 0).reshape(-1, 384)
    _check(feat[0], ref)


def test_get_feature_from_input_includes_cls(vits):
    sd, ex = vits
    video = synth.synth_video(1, 98, 126, seed=73)
    m = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1)
    s = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1)
    tok = ex.get_feature_from_input(((video - m) / s).cuda(), layers=[2])
    assert tok.shape == (1, 1 + 13 * 17, 384)
    ref = A.vit_tokens(video, sd, "dinov2_vits14", layer=2).permute(1, 2, 0).reshape(-1, 384)
    _check(tok[0, 1:], ref)
    assert torch.isfinite(tok[0, 0]).all() and tok[0, 0].abs().sum() > 0


def test_get_dino_features_video_contract(vits):
    from dino_tracker_amd.utils import get_dino_features_video
    sd, _ = vits
    video = synth.synth_video(2, 98, 126, seed=74)
    out = get_dino_features_video(video.cuda(), model_name="dinov2_vits14", facet="tokens", stride=7, layer=1,
                                  device="cuda:0", state_dict=sd)
    assert out.device.type == "cpu" and out.shape == (2, 384, 13, 17)  # utils.py:53,67: T x C x ph x pw on the CPU
    ref = A.vit_tokens(video[1:2], sd, "dinov2_vits14", layer=1)
    _check(out[1].permute(1, 2, 0).reshape(-1, 384), ref.permute(1, 2, 0).reshape(-1, 384))


def test_vitb_width(vits):
    """ViT-B/14 (D = 768, 12 heads) shares the kernels."""
    sd = synth.make_vit_weights("dinov2_vitb14", seed=5)
    ex = VitExtractor("dinov2_vitb14", stride=7, device="cuda:0", state_dict=sd)
    video = synth.synth_video(1, 98, 126, seed=75)
    feat = ex.encode(video, layer=1)
    ref = A.vit_tokens(video, sd, "dinov2_vitb14", layer=1).permute(1, 2, 0).reshape(-1, 768)
    _check(feat[0], ref)


def test_attention_large_logits_force_rescale():
    """Online-softmax deferred-maximum branch: with 6x larger q/k projections the scores spread over tens of log2
    units, so the running maximum is overtaken by more than 8 many times along the key sweep (with the default weights
    the rescale only happens on the first tile).  The first block's output must still match the fp32 oracle."""
    sd = {k: v.clone() for k, v in synth.make_vit_weights("dinov2_vits14", seed=9).items()}
    d = 384
    sd["blocks.0.attn.qkv.weight"][:2 * d] *= 6.0  # q and k rows
    ex = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd)
    video = synth.synth_video(1, 238, 322, seed=76)  # 33 x 45 = 1485 tokens: 24 key tiles
    feat = ex.encode(video, layer=0)
    ref = A.vit_tokens(video, sd, "dinov2_vits14", layer=0).permute(1, 2, 0).reshape(-1, d)
    _check(feat[0], ref, cos_min=0.998, rel_max=3e-2)


# ---- against the un-modified reference extractor (tests/golden/p1_small.npz, written by make_golden.py) -----------------
def _gold():
    import os
    import numpy as np
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "p1_small.npz"))


@pytest.mark.parametrize("tag,ls", [("ls1", 1.0), ("ls01", 0.1)])
def test_tokens_match_reference_extractor_golden(tag, ls):
    """utils.get_dino_features_video (reference signature) on the device vs the reference's own extractor code run on
    the DINOv2-API stub; layer=None = last block.  bf16 operand tolerance, stated per token."""
    import make_golden as MG
    from dino_tracker_amd.utils import get_dino_features_video
    gold = torch.from_numpy(_gold()[f"tokens_{tag}_l11"])
    video = MG.p1_video()[:gold.shape[0]]
    out = get_dino_features_video(video.cuda(), model_name=MG.P1_CASE["model"], facet="tokens", stride=7, layer=None,
                                  device="cuda:0", state_dict=MG.p1_weights(ls))
    assert out.shape == gold.shape and out.device.type == "cpu"
    for t in range(gold.shape[0]):
        cos, rel = _check(out[t].permute(1, 2, 0).reshape(-1, 384), gold[t].permute(1, 2, 0).reshape(-1, 384))
        print(f"tokens {tag} frame {t}: min cos {cos:.6f} rel {rel:.2e}")


def test_cls_row_and_layer_mean_match_reference_golden():
    """get_feature_from_input(img, [2, 5]): mean over two hooked layers, CLS row included (models/extractor.py:137-150)."""
    import make_golden as MG
    gold = torch.from_numpy(_gold()["feature_with_cls_l2_l5"])
    ex = VitExtractor(MG.P1_CASE["model"], stride=7, device="cuda:0", state_dict=MG.p1_weights(1.0))
    m = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1)
    s = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1)
    x = ((MG.p1_video()[:1] - m) / s).cuda()
    tok = ex.get_feature_from_input(x, layers=[2, 5]).cpu()
    assert tok.shape == gold.shape
    _check(tok[0], gold[0])
    cls_err = (tok[0, 0] - gold[0, 0]).norm() / gold[0, 0].norm()
    assert cls_err < 2e-2, cls_err  # the CLS token itself, not just "finite"


def test_qkv_facets_match_reference_golden():
    """a4: qkv record, key facet through utils.get_dino_features_video(facet='keys'), key self-similarity."""
    import make_golden as MG
    from dino_tracker_amd.utils import get_dino_features_video
    g = _gold()
    sd = MG.p1_weights(1.0)
    ex = VitExtractor(MG.P1_CASE["model"], stride=7, device="cuda:0", state_dict=sd)
    m = torch.tensor(A.IMAGENET_MEAN).view(1, 3, 1, 1)
    s = torch.tensor(A.IMAGENET_STD).view(1, 3, 1, 1)
    video = MG.p1_video()[:1]
    x = ((video - m) / s).cuda()
    qkv = ex.get_qkv_feature_from_input(x)
    assert len(qkv) == 12
    _check(qkv[1][0].cpu(), torch.from_numpy(g["qkv_l1"])[0], cos_min=0.9995, rel_max=2e-2)
    keys = get_dino_features_video(video.cuda(), model_name=MG.P1_CASE["model"], facet="keys", stride=7, layer=3,
                                   device="cuda:0", state_dict=sd)
    gk = torch.from_numpy(g["keys_l3"])
    assert keys.shape == gk.shape
    _check(keys[0].permute(1, 2, 0).reshape(-1, 384), gk[0].permute(1, 2, 0).reshape(-1, 384), cos_min=0.9995)
    q = ex.get_queries_from_input(x, layers=[1])
    v = ex.get_values_from_input(x, layers=[1])
    gq = torch.from_numpy(g["qkv_l1"]).reshape(1, 222, 3, 384)
    _check(q[0].cpu(), gq[0, :, 0], cos_min=0.9995)
    _check(v[0].cpu(), gq[0, :, 2], cos_min=0.9995)
    ssim = ex.get_keys_self_sim_from_input(x, layer_num=1).cpu()
    assert (ssim - torch.from_numpy(g["keys_self_sim_l1"])).abs().max() < 2e-2
    attn = ex.get_attn_feature_from_input(x)[1]
    assert attn.shape == (1, 6, 222, 222) and (attn.sum(-1) - 1).abs().max() < 1e-4
    ref_attn = torch.softmax((gq[:, :, 0].reshape(1, 222, 6, 64).permute(0, 2, 1, 3) * 0.125)
                             @ gq[:, :, 1].reshape(1, 222, 6, 64).permute(0, 2, 3, 1), dim=-1)
    assert (attn.cpu() - ref_attn).abs().max() < 5e-3


def test_full_resolution_all_blocks_bench_weights():
    """476 x 854, all 12 blocks, the benchmark's weights (LayerScale 0.1) vs the fp32 oracle: the feature-level error
    of the bf16 path on exactly what bench.py runs (one frame; the oracle's ViT costs ~1.6 TFLOP on the host)."""
    sd = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1)
    ex = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd)
    video = synth.synth_video(1, 476, 854, seed=2000)
    feat = ex.encode(video)
    ref = A.vit_tokens(video, sd, "dinov2_vits14").permute(1, 2, 0).reshape(-1, 384)
    cos, rel = _check(feat[0], ref)
    print(f"full-res 12 blocks, bench weights: min token cos {cos:.6f}, rel Frobenius {rel:.3e}")


def test_attention_guards_and_safe_pass():
    """dtk_vit_attention alone on crafted bf16 operands vs an fp64 softmax on the SAME operands.  The kernel exponentiates
    scores optimistically against a reference that starts at 0 (vit_attention2.h, MODE 1); the rows below force every
    branch of that logic: (5) scores up to +-60 -> row sums beyond 2^40 -> power-of-two rescale of O / l; (6) scores of
    +-300 -> inf -> poisoned sum -> safe pass; (7) about -300 for EVERY key -> all-zero row -> safe pass; and ordinary
    rows that share their wave with them.  S is not a multiple of the 64-key tile (masked tail)."""
    from dino_tracker_amd import ops
    from dino_tracker_amd._lib import check, lib
    g = torch.Generator().manual_seed(11)
    F, Hh, S = 1, 2, 1000
    Sp = 1024
    qs = 0.125 * 1.4426950408889634
    q = torch.zeros(F, Hh, Sp, 64)
    k = torch.zeros(F, Hh, Sp, 64)
    v = torch.zeros(F, Hh, Sp, 64)
    q[:, :, :S] = torch.randn(F, Hh, S, 64, generator=g) * qs * 1.5
    k[:, :, :S] = torch.randn(F, Hh, S, 64, generator=g)
    v[:, :, :S] = torch.randn(F, Hh, S, 64, generator=g)
    k[0, 0, :S, 0] += 50.0
    q[0, 0, 5] *= 25.0
    q[0, 0, 6] *= 120.0
    q[0, 0, 7] *= 0.01
    q[0, 0, 7, 0] = -6.0
    q[0, 1, 700] *= 120.0   # a poisoned row in another wave / head, late in the sweep
    qb, kb, vb = q.bfloat16(), k.bfloat16(), v.bfloat16()
    vt = vb.transpose(2, 3).contiguous()  # [F][H][64][Sp]
    out = torch.empty(F, S, Hh * 64, dtype=torch.bfloat16, device="cuda")
    qd, kd, vd = qb.cuda().contiguous(), kb.cuda().contiguous(), vt.cuda().contiguous()
    check(lib().dtk_vit_attention(ops._p(qd), ops._p(kd), ops._p(vd), ops._p(out), F, Hh, S, Sp, ops._stream()))
    s = qb.double()[:, :, :S] @ kb.double()[:, :, :S].transpose(2, 3)            # exp2-domain scores
    p = torch.softmax(s * 0.6931471805599453, dim=-1)
    ref = (p @ vb.double()[:, :, :S]).permute(0, 2, 1, 3).reshape(F, S, Hh * 64)
    got = out.double().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().amax(dim=-1)[0]
    scale = ref.abs().amax(dim=-1)[0].clamp(min=0.05)
    rel = err / scale
    # bf16 P and bf16 output: 2^-8 of the row's largest output, every row, incl. the crafted ones
    assert rel.max() < 6e-3, (int(rel.argmax()), rel.max().item())
    for row in (5, 6, 7, 700):
        assert p[0, row // 1000 if row == 700 else 0].max() > 0  # (rows exist)
    # the crafted rows really are extreme: near one-hot / far from the reference 0
    assert s[0, 0, 6].abs().max() > 200 and s[0, 0, 7].max() < -150 and s[0, 0, 5].abs().max() > 45
