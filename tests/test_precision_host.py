"""CPU: the host side of the escalated ViT precision (round 6) -- weight planes, per-block selection, the C struct the device
reads.  No kernel runs here; the device side is tests/test_gpu_precision.py."""
import ctypes

import pytest
import torch

from dino_tracker_amd import synth
from dino_tracker_amd._lib import VitLayer
from dino_tracker_amd.extractor import VitExtractor


def _planes(arr, i, name, shape, dt):
    n = shape[0] * shape[1]
    buf_t = ctypes.c_uint16 * n
    hi = torch.frombuffer(buf_t.from_address(getattr(arr[i], name)), dtype=dt).reshape(shape)
    lo_addr = getattr(arr[i], name + "_lo")
    lo = None if not lo_addr else torch.frombuffer(buf_t.from_address(lo_addr), dtype=dt).reshape(shape)
    return hi, lo


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_split_weight_planes_reconstruct_the_weights(dt):
    """dtk_vit_layer.*_w / *_w_lo of a split block are the hi / lo planes of w_scale * W: their sum gives W back to 2^-21 (fp16
    planes, scale 2^8 so that the lo halves of |w| ~ 1e-2 are normal numbers) / 2^-15 (bf16, scale 1); un-split blocks keep the
    plain 16-bit weights and NULL lo pointers."""
    sd = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1)
    sd["blocks.1.mlp.fc2.weight"][3, 5] = 190.0        # one large weight: the block's scale drops so that it stays inside fp16
    ex = VitExtractor("dinov2_vits14", stride=7, device="cpu", state_dict=sd, operand_dtype=dt, precision=[0, 1])
    tdt = torch.float16 if dt == "fp16" else torch.bfloat16
    arr = ex._build_layers(dt, ex.split_blocks)
    for i, want_scale in ((0, 256.0 if dt == "fp16" else 1.0), (1, 128.0 if dt == "fp16" else 1.0)):
        assert arr[i].w_scale == want_scale
        for name, key in (("qkv_w", "attn.qkv.weight"), ("proj_w", "attn.proj.weight"), ("fc1_w", "mlp.fc1.weight"), ("fc2_w", "mlp.fc2.weight")):
            w = sd[f"blocks.{i}.{key}"]
            hi, lo = _planes(arr, i, name, tuple(w.shape), tdt)
            assert lo is not None
            rec = (hi.double() + lo.double()) / want_scale
            err = ((rec - w.double()).abs() / w.double().abs().clamp(min=1e-3)).max().item()
            assert err < (2.0 ** -20 if dt == "fp16" else 2.0 ** -14), (i, name, err)
            assert torch.isfinite(hi.float()).all() and hi.float().abs().max() < 65504
    hi, lo = _planes(arr, 2, "qkv_w", (1152, 384), tdt)
    assert lo is None and arr[2].w_scale == 1.0 and torch.equal(hi, sd["blocks.2.attn.qkv.weight"].to(tdt))
    assert ctypes.sizeof(VitLayer) == 18 * 8 + 8      # 14 + 4 pointers, the scale, padding: include/dtk.h dtk_vit_layer


def test_precision_argument_and_report():
    sd = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1)
    ex = VitExtractor("dinov2_vits14", stride=7, device="cpu", state_dict=sd)
    assert ex.precision == "fast" and not ex.split_blocks and ex.on_overflow == "split-bf16"
    assert ex.precision_report()["feature_error_class"].startswith("2^-12")
    ex.set_precision("split")
    assert ex.split_blocks == frozenset(range(12)) and ex.precision_report()["feature_error_class"].startswith("fp32-grade")
    ex.set_precision([3, 1])
    assert ex.precision == "blocks" and ex.precision_report()["split_blocks"] == [1, 3]
    assert ex.precision_report()["feature_error_class"].startswith("mixed")
    ex.set_precision("auto")
    assert ex.calibration is None and not ex.split_blocks
    ex.calibration = {"chosen": "split"}          # (a measurement made under another request does not survive a new one)
    ex.set_precision("auto-blocks")
    assert ex.precision == "auto-blocks" and ex.calibration is None and not ex.split_blocks
    for bad in ("strict", [12], [-1]):
        with pytest.raises(ValueError):
            ex.set_precision(bad)
    with pytest.raises(ValueError):
        VitExtractor("dinov2_vits14", stride=7, device="cpu", state_dict=sd, on_overflow="fp32")


def test_precision_from_the_environment(monkeypatch):
    """$DTK_VIT_PRECISION: how the reference's un-modified preprocessing script (which passes no such argument) is run on split
    operands; an explicit argument wins."""
    sd = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1)
    monkeypatch.setenv("DTK_VIT_PRECISION", "split")
    assert len(VitExtractor("dinov2_vits14", stride=7, device="cpu", state_dict=sd).split_blocks) == 12
    assert not VitExtractor("dinov2_vits14", stride=7, device="cpu", state_dict=sd, precision="fast").split_blocks


def test_auto_blocks_selection_logic():
    """precision="auto-blocks" (extractor._pick_blocks) on a synthetic error model: block b alone fast costs e_b, a fast set costs
    amp * sqrt(sum e_b^2).  The cheapest blocks stay fast while the quadrature sum fits block_margin * auto_tol; a measurement above
    the bound (amp > 1) evicts the costliest member; a block whose lone pass saturates (inf) is always escalated; blocks beyond
    `layer` are not touched."""
    sd = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1)
    e = [3e-4, 0.5e-4, 0.6e-4, float("inf"), 0.7e-4, 1.0e-4, 1.2e-4, 0.4e-4]     # blocks 0..7; layer = 7

    def model(amp, calls):
        def measure(split_blocks):
            fast = [b for b in range(8) if b not in split_blocks]
            calls.append(tuple(fast))
            v = (amp if len(fast) > 1 else 1.0) * sum(e[b] ** 2 for b in fast) ** 0.5
            return v, v == float("inf")
        return measure

    ex = VitExtractor("dinov2_vits14", stride=7, device="cpu", state_dict=sd, precision="auto-blocks", auto_tol=2.5e-4, block_margin=0.8)
    ex.calibration, calls = {}, []
    split = ex._pick_blocks(model(1.0, calls), layer=7)
    # ascending: 7 (0.4), 1 (0.5), 2 (0.6), 4 (0.7), 5 (1.0), 6 (1.2): sqrt(.16+.25+.36+.49+1.0) = 1.503e-4 <= 2e-4; + 1.44 -> 1.92e-4 <= 2e-4; + block 0 no
    assert split == frozenset({0, 3, 8, 9, 10, 11}), split          # (blocks 8 .. 11 lie beyond `layer`: not measured, escalated)
    c = ex.calibration
    assert c["chosen"] == "blocks" and c["blocks"]["fast_blocks"] == [1, 2, 4, 5, 6, 7] and abs(c["blocks"]["measured"] - 1.924e-4) < 1e-6
    assert calls[:8] == [(b,) for b in range(8)] and len(calls) == 9
    ex.calibration, calls = {}, []
    split = ex._pick_blocks(model(1.2, calls), layer=7)                                  # sets measure 1.2 x the quadrature sum: 2.31e-4 -> block 6 leaves -> 1.80e-4
    assert split == frozenset({0, 3, 6, 8, 9, 10, 11}) and len(ex.calibration["blocks"]["passes"]) == 2, (split, ex.calibration)
    ex.calibration = {}
    assert ex._pick_blocks(model(100.0, []), layer=7) == frozenset(range(12)) - {7} and ex.calibration["blocks"]["fast_blocks"] == [7]
    ex.calibration = {}
    assert ex._pick_blocks(lambda split_blocks: (1e-3, False), layer=7) == frozenset(range(12)) and ex.calibration["chosen"] == "split"
