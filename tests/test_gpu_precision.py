"""-m gpu: the ESCALATED precision of the ViT encoder (round 6, csrc/vit_split.h).

Plain fp16 operands put the features 1.3e-4 (benchmark weights) .. 2.1e-3 (DINOv2-like outlier statistics) from the fp32
reference -- scripts/p1_error_budget.py emulates exactly those roundings in float64 and shows that no single tensor
dominates.  precision="split" runs every product of every block on hi + lo operands (three MFMAs per product): the features
are then what the fp32 oracle itself is from float64.  Here:
  * the attention stage on split operands vs float64, fp16 and bf16 planes (crafted rows included);
  * whole-encoder features vs the FLOAT64 oracle: benchmark weights, outlier weights, ViT-L, mixed (per-block) escalation;
  * the range escalation: a saturated fp16 call is re-run on SPLIT bf16 operands (16 significant bits at fp32's range);
  * precision="auto": the calibration measures fast-vs-split on the first frames and picks;
  * end to end FROM THE VIDEO under the outlier weights: positions and flags vs the oracle on the same video."""
import json
import os
import sys

import pytest
import torch

from dino_tracker_amd import ops, synth
from dino_tracker_amd._lib import OPERAND_BF16, OPERAND_F16, check, lib
from dino_tracker_amd.extractor import VitExtractor
from oracle import ref_algo as A

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return ((got - ref).norm() / ref.norm()).item()


def _ref64(video, sd, name, layer):
    """The oracle in float64 (the fp32 oracle is itself 5e-7 .. 3e-6 from this): [T][HW][D]."""
    sd64 = {k: v.double() for k, v in sd.items()}
    d = sd["cls_token"].shape[-1]
    return torch.stack([A.vit_tokens(video[t:t + 1].double(), sd64, name, layer=layer).permute(1, 2, 0).reshape(-1, d)
                        for t in range(video.shape[0])])


def _split(x, dt):
    hi = x.to(dt)
    lo = (x - hi.float()).to(dt)
    return hi, lo


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_attention_stage_on_split_operands(dt):
    """dtk_vit_attention_split vs softmax(q k^T) v in float64 on the SAME fp32 operands (the hi + lo planes carry them to
    2^-22 / 2^-16): S = 300 (not a multiple of the 64-key tile or of the 128-query block: masked padding keys, clamped query
    rows), two frames x three heads, rows with scores of +-100 binades and a near-one-hot row."""
    tdt = torch.float16 if dt == "fp16" else torch.bfloat16
    F, Hh, S, Sp = 2, 3, 300, 320
    g = torch.Generator().manual_seed(5)
    q = torch.zeros(F, Hh, Sp, 64)
    k = torch.zeros(F, Hh, Sp, 64)
    v = torch.zeros(F, Hh, Sp, 64)
    q[:, :, :S] = torch.randn(F, Hh, S, 64, generator=g) * 0.6
    k[:, :, :S] = torch.randn(F, Hh, S, 64, generator=g)
    v[:, :, :S] = torch.randn(F, Hh, S, 64, generator=g)
    q[0, 0, 5] *= 12.0           # scores of +-100: a sharp row
    k[0, 1, 200, 3] += 30.0      # one key that a query looking along channel 3 picks almost alone
    q[0, 1, 17, 3] = 4.0
    q[1, 2, 299] *= 20.0         # the last valid query row, sharp
    planes = []
    for x, tr in ((q, False), (k, False), (v, True)):
        xx = x.transpose(2, 3).contiguous() if tr else x
        hi, lo = _split(xx, tdt)
        planes += [hi.cuda().contiguous(), lo.cuda().contiguous()]
    oh = torch.empty(F, S, Hh * 64, dtype=tdt, device="cuda")
    ol = torch.empty_like(oh)
    check(lib().dtk_vit_attention_split(*[ops._p(t) for t in planes], ops._p(oh), ops._p(ol), F, Hh, S, Sp,
                                        OPERAND_F16 if dt == "fp16" else OPERAND_BF16, ops._stream()))
    # the operands the device actually saw: hi + lo
    qd, kd, vd = [(planes[2 * i].double() + planes[2 * i + 1].double()).cpu() for i in range(3)]
    vd = vd.transpose(2, 3)
    s = qd[:, :, :S] @ kd[:, :, :S].transpose(2, 3)
    p = torch.softmax(s * 0.6931471805599453, dim=-1)
    ref = (p @ vd[:, :, :S]).permute(0, 2, 1, 3).reshape(F, S, Hh * 64)
    got = (oh.double() + ol.double()).cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().amax(dim=-1)
    scale = ref.abs().amax(dim=-1).clamp(min=0.05)
    relrow = err / scale
    rel = relrow.max().item()
    top = torch.topk(relrow.flatten(), 5)
    print(f"split attention {dt}: max row-relative error {rel:.2e} (median {relrow.median().item():.2e}; worst rows (frame * S + query) "
          f"{top.indices.tolist()} {[f'{x:.1e}' for x in top.values.tolist()]}); max |s| {s.abs().max().item():.0f}")
    # fp16 planes: 2^-22 operands, fp32 accumulation and exponentials; bf16 planes: 2^-16 operands (a score of 100 moves by ~1e-3)
    assert rel < (2e-5 if dt == "fp16" else 4e-3), rel
    assert s.abs().max() > 100


@pytest.mark.parametrize("weights", ["bench", "outlier"])
def test_split_precision_is_fp32_grade(weights):
    """precision="split" against the float64 oracle, next to the fast path on the same frames.  Emulated in float64
    (scripts/p1_error_budget.py): fast 1.3e-4 / 2.1e-3, split 2.1e-7 / 3.6e-6 (bench, layer 11 / outlier, layer 5); the fp32
    oracle itself: 4.8e-7 / 3.0e-6."""
    if weights == "bench":
        sd, layer = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1), 11
    else:
        sd, layer = synth.make_outlier_vit_weights(300.0), 5
    video = synth.synth_video(2, 238, 322, seed=80)
    ref = _ref64(video, sd, "dinov2_vits14", layer)
    fast = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd).encode(video, layer=layer)
    exs = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd, precision="split", check_range=True)
    strict = exs.encode(video, layer=layer)
    assert exs.last_overflow == 0 and torch.isfinite(strict).all()
    r_fast, r_split = _rel(fast, ref), _rel(strict, ref)
    tok_rel = ((strict.double().cpu() - ref).norm(dim=-1) / ref.norm(dim=-1)).max().item()
    print(f"{weights}: fast {r_fast:.2e}, split {r_split:.2e} (worst token {tok_rel:.2e}) vs the float64 oracle")
    assert r_split < (2e-6 if weights == "bench" else 1.5e-5), r_split
    assert tok_rel < (4e-6 if weights == "bench" else 6e-5), tok_rel
    assert r_split < r_fast / 50
    rep = exs.precision_report()
    assert rep["feature_error_class"].startswith("fp32-grade") and rep["split_blocks"] == list(range(12))
    # the two forms of the split GEMM (LDS-DMA 256 x 128 tiles in production, register-staged 128 x 128 tiles with tiled_gemms) agree
    # to fp32 summation order
    ext = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd, precision="split")
    ext.tiled_gemms = True
    assert _rel(ext.encode(video, layer=layer), strict) < 2e-6
    # tokens (CLS kept) and the qkv facet come out of the same split pass
    tok = exs.encode(video[:1], layer=layer, want="tokens")
    assert torch.equal(tok[0, 1:], strict[0])
    qkv = exs.encode(video[:1], layer=layer, want="qkv").cpu()
    sd64 = {k: v.double() for k, v in sd.items()}
    qref = A.vit_qkv(video[:1].double(), sd64, "dinov2_vits14", layer)
    assert _rel(qkv, qref) < 2e-5


def test_blocks_escalate_individually_and_mixed_passes_hand_over():
    """precision=[...]: split blocks and fast blocks in one pass (a fast block's pending 16-bit update is applied by the next
    split block's LayerNorm; a split block leaves none; the last block may be either kind).  Under the outlier weights block 0
    carries the largest share of the fast path's error (emulation: 2.14e-3 -> 1.36e-3 with block 0 exact)."""
    sd = synth.make_outlier_vit_weights(300.0)
    video = synth.synth_video(1, 238, 322, seed=80)
    ref = _ref64(video, sd, "dinov2_vits14", 5)
    rel = {}
    for name, pr in (("fast", "fast"), ("first", [0]), ("last", [5]), ("odd", [1, 3, 5]), ("even", [0, 2, 4]), ("split", "split")):
        ex = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd, precision=pr)
        rel[name] = _rel(ex.encode(video, layer=5), ref)
        assert ex.last_overflow == 0
    print("mixed escalation:", json.dumps(rel))
    assert rel["split"] < 1.5e-5 < rel["first"] < rel["fast"]
    assert rel["first"] < 0.8 * rel["fast"]          # block 0 alone buys a third
    assert rel["even"] < rel["fast"] and rel["odd"] < rel["fast"] and rel["last"] <= rel["fast"] * 1.02
    with pytest.raises(ValueError):
        VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd, precision=[12])


def test_range_escalation_is_split_bf16():
    """A value beyond 65504 needs fp32's exponent.  Round 5 re-ran such a call on plain bf16 operands (8 significant bits: 7-10x
    FURTHER from the reference than the fp16 pass it replaced).  Round 6: the default re-runs it on SPLIT bf16 operands -- fp32's
    range and 16 significant bits --, bit-identical to VitExtractor(operand_dtype="bf16", precision="split"), sticky, counted,
    reported."""
    sd = synth.make_outlier_vit_weights(3.0e5)
    video = synth.synth_video(1, 140, 210, seed=81)
    ref = _ref64(video, sd, "dinov2_vits14", 4)
    exh = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd)
    with pytest.warns(RuntimeWarning, match="SPLIT bf16"):
        healed = exh.encode(video, layer=4)
    assert exh.range_fallbacks == 1 and exh.operand_dtype == "bf16" and exh.last_overflow == 0
    assert exh.precision_report()["feature_error_class"].startswith("2^-16")
    exs = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd, operand_dtype="bf16", precision="split")
    assert torch.equal(healed, exs.encode(video, layer=4))
    assert torch.equal(exh.encode(video, layer=4), healed) and exh.range_fallbacks == 1     # sticky
    plain = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd, operand_dtype="bf16").encode(video, layer=4)
    r_split, r_plain = _rel(healed, ref), _rel(plain, ref)
    print(f"range escalation: split bf16 {r_split:.2e}, plain bf16 {r_plain:.2e}")
    assert torch.isfinite(healed).all() and r_split < 3e-4 and r_split < r_plain / 20


def test_auto_precision_measures_and_picks():
    """precision="auto": the first frames of the first call are encoded both ways; the extractor stays on split operands when the
    fast features are further than auto_tol (2.5e-4 relative) from them -- benchmark weights: fast (1.3e-4); outlier weights:
    split (2e-3) -- and says so in `calibration` / precision_report()."""
    video = synth.synth_video(3, 238, 322, seed=80)
    sd = synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=0.1)
    ex = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd, precision="auto")
    a = ex.encode(video)
    c = ex.calibration
    print("auto, benchmark weights:", json.dumps(c))
    assert c["chosen"] == "fast" and 5e-5 < c["rel_fast_vs_split"] < 2.5e-4 and c["frames"] == 2 and not ex.split_blocks
    assert torch.equal(a, VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sd).encode(video))
    sdo = synth.make_outlier_vit_weights(300.0)
    exo = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sdo, precision="auto")
    b = exo.encode(video, layer=5)
    co = exo.calibration
    print("auto, outlier weights:", json.dumps(co))
    assert co["chosen"] == "split" and co["rel_fast_vs_split"] > 1e-3 and len(exo.split_blocks) == 12
    assert torch.equal(b, VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=sdo, precision="split").encode(video, layer=5))
    assert exo.precision_report()["calibration"]["chosen"] == "split"
    exo.encode(video[:1], layer=5)                       # calibrated once
    assert exo.calibration is co


def test_vitl_split_precision():
    """The reference's shipped width (config/preprocessing.yaml: dinov2_vitl14, block 15): D = 1024, 16 heads -- the split
    GEMMs' column tiles and the attention's head loop at that width, vs the float64 oracle."""
    sd = synth.make_vit_weights("dinov2_vitl14", seed=6, layerscale=0.1)
    video = synth.synth_video(1, 140, 210, seed=83)
    ref = _ref64(video, sd, "dinov2_vitl14", 15)
    fast = VitExtractor("dinov2_vitl14", stride=7, device="cuda:0", state_dict=sd).encode(video, layer=15)
    strict = VitExtractor("dinov2_vitl14", stride=7, device="cuda:0", state_dict=sd, precision="split").encode(video, layer=15)
    r_fast, r_split = _rel(fast, ref), _rel(strict, ref)
    print(f"ViT-L block 15: fast {r_fast:.2e}, split {r_split:.2e}")
    assert r_split < 3e-6 and r_split < r_fast / 50


def test_end_to_end_from_the_video_under_outlier_weights():
    """VERDICT r5 item 1 "done" criterion: video -> HIP ViT -> HIP Delta-DINO -> HIP infer vs the fp32 oracle on the same video
    with the OUTLIER ViT weights (massive activations, gains to 8, a sharp block, a 300 x MLP), 854 x 476, T = 16, 256 queries =
    4096 positions.  With precision="split": p99 <= 1e-3 px, every flag of a query without an arbitrated point identical, every
    point beyond 1e-3 px arbitrated in float64 (0 failures).  The fast path on the same weights is measured next to it (no bound
    asserted: it is what the escalation exists for), and precision="auto" must pick split here.  The oracle runs on the GPU in
    fp32 (pinned against its CPU form in tests/test_gpu_fullsize.py)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import e2e_error
    out = {}
    for pr in ("split", "fast"):
        r = e2e_error.run(476, 854, 16, 16, oracle_device="cuda", weights="outlier", precision=pr)
        out[pr] = {k: r[k] for k in ("feature_rel_err_P1", "feature_rel_err_refined", "px_err_vs_oracle_on_same_video",
                                     "points_beyond_1e-3px", "arbitration_failures", "arbitrated_rate", "occ_mismatch_same_video",
                                     "occ_mismatch_same_video_queries_without_a_tie", "encode_seconds", "precision_report")}
        out[pr]["arbitrated_gaps"] = [(a["gap64"], a["delta"], a["dist_px"], a["ok"]) for a in r["arbitrated"]][:32]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "e2e_error_outlier_test_476x854x16.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("end to end under outlier weights:", json.dumps(out))
    s = out["split"]
    assert s["feature_rel_err_P1"] < 2e-5, s
    assert s["px_err_vs_oracle_on_same_video"]["p99"] <= 1e-3, s
    assert s["arbitration_failures"] == 0 and s["points_beyond_1e-3px"] <= 8, s
    assert s["occ_mismatch_same_video_queries_without_a_tie"] == 0, s
    assert out["fast"]["feature_rel_err_P1"] > 20 * s["feature_rel_err_P1"]
    ex = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=synth.make_outlier_vit_weights(300.0), precision="auto")
    ex.encode(synth.synth_video(2, 476, 854, seed=2000).cuda())
    assert ex.calibration["chosen"] == "split", ex.calibration


def test_end_to_end_from_the_video_at_vitl_block15():
    """The reference AS CONFIGURED (config/preprocessing.yaml:9-12: dinov2_vitl14, block 15; models/networks/delta_dino.py:9 hard-codes
    1024): video -> HIP ViT-L -> HIP Delta-DINO (C = 1024) -> HIP infer (the C = 1024 tracker path) vs the fp32 oracle on the same
    video, 854 x 476, T = 16, 256 queries = 4096 positions, every query (VERDICT r5 missing #2: until round 5 the width had an
    8-query x 4-frame test on given features).  precision="auto" calibrates on the first two frames (ViT-L's fast features sit at
    3.3e-4 relative, above auto_tol) and must land on split operands; asserted: p99 <= 1e-3 px, every point beyond 1e-3 px arbitrated in
    float64, flags identical for the queries without an arbitrated point; on IDENTICAL features no exemption.  The fast path is
    measured beside it."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import e2e_error
    out = {}
    for pr in ("auto", "fast"):
        r = e2e_error.run(476, 854, 16, 16, oracle_device="cuda", precision=pr, model="dinov2_vitl14")
        out[pr] = {k: r[k] for k in ("feature_rel_err_P1", "feature_rel_err_refined", "px_err_vs_oracle_on_same_video",
                                     "points_beyond_1e-3px", "arbitration_failures", "arbitrated_rate", "occ_mismatch_same_video",
                                     "occ_mismatch_same_video_queries_without_a_tie", "occ_mismatch_same_features", "encode_seconds",
                                     "precision_report", "track_tiers")}
        out[pr]["same_features"] = {k: r["px_err_vs_oracle_on_same_features"][k] for k in ("max", "p99", "points_beyond_1e-3px", "arbitration_failures")}
        out[pr]["arbitrated_gaps"] = [(a["gap64"], a["delta"], a["dist_px"], a["ok"]) for a in r["arbitrated"]][:16]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "e2e_error_vitl_test_476x854x16.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("end to end at ViT-L block 15:", json.dumps(out))
    a = out["auto"]
    assert a["precision_report"]["calibration"]["chosen"] == "split", a["precision_report"]
    assert a["feature_rel_err_P1"] < 2e-5, a
    assert a["px_err_vs_oracle_on_same_video"]["p99"] <= 1e-3, a
    assert a["arbitration_failures"] == 0 and a["points_beyond_1e-3px"] <= 8, a
    assert a["occ_mismatch_same_video_queries_without_a_tie"] == 0, a
    s = a["same_features"]
    assert s["arbitration_failures"] == 0 and s["points_beyond_1e-3px"] <= 4 and a["occ_mismatch_same_features"] <= 0 + 90 * s["points_beyond_1e-3px"], a


def test_auto_blocks_escalates_only_what_is_needed_at_vitl():
    """precision="auto-blocks": where the all-fast pass measures beyond auto_tol (ViT-L block 15: 3.3e-4 against 2.5e-4) the
    calibration measures every block's lone contribution and keeps on the fast kernels the largest set that MEASURES below
    block_margin * auto_tol; the rest runs on split operands.  Asserted: some but not all of the 16 blocks are escalated, the
    features of the whole video are within auto_tol of the float64 oracle, and end to end from the video (854 x 476, T = 16, 256
    queries, every query) the positions hold the same bar as the all-split pass: p99 <= 1e-3 px, every point beyond 1e-3 px
    arbitrated in float64, flags identical for the queries without an arbitrated point."""
    sd = synth.make_vit_weights("dinov2_vitl14", seed=6, layerscale=0.1)
    video = synth.synth_video(3, 140, 210, seed=83)
    ref = _ref64(video, sd, "dinov2_vitl14", 15)
    ex = VitExtractor("dinov2_vitl14", stride=7, device="cuda:0", state_dict=sd, precision="auto-blocks")
    got = ex.encode(video, layer=15)
    c = ex.calibration
    print("auto-blocks, ViT-L block 15:", json.dumps(c))
    assert c["chosen"] == "blocks" and c["rel_fast_vs_split"] > c["tol"]
    b = c["blocks"]
    assert 0 < len(b["fast_blocks"]) < 16 and sorted(b["fast_blocks"] + b["split_blocks"]) == list(range(16))
    assert b["measured"] <= b["bound"] == 0.8 * c["tol"]
    assert ex.split_blocks == frozenset(b["split_blocks"]) | frozenset(range(16, 24))   # (blocks beyond the calibrated layer: escalated)
    assert ex.precision_report()["feature_error_class"].startswith("mixed")
    r = _rel(got, ref)
    print(f"auto-blocks features vs float64: {r:.2e} ({len(b['split_blocks'])} of 16 blocks split)")
    assert r <= c["tol"]
    ex.encode(video[:1], layer=15)
    assert ex.calibration is c                           # calibrated once
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import e2e_error
    r = e2e_error.run(476, 854, 16, 16, oracle_device="cuda", precision="auto-blocks", model="dinov2_vitl14")
    a = {k: r[k] for k in ("feature_rel_err_P1", "px_err_vs_oracle_on_same_video", "points_beyond_1e-3px", "arbitration_failures",
                           "arbitrated_rate", "occ_mismatch_same_video", "occ_mismatch_same_video_queries_without_a_tie",
                           "encode_seconds", "precision_report")}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "e2e_error_vitl_auto_blocks_476x854x16.json"), "w") as fh:
        json.dump(a, fh, indent=1)
    print("end to end at ViT-L block 15, auto-blocks:", json.dumps(a))
    assert a["precision_report"]["calibration"]["chosen"] == "blocks"
    assert a["feature_rel_err_P1"] <= 2.5e-4, a
    assert a["px_err_vs_oracle_on_same_video"]["p99"] <= 1e-3, a
    assert a["arbitration_failures"] == 0, a
    assert a["occ_mismatch_same_video_queries_without_a_tie"] == 0, a


@pytest.mark.parametrize("name,layer,dt", [("dinov2_vits14", 3, "fp16"), ("dinov2_vits14", 2, "bf16"), ("dinov2_vitl14", 1, "fp16")])
def test_split_gemm_staged_epilogues_are_bit_identical(name, layer, dt):
    """Round 6: gemm_split_dma_kernel stages its output tile in LDS (hi / lo planes of the GELU hidden and of Q / K; the fp32 update of
    the residual stream, LayerScale applied where x is updated) and writes whole rows; its operand stages arrive through buffer
    descriptors.  Same arithmetic per value, so the features are BIT-identical to the direct 8-byte-store epilogues
    (DTK_VIT_GEMM_WIDE_V1), on a frame whose token count is not a multiple of the 256-row tile."""
    sd = synth.make_vit_weights(name, seed=9, layerscale=0.1)
    # 154 x 238: 694 tokens per frame (V^T leaves element by element); 140 x 154: 400 = positions in fours (8-byte pieces, the 854 x 476 form)
    for hw in ((154, 238), (140, 154)):
        video = synth.synth_video(3, hw[0], hw[1], seed=84)
        out = {}
        for form in ("new", "v1"):
            ex = VitExtractor(name, stride=7, device="cuda:0", state_dict=sd, precision="split", operand_dtype=dt)
            ex.gemm_wide_v1 = form == "v1"
            out[form] = ex.encode(video, layer=layer)
            q = ex.encode(video[:1], layer=layer, want="qkv")
            out[form + "_qkv"] = q
        assert torch.isfinite(out["new"]).all()
        assert torch.equal(out["new"], out["v1"]) and torch.equal(out["new_qkv"], out["v1_qkv"]), hw
