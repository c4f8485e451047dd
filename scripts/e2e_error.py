"""End-to-end error from the VIDEO: video -> HIP ViT -> HIP Delta-DINO -> HIP infer, against the fp32 oracle run on the
same video (oracle ViT -> oracle refine -> oracle infer).  north_star's 1e-3 px is stated on identical inputs; the
P3 / P2 tests hold it on identical FEATURES, this script measures what the 16-bit operands of P1 add on top (fp16 by
default since round 3; `bf16` as fifth argument gives the round-2 arithmetic for comparison).
Writes gpurun_out/e2e_error_*.json (copied to profiles/ by hand).
Usage: python scripts/e2e_error.py [H W T nq [fp16|bf16 [split,fp16 [cpu|cuda [bench|ls1|outlier [fast,split,auto,...]]]]]]
  (5: ViT operand type; 6: Delta-DINO operand modes, one run each; 7: where the oracle runs; 8: the ViT weights -- round 6:
  `outlier` = synth.make_outlier_vit_weights, DINOv2-like statistics; 9: VitExtractor precision modes, one run each; 10: the model,
  dinov2_vitl14 = the reference's shipped configuration, block 15, C = 1024)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dino_tracker_amd import ops, synth  # noqa: E402
from dino_tracker_amd.dataset import RangeNormalizer  # noqa: E402
from dino_tracker_amd.extractor import VitExtractor  # noqa: E402
from dino_tracker_amd.model_inference import ModelInference  # noqa: E402
from dino_tracker_amd.tracker import Tracker  # noqa: E402
from oracle import ref_algo as A  # noqa: E402


TIE_MARGIN = 5e-6  # cosine units; see argmax_margins


def argmax_margins(refined, queries, H, W, radius=35.0, stride=7):
    """How far the oracle's OWN first-pass decisions are from flipping: for query n and frame t the maximum of the ReLU'd
    cosine map minus the largest value OUTSIDE the 35-px disk around the arg-max (tracker_head.py:84,115: the arg-max picks
    the disk, everything else follows continuously).  With the benchmark's untrained ViT the maps are nearly flat far from
    the peak -- median margin 4e-3, 4 % below 2e-4, and single points at ONE or TWO fp32 ulps (6e-8 .. 1.2e-7): such a point
    has no defined answer (a different summation order in fp32 flips it by hundreds of px), so position errors are reported
    separately for margins below TIE_MARGIN.  Measured on MI355X (854x476x8, 64 queries, fp16 operands): the one point at
    margin 1.2e-7 lands 717 px away (with bf16 operands too), every other point -- margins from 9.9e-6 up -- is within
    3.6e-4 px."""
    import torch.nn.functional as F
    T, C, h, w = refined.shape
    tq = queries[:, 2].long()
    q_emb = A.sample_bilinear(refined, queries[:, :2], tq, H, W)
    rr = torch.arange(h, device=refined.device)[None, :, None]
    cc = torch.arange(w, device=refined.device)[None, None, :]
    out = []
    for t in range(T):
        fr = refined[t].reshape(C, -1)
        x = F.relu((q_emb @ fr) / (q_emb.norm(dim=1)[:, None] * fr.norm(dim=0)[None]).clamp(min=1e-8))
        k = x.argmax(1)
        row, col = k // w, k % w
        d2 = (float(stride) * (rr - row[:, None, None])) ** 2 + (float(stride) * (cc - col[:, None, None])) ** 2
        far = x.reshape(-1, h, w).masked_fill(d2 <= radius ** 2, -1.0).reshape(len(k), -1).max(1).values
        out.append(x.max(1).values - far)
    return torch.stack(out, 1)  # [N, T]


_ORACLE_CACHE = {}


def vit_weights(weights, layerscale=0.1, model="dinov2_vits14"):
    """bench: the benchmark's seeded ViT (LayerScale `layerscale`); ls1: the same with LayerScale 1.0 (the hub models'
    init_values); outlier: synth.make_outlier_vit_weights (massive activations, gains to 8, a sharp block, a 300 x MLP)."""
    if weights == "outlier":
        return synth.make_outlier_vit_weights(300.0, model)
    return synth.make_vit_weights(model, seed=2 if model == "dinov2_vits14" else 6, layerscale=1.0 if weights == "ls1" else layerscale)


MODEL_LAYER = {"dinov2_vits14": None, "dinov2_vitb14": None, "dinov2_vitl14": 15}   # config/preprocessing.yaml:9-12: ViT-L, block 15


def _oracle(H, W, T, nq, layerscale, seed, od="cpu", weights="bench", model="dinov2_vits14"):
    """The fp32 oracle's side of the comparison (minutes on CPU, seconds with od = "cuda": the restatement takes its device from
    its inputs), cached so that several device configurations share it."""
    key = (H, W, T, nq, layerscale, seed, od, weights, model)
    if key not in _ORACLE_CACHE:
        name = model
        sd_cpu = vit_weights(weights, layerscale, model)
        video_cpu = synth.synth_video(T, H, W, seed=seed)
        head_cpu = synth.synth_head_weights(3)
        delta_cpu = synth.synth_delta_dino_weights(synth.VIT_CONFIGS[model]["dim"], seed=4)
        queries_cpu = synth.grid_queries(nq, nq, H, W, 0, margin=min(60.0, H / 6))
        to = lambda d: {k: v.to(od) for k, v in d.items()}  # noqa: E731
        sd, head, delta, video, queries = to(sd_cpu), to(head_cpu), to(delta_cpu), video_cpu.to(od), queries_cpu.to(od)
        t0 = time.time()
        dino = torch.stack([A.vit_tokens(video[t:t + 1], sd, name, layer=MODEL_LAYER[model]) for t in range(T)])
        refined = A.refine_features(video, dino, delta)
        rt, ro, rcs, _ = A.infer(refined, queries, head, H, W, return_aux=True)
        if od != "cpu":
            torch.cuda.synchronize()
        _ORACLE_CACHE[key] = dict(name=name, sd=sd_cpu, video=video_cpu, head=head, head_cpu=head_cpu, delta=delta_cpu,
                                  queries=queries, dino=dino, refined=refined, rt=rt, ro=ro, rcs=rcs, seconds=time.time() - t0,
                                  margin=argmax_margins(refined, queries, H, W))
    return _ORACLE_CACHE[key]


def arbitrate(o, dev_refined, traj, H, W, flagged):
    """Every flagged point (n, t) -- device position more than 1e-3 px from the oracle's -- must be the reference's answer for
    a near-tie of the cosine map decided the other way (oracle.ref_algo.tie_arbiter): the cell the device took has a FLOAT64
    cosine within its band of the float64 maximum, and the head evaluated around that cell lands within 1e-3 px of the device's
    position.  The band (round 5, VERDICT r4 weak #1b): fp32 rounding of the two evaluations (A.fp32_dot_band(C) = 2 sqrt(C)
    2^-24 = 2.3e-6) + what the device's features MEASURABLY moved the float64 cosines of the two cells in question by -- not the
    global feature deviation (round 4: 6.6e-4)."""
    refined, queries, head = o["refined"], o["queries"], o["head"]
    C = refined.shape[1]
    out = []
    for n, t in flagged:
        r = A.tie_arbiter(refined, queries[n], t, traj[n, t], head, H, W, A.fp32_dot_band(C), dev_feats=dev_refined)
        r.update(query=int(n), frame=int(t), margin_fp32=float(o["margin"][n, t]),
                 err_px=float((traj[n, t] - o["rt"][n, t]).norm()))
        out.append(r)
    return out


def run(H, W, T, nq, layerscale=0.1, seed=2000, operand_dtype="fp16", p2_operands=None, oracle_device="cpu", weights="bench",
        precision="fast", on_overflow="split-bf16", model="dinov2_vits14"):
    """oracle_device: "cpu" (the form pinned on the reference; minutes at T = 16) or "cuda" (the same restatement on device
    tensors in fp32 -- what makes T = 90 / 1024 queries affordable; pinned against the CPU form in tests/test_gpu_fullsize.py)."""
    dev = "cuda:0"
    od = oracle_device
    o = _oracle(H, W, T, nq, layerscale, seed, od, weights, model)
    name, sd, video, head, delta, queries = o["name"], o["sd"], o["video"], o["head"], o["delta"], o["queries"]
    dino, refined, rt, ro, rcs = o["dino"], o["refined"], o["rt"], o["ro"], o["rcs"]
    ex = VitExtractor(name, stride=7, device=dev, state_dict=sd, operand_dtype=operand_dtype, precision=precision,
                      on_overflow=on_overflow)
    torch.cuda.synchronize()
    t0 = time.time()
    feat = ex.encode(video.to(dev), layer=MODEL_LAYER[model])
    torch.cuda.synchronize()
    encode_seconds = time.time() - t0
    trk = Tracker(video=video.to(dev), dino_features=feat, dino_patch_size=14, stride=7, device=dev)
    trk.tracker_head.load_state_dict(o["head_cpu"])
    trk.delta_dino.load_state_dict(delta)
    if p2_operands is not None:
        trk.delta_dino.conv_operands = p2_operands
    trk.to(dev).eval()
    mi = ModelInference(trk, RangeNormalizer((W, H, T), device=dev), 0.7, 0.6)
    traj, occ = mi.infer(queries.to(dev))
    traj, occ = traj.to(od), occ.to(od)
    ph, pw = A.feature_grid(H, W)
    # feature-level error too
    dfe = feat.to(od).reshape(T, ph, pw, -1).permute(0, 3, 1, 2)
    rel = ((dfe - dino).norm() / dino.norm()).item()
    dev_refined = trk.refined_features.to(od)
    rel_refined = ((dev_refined - refined).norm() / refined.norm()).item()
    # P3 on IDENTICAL features (the device's own refined volume through the oracle): isolates P1's / P2's contribution
    rt_same, ro_same = A.infer(dev_refined, queries, head, H, W)
    err2 = (traj - rt).norm(dim=-1)
    err = err2.reshape(-1)
    err_same2 = (traj - rt_same).norm(dim=-1)
    err_same = err_same2.reshape(-1)
    q = torch.tensor([0.5, 0.9, 0.99, 1.0], device=err.device)
    margin = o["margin"].reshape(-1)
    tie = margin < TIE_MARGIN
    dec = err[~tie]
    flagged = [(int(n), int(t)) for n, t in torch.nonzero(err2 > 1e-3).tolist()]
    arb = arbitrate(o, dev_refined, traj, H, W, flagged)
    # identical features: a point beyond 1e-3 px can only be an fp32 near-tie (two evaluation orders): fp32 band alone
    flagged_same = [(int(n), int(t)) for n, t in torch.nonzero(err_same2 > 1e-3).tolist()]
    arb_same = []
    for n, t in flagged_same:
        r = A.tie_arbiter(dev_refined, queries[n], t, traj[n, t], head, H, W, A.fp32_dot_band(dev_refined.shape[1]))
        r.update(query=int(n), frame=int(t), err_px=float(err_same2[n, t]))
        arb_same.append(r)
    # occlusion flags of a query whose trajectory contains an arbitrated (tie) point follow that point: compared apart
    tie_q = sorted({a["query"] for a in arb})
    clean = torch.ones(len(queries), dtype=torch.bool, device=occ.device)
    clean[tie_q] = False
    clean_same = torch.ones(len(queries), dtype=torch.bool, device=occ.device)
    clean_same[sorted({a["query"] for a in arb_same})] = False
    fl = lambda x: [float(v) for v in x]  # noqa: E731
    return {
        "argmax_margin": {"tie_threshold_cos": TIE_MARGIN, "points": int(err.numel()), "ties": int(tie.sum()),
                          "tie_margins": fl(margin[tie][:64]), "tie_errors_px": fl(err[tie][:64]),
                          "margin_p01_p05_p50": fl(margin.quantile(torch.tensor([0.01, 0.05, 0.5], device=margin.device))),
                          "smallest_margin_of_a_point_within_1e-3px": float(margin[err <= 1e-3].min())},
        "px_err_decidable_points": {"p50": dec.quantile(q[0]).item(), "p99": dec.quantile(q[2]).item(), "max": dec.max().item(),
                                    "frac_le_1e-3": (dec <= 1e-3).float().mean().item()},
        "points_beyond_1e-3px": len(flagged), "arbitrated": arb,
        "arbitration_failures": sum(0 if a["ok"] else 1 for a in arb),
        "config": f"{W}x{H}x{T}, {nq * nq} queries, {model} {weights} weights (LayerScale {layerscale if weights == 'bench' else '-'}), "
                  f"seed {seed}, {operand_dtype} ViT operands, precision {precision}, Delta-DINO convolution operands "
                  f"{p2_operands or 'default'}, oracle on {od}",
        "weights": weights, "model": model, "track_tiers": dict(trk.last_track_stats) if trk.last_track_stats else None, "precision": precision, "precision_report": ex.precision_report(), "encode_seconds": encode_seconds,
        "arbitrated_rate": len(flagged) / max(1, int(err.numel())),
        "feature_rel_err_P1": rel, "feature_rel_err_refined": rel_refined,
        "px_err_vs_oracle_on_same_video": {"p50": err.quantile(q[0]).item(), "p90": err.quantile(q[1]).item(),
                                           "p99": err.quantile(q[2]).item(), "max": err.max().item(),
                                           "frac_le_1e-3": (err <= 1e-3).float().mean().item(),
                                           "frac_le_1e-1": (err <= 1e-1).float().mean().item(),
                                           "frac_le_1px": (err <= 1.0).float().mean().item()},
        "px_err_vs_oracle_on_same_features": {"max": err_same.max().item(), "p99": err_same.quantile(q[2]).item(),
                                              "points_beyond_1e-3px": len(flagged_same), "arbitrated": arb_same,
                                              "arbitration_failures": sum(0 if a["ok"] else 1 for a in arb_same)},
        "occ_mismatch_same_video": int((occ != ro).sum()), "occ_mismatch_same_features": int((occ != ro_same).sum()),
        "occ_mismatch_same_video_queries_without_a_tie": int((occ[clean] != ro[clean]).sum()),
        "occ_mismatch_same_features_queries_without_a_tie": int((occ[clean_same] != ro_same[clean_same]).sum()),
        "occ_total": int(ro.numel()), "anchors_oracle": int((rcs >= 0.7).sum()), "oracle_seconds": o["seconds"],
    }


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else [238, 322, 6, 4]
    dt = sys.argv[5] if len(sys.argv) > 5 else "fp16"
    modes = sys.argv[6].split(",") if len(sys.argv) > 6 else [None]
    modes = [None if m in ("default", "None") else m for m in modes]
    od = sys.argv[7] if len(sys.argv) > 7 else "cpu"
    weights = sys.argv[8] if len(sys.argv) > 8 else "bench"
    precisions = sys.argv[9].split(",") if len(sys.argv) > 9 else ["fast"]
    model = sys.argv[10] if len(sys.argv) > 10 else "dinov2_vits14"
    out = []
    for pr in precisions:
        # "bf16" as a precision name: plain bf16 operands (the round-5 heal target), for the three-way table of docs/PARITY.md
        kw = dict(operand_dtype="bf16", precision="fast") if pr == "bf16" else dict(operand_dtype=dt, precision=pr)
        out += [run(*a, p2_operands=m, oracle_device=od, weights=weights, model=model, **kw) for m in modes]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = f"{a[0]}x{a[1]}x{a[2]}_{dt}" + ("" if weights == "bench" and precisions == ["fast"] else f"_{weights}_{'-'.join(precisions)}") + \
        ("" if model == "dinov2_vits14" else "_" + model[-5:])
    with open(os.path.join(ROOT, "gpurun_out", f"e2e_error_{tag}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    # the arbitrated lists are long under outlier weights: print the summary only
    for r in out:
        print(json.dumps({k: r[k] for k in ("config", "feature_rel_err_P1", "feature_rel_err_refined", "px_err_vs_oracle_on_same_video",
                                            "points_beyond_1e-3px", "arbitration_failures", "arbitrated_rate", "occ_mismatch_same_video",
                                            "occ_mismatch_same_video_queries_without_a_tie", "encode_seconds", "precision_report")}, indent=1))
