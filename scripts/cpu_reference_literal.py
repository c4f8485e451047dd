"""CPU timing of the LITERAL reference next to the oracle port (VERDICT r2 item 9a / SURVEY 8d).

bench.py's `cpu_baseline` is `kind: "port"`: the oracle (oracle/ref_algo.py) computes ONE map per (source, target frame).
The reference's own `Tracker.forward` computes T x (T + 1) maps per call and keeps T (models/tracker.py:159-160) and gathers
two (T + 1)-frame copies of the volume per call (:316-317), so its literal per-query cost is far higher.  This script runs
the reference's un-modified classes (Tracker / ModelInference / TrackerHead; DeltaDINO built at the embedding width, as in
tests/golden/make_golden.py, because the reference hard-codes 1024 channels) on K queries at full T with all their anchors,
and the oracle on the same inputs, on the same host cores, and prints the ratio.

    python scripts/cpu_reference_literal.py [--frames 90] [--queries 2] [--width 384] [--threads N]

Needs the reference checkout ($DTK_REFERENCE_ROOT or /root/reference): runs in the build container, or on a GPU box through
scripts/gpurun_with_reference.sh.  Writes gpurun_out/cpu_reference_literal_C<width>.json.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dino_tracker_amd import synth  # noqa: E402
from oracle import ref_algo as A  # noqa: E402
from oracle import ref_harness  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=90)
    ap.add_argument("--queries", type=int, default=2)
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    H, W, T, C, K = 476, 854, args.frames, args.width, args.queries
    R = ref_harness.load()
    video = torch.zeros(T, 3, H, W)
    feats = synth.synth_features(T, C, 67, 121, seed=1000)          # the dense-anchor feature field of bench.py --features synthetic
    head = synth.synth_head_weights(3)
    nx = max(1, int(round(K ** 0.5)))
    queries = synth.grid_queries(nx, (K + nx - 1) // nx, H, W, 0)[:K]
    tmp = tempfile.mkdtemp()
    emb_path = os.path.join(tmp, "dino_embed_video.pt")
    torch.save(feats, emb_path)
    trk = R.tracker.Tracker(video=video, ckpt_path=tmp, dino_embed_path=emb_path, dino_patch_size=14, stride=7, device="cpu")
    trk.delta_dino = R.delta_dino.DeltaDINO(channels=[3, 64, 128, 256, C], vit_stride=7)  # zero-initialised last conv: identity
    trk.tracker_head.load_state_dict(head)
    rn = R.dataset.RangeNormalizer(shapes=(W, H, T))
    t0 = time.perf_counter()
    with torch.no_grad():  # (the scripts run under @torch.no_grad(): inference_grid.py:12)
        mi = R.model_inference.ModelInference(trk, rn, anchor_cosine_similarity_threshold=0.7, cosine_similarity_threshold=0.6)
    t_refine = time.perf_counter() - t0                              # cache_refined_embeddings: Delta-DINO on all T frames
    trk.eval()
    out = {"config": f"854x476x{T}, C={C}, {K} queries, dense-anchor synthetic features", "threads": torch.get_num_threads(),
           "host_cpus": os.cpu_count(), "literal_cache_refined_embeddings_s": t_refine}
    with torch.no_grad():
        t0 = time.perf_counter()
        call = trk(R.model_inference.generate_trajectory_input(queries[0], video))  # one literal Tracker.forward call
        out["literal_s_per_forward_call"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        traj, occ = mi.infer(queries, batch_size=None)
        out["literal_s_per_query"] = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    rt, ro, cs, _ = A.infer(feats, queries, head, H, W, return_aux=True)
    out["port_s_per_query"] = (time.perf_counter() - t0) / K
    out["anchors_per_query"] = float((cs >= 0.7).sum()) / K
    out["literal_over_port"] = out["literal_s_per_query"] / out["port_s_per_query"]
    out["max_dxy_px_literal_vs_port"] = float((traj[..., :2] - rt).abs().max())
    out["occ_mismatch_literal_vs_port"] = int((occ != ro).sum())
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"cpu_reference_literal_C{C}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
