#!/usr/bin/env python
"""Benchmark of the per-video inference hot path on MI355X (contract: see the round prompt / DESIGN.md section 6).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over ONE synthetic video per rank (weak scaling: each rank tracks its own video,
rank 0 gathers the trajectories over RCCL).  Metric: query-points*frames/s = world * N * T * K / wall.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TF = 2500.0   # dense bf16/fp16 MFMA
F32_PEAK_TF = 157.3         # f32 vector / f32-input MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=90)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--width", type=int, default=384, help="feature width C (384 = ViT-S/14)")
    ap.add_argument("--method", default="auto", choices=["auto", "exact", "mfma"])
    ap.add_argument("--stages", default="extract,refine,track",
                    help="comma list of extract (ViT), refine (Delta-DINO), track (ModelInference.infer); stages that "
                         "are left out are computed once outside the timed region")
    ap.add_argument("--features", default="vit", choices=["vit", "synthetic"],
                    help="vit: features come from the (random-weight) ViT on the synthetic video; synthetic: the "
                         "feature-level generator with dense anchors (worst case for the tracker stage)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=1, help="queries in the bounded CPU-baseline sample")
    ap.add_argument("--cpu-frames", type=int, default=45,
                    help="frames in the bounded CPU-baseline sample (the oracle's work grows with T^2: T + T*T maps per query)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(dev))  # nccl == RCCL on ROCm

    from dino_tracker_amd import ops, synth
    from dino_tracker_amd.dataset import RangeNormalizer
    from dino_tracker_amd.model_inference import ModelInference
    import gpu_util
    gpu_util.DEV = dev

    H, W, T, N, C = 476, 854, args.frames, args.queries, args.width
    nx = int(round(N ** 0.5))
    ny = N // nx
    assert nx * ny == N, "--queries must be a square number (grid)"
    method = {"exact": ops.TRACK_EXACT, "mfma": ops.TRACK_MFMA}.get(args.method)
    if method is None:
        method = ops.TRACK_MFMA if ops.feat_f16_bytes(__import__("dino_tracker_amd")._lib.make_geom(T, C, H, W)) > 0 else ops.TRACK_EXACT

    stages = [x for x in args.stages.split(",") if x]
    from dino_tracker_amd.extractor import VitExtractor
    from dino_tracker_amd.tracker import Tracker
    import ctypes
    from dino_tracker_amd._lib import lib, make_geom
    model_name = {384: "dinov2_vits14", 768: "dinov2_vitb14", 1024: "dinov2_vitl14"}[C]
    # per-rank synthetic inputs: translating-texture video, seeded random weights
    video = synth.synth_video(T, H, W, seed=2000 + rank).to(dev)
    head = synth.synth_head_weights(3)
    delta = synth.synth_delta_dino_weights(C, seed=4)
    queries = synth.grid_queries(nx, ny, H, W, 0).to(dev)
    # no DINOv2 checkpoint exists offline: seeded random weights of the named architecture; LayerScale mean 0.1 keeps
    # the untrained encoder from collapsing all tokens onto one vector (synth.make_vit_weights)
    ex = VitExtractor(model_name, stride=7, device=dev,
                      state_dict=synth.make_vit_weights(model_name, seed=2, layerscale=0.1))
    if args.features == "vit":
        feats0 = ex.encode(video)
    else:
        feats0 = synth.synth_features(T, C, 67, 121, seed=1000 + rank).to(dev).permute(0, 2, 3, 1).reshape(T, 67 * 121, C).contiguous()
        stages = [x for x in stages if x != "extract"]
    trk = Tracker(video=video, dino_features=feats0, dino_patch_size=14, stride=7, device=dev, track_method=method)
    trk.tracker_head.load_state_dict(head)
    trk.delta_dino.load_state_dict(delta)
    trk.to(dev).eval()
    mi = ModelInference(trk, RangeNormalizer((W, H, T), device=dev), 0.7, 0.6)  # caches refined features once

    gather_buf = None

    def step():
        nonlocal gather_buf
        if "extract" in stages:
            trk.set_video(video, ex.encode(video))        # P1: ViT-S/14 block-11 tokens, stride 7
        if "refine" in stages or "extract" in stages:
            trk.cache_refined_embeddings()                # P2: dino + Delta-DINO(video)
        traj, occ = mi.infer(queries) if "track" in stages else (None, None)   # P3
        if world > 1 and traj is not None:
            from dino_tracker_amd import sharding
            gather_buf = sharding.gather_results(traj, occ, N, T, dev)  # RCCL gather of the results only
        return traj, occ

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- per-kernel profile pass (separate from the timed region) ------------------------------------------
    pairs = int(mi.last_counts[0]) if hasattr(mi, "last_counts") else 0
    maps = N * T + pairs * T
    ops.profile_enable(True)
    step()
    prof = ops.profile_collect()
    ops.profile_enable(False)
    roofline = None
    if prof:
        HW, S = 67 * 121, 67 * 121 + 1
        depth = 12 if C in (384, 768) else 24
        f_delta = 2.0 * (406504 * 4800 + 101626 * 204800 + 25466 * 819200 + 6420 * 6400 * C)
        # ALGORITHMIC flops of one step per kernel, and the peak that bounds it (TFLOP/s, dense)
        algo = {
            "vit_attention": (4.0 * S * S * C * depth * T, MFMA_F16_PEAK_TF),
            "vit_gemm_qkv": (2.0 * S * C * 3 * C * depth * T, MFMA_F16_PEAK_TF),
            "vit_gemm_proj": (2.0 * S * C * C * depth * T, MFMA_F16_PEAK_TF),
            "vit_gemm_fc1": (2.0 * S * C * 4 * C * depth * T, MFMA_F16_PEAK_TF),
            "vit_gemm_fc2": (2.0 * S * C * 4 * C * depth * T, MFMA_F16_PEAK_TF),
            # split-fp16 kernels: every fp32 MAC costs three fp16 MFMA MACs (hi*hi + hi*lo + lo*hi)
            "vit_patch_embed": (2.0 * HW * 588 * C * T, MFMA_F16_PEAK_TF / 3),
            "dd_conv1": (2.0 * 406504 * 4800 * T, MFMA_F16_PEAK_TF / 3),
            "dd_conv23": (2.0 * (101626 * 204800 + 25466 * 819200) * T, MFMA_F16_PEAK_TF / 3),
            "dd_conv4": (2.0 * 6420 * 6400 * C * T, MFMA_F16_PEAK_TF / 3),
            "corr_peaks": (2.0 * HW * C * maps, MFMA_F16_PEAK_TF),
            "refine_corr": (2.0 * 225 * C * maps, F32_PEAK_TF),   # (2*RD+5)^2 window cells per map
            "refine_head": (2.0 * (169 + 121) * 144 * maps, F32_PEAK_TF),
        }
        dom = max(prof, key=lambda k: prof[k][0])
        ms, launches = prof[dom]
        if dom in algo:
            fl, peak = algo[dom]
            achieved = fl / (ms * 1e-3) / 1e12
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None}
        else:
            roofline = {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": None, "traffic": None}
        # HBM-side bytes per launch of that kernel from the committed PMC passes (scripts/pmc_traffic.py)
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")) as fh:
                tr = json.load(fh)["kernels"].get(dom)
            if tr:
                roofline["traffic"] = tr["bytes_per_launch"]
                roofline["traffic_note"] = "bytes per launch, rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE (profiles/r01_pmc_traffic.json)"
        except (OSError, ValueError, KeyError):
            pass
        roofline["avg_launch_ms"] = round(ms / max(launches, 1), 4)
        roofline["launches"] = launches
        roofline["kernel_ms"] = {k: round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("DTK_BENCH_KERNELS", "12"))]}
        roofline["kernel_tflops"] = {k: round(algo[k][0] / (prof[k][0] * 1e-3) / 1e12, 1) for k in prof
                                     if k in algo and prof[k][0] > 0}

    # ---- CPU baseline: the oracle (torch fp32 port of the reference algorithm) on a bounded sample --------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_algo as A
        nq = max(1, args.cpu_queries)
        tc = max(2, min(T, args.cpu_frames))
        sel = torch.linspace(0, N - 1, nq).long()
        q_cpu = queries.cpu()[sel]
        feats_cpu = trk.refined_features.cpu()[:tc].contiguous()
        c0 = time.perf_counter()
        _, _, cs_cpu, _ = A.infer(feats_cpu, q_cpu, head, H, W, return_aux=True)
        cdt = time.perf_counter() - c0
        cpu = {"value": round(nq * tc / cdt, 3), "unit": "query-points*frames/s", "cores": torch.get_num_threads(),
               "kind": "port",
               "sample": f"oracle.infer (ModelInference.infer restatement, fp32 torch CPU) on {nq} of the {N} queries and "
                         f"the first {tc} of the {T} frames with all their anchors ({int((cs_cpu >= 0.7).sum())} anchor "
                         f"pairs, {nq * tc + int((cs_cpu >= 0.7).sum()) * tc} correlation maps; the full workload has "
                         f"{T + T * T} per query, i.e. costs more per query-frame), features resident; {cdt:.1f}s"}

    if rank == 0:
        if int(os.environ.get("DTK_DEBUG", "0")) & 4096:
            dc = (ctypes.c_ulonglong * 4)()
            lib().dtk_debug_counters(dc)
            print("redo reasons [overflow, none, nonpositive, -]:", list(dc), file=sys.stderr)
        tc = (ctypes.c_int * 3)()
        track_counts = None
        if hasattr(lib(), "dtk_debug_track_counts") and lib().dtk_debug_track_counts(tc) == 0:
            track_counts = {"sources": tc[0], "whole_map_tier": tc[1], "exact_tier": tc[2]}  # last dtk_track call
        out = {
            "metric": "query-points*frames/s", "value": round(world * N * T * args.steps / dt, 1),
            "unit": "query-points*frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("mixed: bf16 ViT, split-f16 convs (fp32-grade), " + ("f32 tracker" if method == ops.TRACK_EXACT
                                                                       else "f16 candidates + f32 deciders in the tracker")
                      + "; f32 accumulate"),
            "data": "synthetic",
            "config": {"workload": f"854x480x{T} synthetic video (model res 854x476, 67x121 tokens, C={C}), {N} grid "
                                   f"queries, one video per GPU",
                       "stages": stages, "features": args.features,
                       "track_method": "exact" if method == ops.TRACK_EXACT else "mfma",
                       "anchor_pairs": pairs, "correlation_maps_per_step": maps, "anchor_track_tiers": track_counts,
                       "parallelism": f"video-parallel x{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
