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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=1, help="queries in the bounded CPU-baseline sample")
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

    # per-rank synthetic video at the feature level (SURVEY.md 8d "north-star" generator): dense anchors
    feats = synth.synth_features(T, C, 67, 121, seed=1000 + rank)
    head = synth.synth_head_weights(3)
    queries = synth.grid_queries(nx, ny, H, W, 0).to(dev)
    trk = gpu_util.make_tracker(torch.zeros(T, 3, H, W), feats, head, method=method)
    mi = ModelInference(trk, RangeNormalizer((W, H, T), device=dev), 0.7, 0.6)

    gather_buf = None

    def step():
        nonlocal gather_buf
        traj, occ = mi.infer(queries)
        if world > 1:
            import torch.distributed as dist
            payload = torch.cat([traj.reshape(-1), occ.reshape(-1).float()])
            if rank == 0:
                gather_buf = [torch.empty_like(payload) for _ in range(world)]
            dist.gather(payload, gather_buf if rank == 0 else None, dst=0)
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
    pairs = int(mi.last_counts[0])
    maps = N * T + pairs * T
    ops.profile_enable(True)
    step()
    prof = ops.profile_collect()
    ops.profile_enable(False)
    roofline = None
    if prof:
        dom = max(prof, key=lambda k: prof[k][0])
        ms, launches = prof[dom]
        HW = 67 * 121
        per_map_flops = {"corr_exact": 2.0 * HW * C, "corr16": 2.0 * HW * C, "head_exact": 576.0 * HW,
                         "head16": 576.0 * HW}
        peak = {"corr_exact": F32_PEAK_TF, "head_exact": F32_PEAK_TF, "corr16": MFMA_F16_PEAK_TF,
                "head16": F32_PEAK_TF}
        if dom in per_map_flops:
            achieved = per_map_flops[dom] * maps / (ms * 1e-3) / 1e12
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak[dom],
                        "unit": "TFLOP/s", "frac": round(achieved / peak[dom], 4), "traffic": None,
                        "avg_launch_ms": round(ms / max(launches, 1), 4), "launches": launches}
        else:
            roofline = {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": None, "traffic": None, "avg_launch_ms": round(ms / max(launches, 1), 4),
                        "launches": launches}
        roofline["kernel_ms"] = {k: round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]}

    # ---- CPU baseline: the oracle (torch fp32 port of the reference algorithm) on a bounded sample --------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_algo as A
        nq = max(1, args.cpu_queries)
        sel = torch.linspace(0, N - 1, nq).long()
        q_cpu = queries.cpu()[sel]
        c0 = time.perf_counter()
        _, _, cs_cpu, _ = A.infer(feats, q_cpu, head, H, W, return_aux=True)
        cdt = time.perf_counter() - c0
        cpu = {"value": round(nq * T / cdt, 3), "unit": "query-points*frames/s", "cores": torch.get_num_threads(),
               "kind": "port",
               "sample": f"oracle.infer (ModelInference.infer restatement, fp32 torch CPU) on {nq} of the {N} queries, "
                         f"all {T} frames and all their anchors ({int((cs_cpu >= 0.7).sum())} anchor pairs), features "
                         f"resident; {cdt:.1f}s"}

    if rank == 0:
        out = {
            "metric": "query-points*frames/s", "value": round(world * N * T * args.steps / dt, 1),
            "unit": "query-points*frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if method == ops.TRACK_EXACT else "f16-mfma+f32-refine",
            "data": "synthetic",
            "config": {"workload": f"854x480x{T} synthetic video (model res 854x476, 67x121 tokens, C={C}), {N} grid "
                                   f"queries, one video per GPU",
                       "stages": ["track: ModelInference.infer on cached refined features"],
                       "track_method": "exact" if method == ops.TRACK_EXACT else "mfma",
                       "anchor_pairs": pairs, "correlation_maps_per_step": maps, "parallelism": f"video-parallel x{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
