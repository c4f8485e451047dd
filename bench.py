#!/usr/bin/env python
"""Benchmark of the per-video inference hot path on MI355X (contract: the round prompt / DESIGN.md section 6).

    python bench.py --gpus N --steps K --warmup W         (N > 1 spawns its own ranks when WORLD_SIZE is unset)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (P1 ViT features -> P2 Delta-DINO refinement -> P3 ModelInference.infer) over
  * default (weak scaling): ONE synthetic 854x480x90 video with 1024 grid queries per rank;
  * --videos V (strong scaling): a batch of V videos sharded v = r (mod world) (north_star's 30-video batch:
    4/4/4/4/4/4/3/3 on 8 GPUs), rank 0 receives every result through one RCCL gather per round;
  * --mode query-parallel (strong scaling on ONE video, SURVEY 8e level 2): frames split over the ranks for P1 / P2,
    one all-gather of the refined volume, queries split over the ranks for P3, one gather of the results.
Metric: query-points*frames/s = videos * N * T * K / wall (max over ranks).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TF = 2500.0   # dense bf16/fp16 MFMA
F32_PEAK_TF = 157.3         # f32 vector / f32-input MFMA
H, W = 476, 854             # model resolution of an 854x480 source video (config/train.yaml:6-7)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)   # (the second step still meets one fresh multi-GB hipMalloc: docs/MEASUREMENTS.md)
    ap.add_argument("--frames", type=int, default=90)
    ap.add_argument("--queries", type=int, default=1024)
    ap.add_argument("--width", type=int, default=384, help="feature width C (384 = ViT-S/14)")
    ap.add_argument("--method", default="auto", choices=["auto", "exact", "mfma"])
    ap.add_argument("--videos", type=int, default=0,
                    help="0: one video per rank and step (weak scaling); V > 0: a batch of V videos per step sharded over "
                         "the ranks (strong scaling)")
    ap.add_argument("--mode", default="video-parallel", choices=["video-parallel", "query-parallel"])
    ap.add_argument("--stages", default="extract,refine,track",
                    help="comma list of extract (ViT), refine (Delta-DINO), track (ModelInference.infer); stages that "
                         "are left out are computed once outside the timed region")
    ap.add_argument("--features", default="vit", choices=["vit", "synthetic"],
                    help="vit: features come from the (random-weight) ViT on the synthetic video; synthetic: the "
                         "feature-level generator with dense anchors (worst case for the tracker stage)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-power", action="store_true", help="skip the shader-clock / board-power sampling legs")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 PMC child passes that measure roofline.traffic live")
    ap.add_argument("--no-videos30", action="store_true", help="skip the second timed phase (north_star's 30-video batch, strong scaling)")
    ap.add_argument("--ab", default="", help="comma-separated A / B switches: attention_v2 (round 2-3 attention kernel), "
                                             "gemm_ws_v1 (round 1-3 weight-stationary GEMMs)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of ALL THREE oracle legs (infer, ViT, Delta-DINO); 0 = the fastest of the thread sweep on this box")
    ap.add_argument("--cpu-queries", type=int, default=4, help="queries of the TIMED CPU sample (SURVEY 8d: K = 4, full T)")
    ap.add_argument("--parity-queries", type=int, default=1024,
                    help="queries of parity_sample: HIP infer vs the oracle run ON THE GPU in fp32, on the step's own refined volume "
                         "(default: all of them; 0 disables the leg)")
    ap.add_argument("--cpu-parity-queries", type=int, default=8,
                    help="queries that also go through the oracle on the HOST (>= --cpu-queries; pins GPU-torch against CPU-torch; 8 since "
                         "every query goes through the GPU oracle -- a slow host spends 5 s per query here)")
    ap.add_argument("--parity-video-frames", type=int, default=90,
                    help="frames of the from-the-video parity leg on the GPU oracle (oracle ViT -> refine -> infer; 0 disables it)")
    ap.add_argument("--cpu-vit-frames", type=int, default=2,
                    help="frames the oracle's ViT / Delta-DINO legs are timed on (also the from-the-video parity leg)")
    ap.add_argument("--operands", default="fp16", choices=["fp16", "bf16"], help="operand type of the ViT's matrix units")
    ap.add_argument("--precision", default="auto", choices=["fast", "split", "auto", "auto-blocks"],
                    help="VitExtractor precision: auto (default: the extractor MEASURES fast-vs-split on the first two frames of its first "
                         "call -- outside the timed region -- and keeps the fast operands only if they are within 2.5e-4 of the split ones; "
                         "on the benchmark's weights it measures 1.3e-4 and runs fast: config.vit_precision.calibration), fast (one 16-bit "
                         "number per operand, no check), split (hi + lo operands in every block: fp32-grade features, ~2x the step)")
    ap.add_argument("--video-lengths", default="",
                    help="with --videos V: comma list of frame counts (cycled over the V videos; e.g. DAVIS-like 25..104) -- the batch is "
                         "then scheduled longest-first over the ranks (sharding.lpt_assignment) instead of v = r (mod world)")
    ap.add_argument("--vit-frame-batch", type=int, default=0, help="frames per pass of the ViT encoder (0 = library default)")
    ap.add_argument("--track-round", type=int, default=0, help="sources per round of dtk_track (0 = library default, 4194304)")
    ap.add_argument("--no-train", action="store_true", help="skip the test-time-training leg (BASELINE config 5; extra key `train`)")
    ap.add_argument("--train-widths", default="384", help="feature widths of the training leg, comma-separated (384 = north star, 1024 = the reference's own)")
    ap.add_argument("--train-iters", type=int, default=30, help="timed iterations of the training leg per width and mode")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run` for N ranks on this node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def literal_over_port(C):
    """How much slower the reference's LITERAL per-call path is than the oracle port on the same host threads (measured in the
    build container, where the un-modified reference runs: scripts/cpu_reference_literal.py) -- from the committed file."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_cpu_reference_literal.json")) as fh:
            d = json.load(fh)
        for row in d["runs"]:
            if f"C={int(C)}," in row["config"]:
                return round(row["literal_over_port"], 1)
    except Exception:
        pass
    return None


def train_leg(args):
    """BASELINE.json config 5 next to the headline: per-video test-time training at 854 x 476 x T, config/train.yaml's batch sizes, every
    loss term on -- seconds per iteration of the device-side trainer with the iteration replayed from captured graphs and
    eagerly, kernel time per iteration by origin.  A child process per width (scripts/train_iter_bench.py: its own allocator and
    generators); the inference model of this process is idle meanwhile.  Not part of `value`."""
    res = {"config": f"854x476x{args.frames}, config/train.yaml batch sizes (512 pairs, 4 + 4 frames, 4 x 256 cycle points, 4 x 256 contrastive "
                     "points), all loss terms on; dino_tracker_amd.train control plane, trainer.GraphedIteration",
           "unit": "s per iteration", "data": "synthetic"}
    for width in [int(x) for x in args.train_widths.split(",") if x]:
        cmd = [sys.executable, os.path.join(ROOT, "scripts", "train_iter_bench.py"), "--width", str(width), "--frames", str(args.frames),
               "--iters", str(args.train_iters), "--modes", "graph,eager", "--kernel-share"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            res[f"C={width}"] = json.loads(line[-1]) if (r.returncode == 0 and line) else {"error": (r.stderr or r.stdout)[-400:]}
        except Exception as ex:  # noqa: BLE001
            res[f"C={width}"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one GPU per rank; a launcher that already narrowed the visible devices to one per rank leaves index 0 only
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    dist = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(dev))  # nccl == RCCL on ROCm
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        assert rccl_ranks == world == dist.get_world_size(), (rccl_ranks, world)

    from dino_tracker_amd import ops, sharding, synth
    from dino_tracker_amd._lib import make_geom
    from dino_tracker_amd.dataset import RangeNormalizer
    from dino_tracker_amd.extractor import VitExtractor
    from dino_tracker_amd.model_inference import ModelInference
    from dino_tracker_amd.tracker import Tracker

    T, N, C = args.frames, args.queries, args.width
    nx = int(round(N ** 0.5))
    ny = N // nx
    assert nx * ny == N, "--queries must be a square number (grid)"
    method = {"exact": ops.TRACK_EXACT, "mfma": ops.TRACK_MFMA}.get(args.method)
    if method is None:
        method = ops.TRACK_MFMA if ops.feat_f16_bytes(make_geom(T, C, H, W)) > 0 else ops.TRACK_EXACT
    stages = [x for x in args.stages.split(",") if x]
    qpar = args.mode == "query-parallel" and world > 1
    model_name = {384: "dinov2_vits14", 768: "dinov2_vitb14", 1024: "dinov2_vitl14"}[C]

    # ---- synthetic inputs: translating-texture videos, seeded random weights -----------------------------
    if qpar:
        my_videos = [0]                       # every rank works on THE video
    elif args.videos > 0:
        my_videos = sharding.videos_of_rank(args.videos, rank, world)
    else:
        my_videos = [rank]
    n_distinct = max(1, min(len(my_videos), 2))  # distinct clips held per rank; longer batches cycle through them
    seeds = [2000 + (0 if qpar else (my_videos[i] if my_videos else rank)) for i in range(n_distinct)]
    videos = [synth.synth_video(T, H, W, seed=s).to(dev) for s in seeds]
    head = synth.synth_head_weights(3)
    delta = synth.synth_delta_dino_weights(C, seed=4)
    queries = synth.grid_queries(nx, ny, H, W, 0).to(dev)
    # no DINOv2 checkpoint exists offline: seeded random weights of the named architecture; LayerScale mean 0.1 keeps
    # the untrained encoder from collapsing all tokens onto one vector (synth.make_vit_weights)
    vit_sd = synth.make_vit_weights(model_name, seed=2, layerscale=0.1)
    ex = VitExtractor(model_name, stride=7, device=dev, state_dict=vit_sd, operand_dtype=args.operands, precision=args.precision)
    ex.frame_batch = args.vit_frame_batch
    ex.attention_v2 = "attention_v2" in args.ab
    ex.attention_v4 = ex.attention_v4 or "attention_v4" in args.ab   # rounds 4-5: 64 queries per wave
    ex.gemm_ws_v1 = "gemm_ws_v1" in args.ab
    if args.features == "vit":
        feats0 = ex.encode(videos[0])
    else:
        feats0 = synth.synth_features(T, C, 67, 121, seed=1000 + rank).to(dev).permute(0, 2, 3, 1).reshape(T, 67 * 121, C).contiguous()
        stages = [x for x in stages if x != "extract"]
    trk = Tracker(video=videos[0], dino_features=feats0, dino_patch_size=14, stride=7, device=dev, track_method=method)
    trk.tracker_head.load_state_dict(head)
    trk.delta_dino.load_state_dict(delta)
    trk.to(dev).eval()
    trk.track_round_sources = args.track_round
    mi = ModelInference(trk, RangeNormalizer((W, H, T), device=dev), 0.7, 0.6)  # caches refined features once
    stats_acc = {"sources": 0, "whole_map_tier": 0, "exact_tier": 0, "syncs": 0}

    def one_video(video, trk=trk, mi=mi):
        if "extract" in stages:
            # P1: ViT-S/14 block-11 tokens, stride 7.  The fp16 overflow word of the encoder is read once per video, behind
            # the tracker's own synchronisations, instead of draining the stream between P1 and P2
            trk.set_video(video, ex.encode(video, defer_check=True))
        if "refine" in stages or "extract" in stages:
            trk.cache_refined_embeddings()                # P2: dino + Delta-DINO(video)
        res = mi.infer(queries) if "track" in stages else (None, None)   # P3
        if "extract" in stages:
            try:
                ex.check_overflow()
            except RuntimeError:
                trk.refined_features = None    # the deferred pass saturated: nothing derived from its features may survive
                raise
        return res

    # ---- ragged batches (round 6): clips of different lengths need their own Tracker (geometry is fixed per object); built
    # outside every timed region, one per distinct length this rank is assigned
    ctx = {T: (trk, mi)}

    def ctx_for(L):
        if L not in ctx:
            t2 = Tracker(video=videos[0][:L], dino_features=ex.encode(videos[0][:L]), dino_patch_size=14, stride=7, device=dev,
                         track_method=method)
            t2.tracker_head.load_state_dict(head)
            t2.delta_dino.load_state_dict(delta)
            t2.to(dev).eval()
            t2.track_round_sources = args.track_round
            ctx[L] = (t2, ModelInference(t2, RangeNormalizer((W, H, L), device=dev), 0.7, 0.6))
        return ctx[L]

    def scheduled_batch(lengths):
        """LPT assignment by frame count, this rank's list back to back, ONE gather at the end (sharding.run_scheduled)."""
        def run(v):
            tk, m_ = ctx[lengths[v]]
            return one_video(videos[(v // world) % n_distinct][:lengths[v]], tk, m_)
        return sharding.run_scheduled(lengths, N, dev, run)

    batch_lengths = None
    if args.videos > 0 and args.video_lengths:
        ll = [int(x) for x in args.video_lengths.split(",") if x]
        assert all(1 < x <= T for x in ll), "--video-lengths: every length in (1, --frames]"
        batch_lengths = [ll[v % len(ll)] for v in range(args.videos)]
        mine_tbl = sharding.lpt_assignment([sharding.video_cost(x) for x in batch_lengths], world)[rank]
        for v in mine_tbl:
            ctx_for(batch_lengths[v])

    def step():
        if qpar:
            return sharding.query_parallel_step(trk, ex, mi, videos[0], queries, dev, stages)
        if args.videos > 0 and batch_lengths is not None:
            return scheduled_batch(batch_lengths)
        if args.videos > 0:
            res = sharding.run_sharded(args.videos, N, T, dev, lambda v: one_video(videos[(v // world) % n_distinct]))
            return res
        traj, occ = one_video(videos[0])
        if world > 1 and traj is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            res = sharding.gather_results(traj, occ, N, T, dev)  # RCCL gather of the results only
            ev[1].record()
            gather_events.append(ev)
            return res
        return traj, occ

    gather_events = []   # (start, end) events around the gather of every step: read after the timed region

    def barrier():
        if world > 1:
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
    # shader clock / board power of whole steps: sampled over a REPEAT of the steps after the timed region (the polling thread
    # shares the interpreter with the launch loop; the timed region stays untouched)
    sampler = clock_power = None
    if rank == 0 and world == 1 and not args.no_clock_power:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from power_sampler import PowerSampler
        sampler = PowerSampler(local).start()
        c0 = time.perf_counter()
        for _ in range(max(2, min(args.steps, 5))):
            step()
        torch.cuda.synchronize()
        clock_power = {"steps_repeat": dict(sampler.stop(), ms_per_step=round((time.perf_counter() - c0) / max(2, min(args.steps, 5)) * 1e3, 2))}
    per_rank = None
    if world > 1:
        # every rank's own wall time of the K steps (the line's value uses the MAX), and the time its result gather took
        mine_t = torch.tensor([dt], device=dev, dtype=torch.float64)
        all_t = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(all_t, mine_t)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in all_t]
        dt = max(float(x.item()) for x in all_t)
        timed_ev = gather_events[-args.steps:] if len(gather_events) >= args.steps else gather_events
        g_ms = sum(a.elapsed_time(b) for a, b in timed_ev) / max(len(timed_ev), 1)
        per_rank = {"ms_per_step_by_rank": [round(x, 3) for x in rank_ms], "max": round(max(rank_ms), 3), "min": round(min(rank_ms), 3),
                    "result_gather_ms_rank0": round(g_ms, 4),
                    "note": "wall time of the K timed steps on each rank (value uses the max); the gather is the only collective of a step"}
    videos_per_step = 1 if qpar else (args.videos if args.videos > 0 else world)
    frames_per_step = sum(batch_lengths) if batch_lengths is not None else videos_per_step * T   # query-points x FRAMES: ragged batches count their own

    # ---- north_star's strong-scaling form in the SAME line: a batch of 30 videos sharded v = r (mod world),
    # 4/4/4/4/4/4/3/3 on 8 ranks, one RCCL gather per round; a second timed phase (one step, the kernels are warm) so that the
    # driver's plain `bench.py --gpus N` runs also carry the number north_star's ">= 6x at 8 GPUs on a 30-video batch" is about
    videos30 = None
    if args.videos == 0 and not qpar and not args.no_videos30 and stages == ["extract", "refine", "track"] and args.features == "vit":
        V30 = 30

        def timed(fn):
            barrier()
            c0 = time.perf_counter()
            fn()
            barrier()
            d = time.perf_counter() - c0
            if world > 1:
                tt = torch.tensor([d], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                d = float(tt.item())
            return d

        def sched_report(lengths):   # makespans in cost units (sharding.video_cost): what the schedule costs against the ideal
            cs = [sharding.video_cost(x) for x in lengths]
            rep = {f"world_{w}": {k: round(v, 1) for k, v in sharding.schedule_costs(cs, w).items()} for w in sorted({world, 8})}
            rep["unit"] = "sharding.video_cost units; ideal = max(mean load, largest video), lpt = max_r sum (run_scheduled), lockstep = sum of round maxima (run_sharded); world_8 is arithmetic, not a run, unless n_gpus = 8"
            return rep

        # (a) equal lengths, as in rounds 1-5, now through run_scheduled: LPT = the round-robin counts, ONE gather at the end
        lengths_eq = [T] * V30
        d30 = timed(lambda: scheduled_batch(lengths_eq))
        table = sharding.lpt_assignment([sharding.video_cost(x) for x in lengths_eq], world)
        videos30 = {"videos": V30, "value": round(V30 * N * T / d30, 1), "unit": "query-points*frames/s", "seconds": round(d30, 3),
                    "scaling": "strong", "videos_by_rank": [len(x) for x in table], "gathers": 1, "schedule": "lpt, back to back, one gather",
                    "makespan": sched_report(lengths_eq),
                    "note": "one untimed-warm step of the 30-video batch (bench.py --videos 30 is the same thing as the line's own metric)"}
        # (b) ragged lengths (DAVIS clips run 25 .. 104 frames): the same 30-video batch with five lengths cycled, LPT by frame count
        fr = sorted({max(2, int(round(T * f))) for f in (0.27, 0.45, 0.62, 0.8, 1.0)})
        lengths_rg = [fr[v % len(fr)] for v in range(V30)]
        for v in sharding.lpt_assignment([sharding.video_cost(x) for x in lengths_rg], world)[rank]:
            ctx_for(lengths_rg[v])
        if world == 1:
            scheduled_batch(lengths_rg)     # (one warm pass so that every length's buffers exist before the timed one)
        dr = timed(lambda: scheduled_batch(lengths_rg))
        videos30["ragged"] = {"lengths": fr, "frames_total": sum(lengths_rg), "value": round(N * sum(lengths_rg) / dr, 1),
                              "seconds": round(dr, 3), "makespan": sched_report(lengths_rg),
                              "frames_by_rank": [sum(lengths_rg[v] for v in x) for x in
                                                 sharding.lpt_assignment([sharding.video_cost(x) for x in lengths_rg], world)]}

    # ---- per-kernel pass (separate from the timed region): hipEvents around every launch of the library ---------------
    pairs = int(mi.last_counts[0]) if hasattr(mi, "last_counts") else 0
    maps = N * T + pairs * T
    tiers = dict(trk.last_track_stats) if trk.last_track_stats else None   # the anchor-stage dtk_track call
    ops.profile_enable(True)
    one_video(videos[0])
    prof = ops.profile_collect()
    ops.profile_enable(False)
    roofline = None
    if prof:
        HW, S = 67 * 121, 67 * 121 + 1
        depth = 12 if C in (384, 768) else 24
        from dino_tracker_amd import delta_dino as _ddm
        dd_div = 3.0 if _ddm.conv_operand_mode(None) == 0 else 1.0
        # ALGORITHMIC flops of one video per kernel, and the peak that bounds it (TFLOP/s, dense)
        algo = {
            "vit_attention": (4.0 * S * S * C * depth * T, MFMA_F16_PEAK_TF),
            "vit_gemm_qkv": (2.0 * S * C * 3 * C * depth * T, MFMA_F16_PEAK_TF),
            "vit_gemm_proj": (2.0 * S * C * C * depth * T, MFMA_F16_PEAK_TF),
            "vit_gemm_fc1": (2.0 * S * C * 4 * C * depth * T, MFMA_F16_PEAK_TF),
            "vit_gemm_fc2": (2.0 * S * C * 4 * C * depth * T, MFMA_F16_PEAK_TF),
            # split-fp16 kernels: every fp32 MAC costs three fp16 MFMA MACs (hi*hi + hi*lo + lo*hi), so their peak is a third;
            # Delta-DINO runs on PLAIN fp16 operands by default since round 4 (delta_dino.conv_operand_mode == 1): full peak then
            # (the / 3 had stayed behind and would have printed a fraction above 1)
            "vit_patch_embed": (2.0 * HW * 588 * C * T, MFMA_F16_PEAK_TF / 3),
            "dd_conv1": (2.0 * 406504 * 4800 * T, MFMA_F16_PEAK_TF / dd_div),
            "dd_conv23": (2.0 * (101626 * 204800 + 25466 * 819200) * T, MFMA_F16_PEAK_TF / dd_div),
            "dd_conv4": (2.0 * 6420 * 6400 * C * T, MFMA_F16_PEAK_TF / dd_div),
            # the escalated precision (--precision split): three MFMAs per product by construction
            "vit_attention_split": (4.0 * S * S * C * depth * T, MFMA_F16_PEAK_TF / 3),
            "vit_gemm_qkv_split": (2.0 * S * C * 3 * C * depth * T, MFMA_F16_PEAK_TF / 3),
            "vit_gemm_proj_split": (2.0 * S * C * C * depth * T, MFMA_F16_PEAK_TF / 3),
            "vit_gemm_fc1_split": (2.0 * S * C * 4 * C * depth * T, MFMA_F16_PEAK_TF / 3),
            "vit_gemm_fc2_split": (2.0 * S * C * 4 * C * depth * T, MFMA_F16_PEAK_TF / 3),
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
        # HBM-side bytes per launch of that kernel: NOT measured by this process (PMC counters need a rocprofv3 wrapper
        # around it) -- read from the newest committed PMC pass of the same command (scripts/pmc_traffic.py)
        # (the committed passes are of the DEFAULT workload: another width / frame count / query count gets no traffic figure)
        default_workload = args.width == 384 and args.frames == 90 and args.queries == 1024 and not args.videos
        for prof_file in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json") if default_workload else ():
            try:
                with open(os.path.join(ROOT, "profiles", prof_file)) as fh:
                    tr = json.load(fh)["kernels"].get(dom)
                if tr:
                    roofline["traffic"] = tr["bytes_per_launch"]
                    roofline["traffic_source"] = f"committed file profiles/{prof_file} (a separate rocprofv3 --pmc pass of this command on the builder's box), not this run"
                    roofline["traffic_note"] = "bytes per launch, rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes"
                    break
            except (OSError, ValueError, KeyError):
                pass
        # LIVE traffic of the dominant kernel when it is the attention (the figure above is a committed
        # constant): the stand-alone stage on operands of the step's shape under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
        # (two child processes, separate passes as MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 on gfx950, KiB -> bytes).  Any
        # failure (no rocprofv3, a refused counter, a timeout) leaves the committed figure in place and says so.
        if default_workload and dom == "vit_attention" and world == 1 and not args.no_live_traffic:
            import glob
            import shutil
            import tempfile
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            try:
                import pmc_traffic
                vals = {}
                for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                    td = tempfile.mkdtemp(prefix="dtk_traffic_", dir="/tmp")
                    try:
                        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", td, "-o", "t", "--", sys.executable,
                               os.path.join(ROOT, "scripts", "attn_traffic_child.py"), str(T), str(C // 64), str(S), args.operands]
                        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=240, check=True,
                                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                        dbs = glob.glob(os.path.join(td, "**", "*.db"), recursive=True)
                        pk = pmc_traffic.per_kernel(dbs[0], counter)
                        n_, v_ = next(v for k_, v in pk.items() if k_.startswith("attention") and k_.endswith("_kernel"))   # (attention6_kernel since round 6)
                        vals[counter] = (2.0 if counter == "FETCH_SIZE" else 1.0) * v_ * 1024.0 / n_
                    finally:
                        shutil.rmtree(td, ignore_errors=True)
                roofline["traffic_committed_file"] = roofline.get("traffic")
                roofline["traffic"] = round(vals["FETCH_SIZE"] + vals["WRITE_SIZE"])
                roofline["traffic_source"] = ("THIS run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child passes) of the stand-alone "
                                              "attention stage on random operands of the step's shape (scripts/attn_traffic_child.py)")
                roofline["traffic_fetch_write"] = [round(vals["FETCH_SIZE"]), round(vals["WRITE_SIZE"])]
                roofline["traffic_over_algorithmic"] = round(roofline["traffic"] / (4.0 * T * (C // 64) * S * 64 * 2), 4)
            except Exception as ex_:   # noqa: BLE001
                roofline["traffic_live_error"] = f"{type(ex_).__name__}: {str(ex_)[:200]}"
        roofline["avg_launch_ms"] = round(ms / max(launches, 1), 4)
        roofline["launches"] = launches
        roofline["kernel_ms"] = {k: round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("DTK_BENCH_KERNELS", "12"))]}
        roofline["kernel_tflops"] = {k: round(algo[k][0] / (prof[k][0] * 1e-3) / 1e12, 1) for k in prof
                                     if k in algo and prof[k][0] > 0}
        roofline["kernel_frac_of_peak"] = {k: round(algo[k][0] / (prof[k][0] * 1e-3) / 1e12 / algo[k][1], 4) for k in prof
                                           if k in algo and prof[k][0] > 0}
        # the same fractions against what the matrix pipes deliver at the MEASURED sustained clock of whole steps (the peaks above
        # assume the 2.4 GHz nameplate; the chip runs these steps at its power budget, ~2.0 GHz): every MFMA-bound kernel, not
        # only the attention loop
        sc_steps = ((clock_power or {}).get("steps_repeat", {}).get("sclk_mhz") or {}).get("p50")
        if sc_steps:
            roofline["sclk_mhz_over_steps"] = sc_steps
            roofline["kernel_frac_of_peak_at_measured_clock"] = {
                k: round(v / (sc_steps / 2400.0), 4) for k, v in roofline["kernel_frac_of_peak"].items() if algo[k][1] != F32_PEAK_TF}

    # ---- shader clock and board power DURING the dominant kernel: the stand-alone attention stage on the
    # benchmark's shapes (90 frames x 6 heads, S = 8108, random 16-bit operands), launched back to back for a few hundred ms while a
    # host thread polls the driver; the 2.5 PFLOP/s peak assumes 2.4 GHz, so frac_at_measured_clock = achieved / (peak * sclk / 2400)
    if clock_power is not None and sampler.source is not None and C == 384 and roofline and roofline.get("kernel") == "vit_attention":
        from dino_tracker_amd._lib import OPERAND_BF16, OPERAND_F16, check as _check, lib as _lib
        Sx, heads = 67 * 121 + 1, C // 64
        Spx = (Sx + 127) // 128 * 128
        odt = torch.float16 if args.operands == "fp16" else torch.bfloat16
        gq = torch.Generator(device=dev).manual_seed(5)
        qa = (torch.randn(T, heads, Spx, 64, device=dev, generator=gq) * (0.125 * 1.4426950408889634)).to(odt)
        ka = torch.randn(T, heads, Spx, 64, device=dev, generator=gq).to(odt)
        va = torch.randn(T, heads, 64, Spx, device=dev, generator=gq).to(odt)
        ka[:, :, Sx:] = 0
        va[:, :, :, Sx:] = 0
        oa = torch.empty(T, Sx, C, dtype=odt, device=dev)
        ot = OPERAND_F16 if args.operands == "fp16" else OPERAND_BF16
        launch = lambda: _check(_lib().dtk_vit_attention(ops._p(qa), ops._p(ka), ops._p(va), ops._p(oa), T, heads, Sx, Spx, ot, ops._stream()))  # noqa: E731
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        nl = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sampler.start()
        e0.record()
        for _ in range(nl):
            launch()
        e1.record()
        torch.cuda.synchronize()
        cp = sampler.stop()
        ms_l = e0.elapsed_time(e1) / nl
        tf = 4.0 * Sx * Sx * C * T / (ms_l * 1e-3) / 1e12
        cp.update({"launches": nl, "avg_launch_ms": round(ms_l, 4), "achieved_tflops": round(tf, 1),
                   "note": "dtk_vit_attention alone, back to back, random operands of the benchmark's shape"})
        sc = (cp.get("sclk_mhz") or {}).get("p50")
        if sc:
            cp["frac_of_peak_at_measured_clock"] = round(tf / (MFMA_F16_PEAK_TF * sc / 2400.0), 4)
        clock_power["attention_loop"] = cp
        del qa, ka, va, oa

    # ---- CPU baseline + parity sample: the oracle (torch fp32 port of the reference algorithm) on K queries at
    # full T with all their anchors, one frame of its ViT and of its Delta-DINO, combined by
    #     cpu_qpf = N T / (T t_vit + T t_delta + N t_query).
    # The same K queries are then tracked by the HIP path on the same refined volume: `parity_sample`.
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and "track" in stages:
        from oracle import ref_algo as A
        nq = max(1, min(N, args.cpu_queries))
        npar = max(nq, min(N, args.cpu_parity_queries))
        sel = torch.linspace(0, N - 1, npar).long()
        timed = torch.linspace(0, npar - 1, nq).long()          # the timed K queries, spread over the parity sample
        rest = torch.tensor([i for i in range(npar) if i not in set(timed.tolist())], dtype=torch.long)
        q_cpu = queries.cpu()[sel]
        one_video(videos[0])                                 # the state the sample is compared against
        refined_cpu = trk.refined_features.cpu()
        # torch's CPU kernels on maps of 67 x 121 lose time to thread fan-out on a many-core host (round 3: 3 x slower on 128
        # threads than on 8).  The infer leg runs on a FIXED thread count (--cpu-threads, default 8) so that the baseline is
        # comparable from round to round; the sweep over thread counts of a slice of the first pass is reported beside it.
        ncore = torch.get_num_threads()
        probe_src = refined_cpu[0].reshape(C, -1).t()[:128].contiguous()
        sweep = {}
        for thr in sorted({ncore, 64, 32, 16, 8}):
            if thr > ncore:
                continue
            torch.set_num_threads(thr)
            c0 = time.perf_counter()
            A.track(probe_src, refined_cpu, torch.zeros(128, dtype=torch.long), head, H, W)
            sweep[str(thr)] = round(time.perf_counter() - c0, 3)
        # ONE stated thread count for all three legs (infer ran on 8 threads, ViT / Delta-DINO on all):
        # --cpu-threads N > 0 fixes it; 0 (default) = the sweep's fastest -- the baseline is then the best this host does
        best_thr = max(1, min(args.cpu_threads, ncore)) if args.cpu_threads > 0 else int(min(sweep, key=lambda k: sweep[k]))
        torch.set_num_threads(best_thr)
        c0 = time.perf_counter()
        rt_t, ro_t, cs_t, _ = A.infer(refined_cpu, q_cpu[timed], head, H, W, return_aux=True)
        t_query = (time.perf_counter() - c0) / nq
        a_bar = float((cs_t >= 0.7).sum()) / nq
        rt = torch.zeros(npar, T, 2)
        ro = torch.zeros(npar, T, dtype=torch.bool)
        cs_all = torch.zeros(npar, T)
        rt[timed], ro[timed], cs_all[timed] = rt_t, ro_t, cs_t
        if rest.numel():
            r2, o2, c2, _ = A.infer(refined_cpu, q_cpu[rest], head, H, W, return_aux=True)
            rt[rest], ro[rest], cs_all[rest] = r2, o2, c2
        nf = max(1, min(T, args.cpu_vit_frames))   # (the ViT / Delta-DINO legs stay on best_thr threads: one stated count)
        vcpu = videos[0][:nf].cpu()
        c0 = time.perf_counter()
        dino_cpu = torch.stack([A.vit_tokens(vcpu[i:i + 1], vit_sd, model_name) for i in range(nf)]) if args.features == "vit" else None
        t_vit = (time.perf_counter() - c0) / nf if dino_cpu is not None else 0.0
        from_video = None
        if dino_cpu is None:
            dino_cpu = trk.dino_embed_video[:nf].cpu()
        c0 = time.perf_counter()
        refined_nf = A.refine_features(vcpu, dino_cpu, delta)
        t_delta = (time.perf_counter() - c0) / nf
        torch.set_num_threads(ncore)
        total = T * t_vit + T * t_delta + N * t_query
        cpu = {"value": round(N * T / total, 3), "unit": "query-points*frames/s", "cores": best_thr,
               "kind": "port",
               "sample": f"oracle (fp32 torch restatement of the reference; the un-modified reference is not on this box, its "
                         f"literal per-call path is ~T x more expensive: profiles/r03_cpu_reference_literal.json): "
                         f"infer on {nq} of the {N} queries at full T={T} with all their anchors ({a_bar:.1f} per query) "
                         f"{t_query:.2f} s/query; ViT {t_vit:.2f} s/frame and Delta-DINO {t_delta:.2f} s/frame on {nf} frame(s); ALL legs on "
                         f"{best_thr} threads (the fastest of the sweep of a 128-map slice, thread_sweep_s, unless --cpu-threads); "
                         f"N T / (T t_vit + T t_delta + N t_query), SURVEY 8d",
               "t_query_s": round(t_query, 3), "t_vit_s": round(t_vit, 3), "t_delta_s": round(t_delta, 3),
               "anchors_per_query": round(a_bar, 2), "host_threads": ncore, "thread_sweep_s": sweep,
               "literal_over_port": literal_over_port(C),
               "literal_over_port_source": "committed file profiles/r03_cpu_reference_literal.json: measured in round 3 in the BUILD "
                                           "container (where the un-modified reference runs) on that container's cores -- not on this box"}
        tg, og = mi.infer(queries[sel.to(dev)])
        host_leg = {"queries": npar, "frames": T, "correlation_maps": int(npar * T + float((cs_all >= 0.7).sum()) * T),
                    "max_dxy_px": round(float((tg.cpu() - rt).abs().max()), 6),
                    "occ_mismatch": int((og.cpu() != ro).sum()), "occ_flags": int(ro.numel()),
                    "note": "HIP infer vs the oracle ON THE HOST (the form pinned on the reference), same refined volume"}
        parity = dict(host_leg, tiers=dict(trk.last_track_stats))
        if args.parity_queries > 0:
            # EVERY query: the same restatement on device tensors in fp32 (oracle/ref_algo.py takes its device
            # from its inputs; no TF32 on gfx950, matmul precision "highest"), on the refined volume the timed step produced
            torch.backends.cuda.matmul.allow_tf32 = False
            torch.set_float32_matmul_precision("highest")
            nfull = min(N, args.parity_queries)
            selg = torch.linspace(0, N - 1, nfull).long().to(dev)
            head_g = {k: v.to(dev) for k, v in head.items()}
            refined_g = trk.refined_features
            c0 = time.perf_counter()
            gt_, go_, gcs_, _ = A.infer(refined_g, queries[selg], head_g, H, W, return_aux=True)
            torch.cuda.synchronize()
            t_oracle = time.perf_counter() - c0
            th, oh = mi.infer(queries[selg])
            err = (th - gt_).norm(dim=-1)
            flagged = torch.nonzero(err > 1e-3).tolist()
            arb = []
            for n_, t_ in flagged[:64]:     # an fp32 near-tie of the map decided the other way, or a failure: float64 arbiter, fp32 band only
                r_ = A.tie_arbiter(refined_g, queries[selg][n_], t_, th[n_, t_], head_g, H, W, A.fp32_dot_band(C))
                arb.append({"query": n_, "frame": t_, "err_px": round(float(err[n_, t_]), 4), "gap64": r_["gap64"],
                            "band": r_["delta"], "dist_px": r_["dist_px"], "ok": bool(r_["ok"] and r_["gap64"] <= r_["delta"])})
            clean = torch.ones(nfull, dtype=torch.bool, device=dev)
            clean[[a["query"] for a in arb if a["ok"]]] = False
            # GPU-torch against CPU-torch on the host leg's queries
            pin = (gt_[torch.searchsorted(selg, sel.to(dev))].cpu() - rt).abs().max() if nfull == N else None
            parity = {"queries": nfull, "frames": T, "positions": int(nfull * T),
                      "correlation_maps": int(nfull * T + float((gcs_ >= 0.7).sum()) * T),
                      "max_dxy_px": round(float(err[err <= 1e-3].max()) if bool((err <= 1e-3).any()) else -1.0, 6),
                      "points_beyond_1e-3px": len(flagged), "arbitrated_fp32_near_ties": arb,
                      "mismatches": sum(0 if a["ok"] else 1 for a in arb) + max(0, len(flagged) - 64),
                      "occ_mismatch": int((oh != go_).sum()),
                      "occ_mismatch_queries_without_a_tie": int((oh[clean] != go_[clean]).sum()), "occ_flags": int(go_.numel()),
                      "oracle": f"oracle/ref_algo.py infer on cuda tensors, fp32, {t_oracle:.1f} s",
                      "oracle_cuda_vs_host_max_dxy_px": None if pin is None else round(float(pin), 6),
                      "host_leg": host_leg, "tiers": dict(trk.last_track_stats),
                      "note": "HIP infer vs oracle infer on the same refined volume (the one the timed step produced), every query"}
        if args.features == "vit" and nf >= 2:
            # from the VIDEO: the first nf frames through oracle ViT -> oracle refine -> oracle infer against the device's
            # ViT -> Delta-DINO -> infer on the same frames (what the 16-bit operands of P1 add; tests assert it at T = 8)
            dev_feat = trk.dino_embed_video[:nf].cpu()
            f_rel = float((dev_feat - dino_cpu).norm() / dino_cpu.norm())
            qv = q_cpu.clone()
            qv[:, 2] = 0
            rv, ov = A.infer(refined_nf, qv, head, H, W)
            trk2 = Tracker(video=videos[0][:nf], dino_features=ex.encode(videos[0][:nf]), dino_patch_size=14, stride=7,
                           device=dev, track_method=method)
            trk2.tracker_head.load_state_dict(head)
            trk2.delta_dino.load_state_dict(delta)
            trk2.to(dev).eval()
            mi2 = ModelInference(trk2, RangeNormalizer((W, H, nf), device=dev), 0.7, 0.6)
            tv, ovd = mi2.infer(qv.to(dev))
            # the LAST frame of the timed step's own 90-frame encoder pass (the first nf frames above sit at the start of it)
            last_cpu = A.vit_tokens(videos[0][T - 1:T].cpu(), vit_sd, model_name)
            last_dev = trk.dino_embed_video[T - 1].cpu()
            f_rel_last = float((last_dev - last_cpu).norm() / last_cpu.norm())
            parity["from_video"] = {"frames": nf, "queries": npar, "feature_rel_err_P1": round(f_rel, 7),
                                    "feature_rel_err_P1_last_frame_of_the_pass": round(f_rel_last, 7),
                                    "max_dxy_px": round(float((tv.cpu() - rv).abs().max()), 6),
                                    "occ_mismatch": int((ovd.cpu() != ov).sum()),
                                    "note": "video -> HIP ViT/Delta-DINO/infer vs video -> oracle ViT/refine/infer"}

        if parity is not None and args.features == "vit" and args.parity_queries > 0 and args.parity_video_frames > 0:
            # from the VIDEO at full length on the GPU oracle: oracle ViT -> oracle refine -> oracle infer vs the step's own results
            nfv = min(T, args.parity_video_frames)
            sd_g = {k: v.to(dev) for k, v in vit_sd.items()}
            delta_g = {k: v.to(dev) for k, v in delta.items()}
            c0 = time.perf_counter()
            dino_o = torch.stack([A.vit_tokens(videos[0][t:t + 1], sd_g, model_name) for t in range(nfv)])
            refined_o = A.refine_features(videos[0][:nfv], dino_o, delta_g)
            qv = queries[selg].clone()
            ot, oo = A.infer(refined_o, qv, head_g, H, W)
            torch.cuda.synchronize()
            t_or = time.perf_counter() - c0
            if nfv == T:
                one_video(videos[0])
                dv_ref, (tv_, ov_) = trk.refined_features, mi.infer(qv)
            else:
                trk3 = Tracker(video=videos[0][:nfv], dino_features=ex.encode(videos[0][:nfv]), dino_patch_size=14, stride=7,
                               device=dev, track_method=method)
                trk3.tracker_head.load_state_dict(head)
                trk3.delta_dino.load_state_dict(delta)
                trk3.to(dev).eval()
                mi3 = ModelInference(trk3, RangeNormalizer((W, H, nfv), device=dev), 0.7, 0.6)
                dv_ref, (tv_, ov_) = trk3.refined_features, mi3.infer(qv)
            e2 = (tv_ - ot).norm(dim=-1)
            fl = torch.nonzero(e2 > 1e-3).tolist()
            arbv = []
            for n_, t_ in fl[:128]:   # band: fp32 rounding + the MEASURED deviation of the two cosines in question (tie_arbiter)
                r_ = A.tie_arbiter(refined_o, qv[n_], t_, tv_[n_, t_], head_g, H, W, A.fp32_dot_band(C), dev_feats=dv_ref)
                arbv.append({"query": n_, "frame": t_, "err_px": round(float(e2[n_, t_]), 4), "gap64": r_["gap64"], "band": r_["delta"],
                             "dist_px": r_["dist_px"], "ok": bool(r_["ok"] and r_["gap64"] <= r_["delta"])})
            cl = torch.ones(len(qv), dtype=torch.bool, device=dev)
            cl[[a["query"] for a in arbv if a["ok"]]] = False
            ok_e = e2[e2 <= 1e-3]
            parity["from_video_every_query"] = {
                "frames": nfv, "queries": len(qv), "positions": int(e2.numel()),
                "feature_rel_err_refined": round(float((dv_ref - refined_o).norm() / refined_o.norm()), 7),
                "max_dxy_px": round(float(ok_e.max()), 6), "p99_dxy_px": round(float(e2.flatten().quantile(0.99)), 6),
                "points_beyond_1e-3px": len(fl), "arbitrated_near_ties": arbv[:16],
                "mismatches": sum(0 if a["ok"] else 1 for a in arbv) + max(0, len(fl) - 128),
                "occ_mismatch": int((ov_ != oo).sum()), "occ_mismatch_queries_without_a_tie": int((ov_[cl] != oo[cl]).sum()),
                "oracle": f"oracle ViT -> refine -> infer on cuda tensors, fp32, {t_or:.1f} s",
                "note": "video -> HIP ViT/Delta-DINO/infer vs video -> oracle ViT/refine/infer, every query and frame"}

    if rank == 0:
        from dino_tracker_amd import delta_dino as _dd
        p2_mode = {0: "split-f16 conv operands (fp32-grade)", 1: "f16 conv operands"}[_dd.conv_operand_mode(None)]
        out = {
            "metric": "query-points*frames/s", "value": round(frames_per_step * N * args.steps / dt, 1),
            "unit": "query-points*frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if (qpar or args.videos > 0) else "weak",
            "vs_baseline": None, "dtype": (f"mixed: {ex.operand_dtype} ViT operands ({'hi + lo split in ' + str(len(ex.split_blocks)) + ' blocks' if ex.split_blocks else 'one per value'}), {p2_mode} in Delta-DINO, " + ("f32 tracker" if method == ops.TRACK_EXACT
                                                                       else "f16 candidates + f32 deciders in the tracker")
                      + "; f32 accumulate"),
            "data": "synthetic",
            "config": {"workload": f"854x480x{T} synthetic video (model res 854x476, 67x121 tokens, C={C}), {N} grid "
                                   f"queries; {videos_per_step} video(s) per step over {world} GPU(s)",
                       "stages": stages, "features": args.features, "mode": args.mode if world > 1 else "single",
                       "track_method": "exact" if method == ops.TRACK_EXACT else "mfma",
                       "anchor_pairs": pairs, "correlation_maps_per_video": maps, "anchor_track_tiers": tiers,
                       "rccl_ranks": rccl_ranks, "vit_precision": ex.precision_report(),
                       "vit_frames_per_rank": ([sharding.split_range(T, r, world)[1] - sharding.split_range(T, r, world)[0]
                                                for r in range(world)] if qpar else [T] * world),
                       "parallelism": (f"query-parallel x{world} (frames split for P1/P2, queries for P3)" if qpar
                                       else f"video-parallel x{world}")},
            "videos30": videos30, "per_rank": per_rank, "roofline": roofline, "clock_power": clock_power, "cpu_baseline": cpu, "parity_sample": parity,
        }
        if world == 1 and not args.no_train:
            out["train"] = train_leg(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
