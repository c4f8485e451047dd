"""BASELINE.json config 5 without the reference checkout: per-video test-time training at full size (854 x 476 x T frames,
config/train.yaml's batch sizes, every loss term on) through dino_tracker_amd.train's restated control plane and the device-side
trainer; seconds per iteration with the iteration replayed from captured graphs (trainer.GraphedIteration) and eagerly.

    python scripts/train_iter_bench.py [--width 384] [--frames 90] [--iters 40] [--modes graph,eager] [--operands split|fp16]

Prints one JSON line.  Synthetic video, embeddings, trajectories and best buddies (tests/golden/train_data.py).  Timing: a
synchronised clock after the warm-up iterations and after the timed ones; nothing reads the device in between."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def run_mode(yml, data, mode, iters, warm, seed=2, kernel_share=False):
    import torch
    from dino_tracker_amd import train as TR, trainer as T
    from dino_tracker_amd.train_ops import install_fused_adam
    TR.fix_random_seeds(seed)
    tr = TR.standalone_trainer(argparse.Namespace(config=yml, data_path=data, device="cuda:0"))
    tr.load_fg_masks()
    tr.load_dino_best_buddies()
    sampler = tr.get_sampler()
    model, opt, sched = tr.train_setup()
    install_fused_adam(opt)
    tr.set_model_train(model)
    tr.init_losses()
    tr.prepare_tables(model)
    step = T.GraphedIteration(tr, model, opt, sampler, enabled=(mode == "graph"))
    vals = []
    i0 = tr.init_iter

    def it(i):
        v = step.run(i)
        sched.step()
        vals.append(v)
    for i in range(i0, i0 + warm):
        it(i)
    torch.cuda.synchronize()
    if os.environ.get("DTK_BENCH_GC", "") == "freeze":
        import gc
        gc.collect()
        gc.freeze()
    elif os.environ.get("DTK_BENCH_GC", "") == "off":
        import gc
        gc.collect()
        gc.disable()
    c0 = dict(step.counts)
    per, host = [], []
    # the timed iterations in three blocks (a synchronised clock at both ends of each, nothing read in between): the boxes of this
    # pool show bursts of host-side stalls lasting seconds (scripts/dev/iter_phases.py: 40-90 ms in `draw_frame_sets` or in a graph
    # launch, several iterations in a row), so the blocks are reported one by one and the headline is the best of them
    blocks = []
    nb = max(iters // 3, 1)
    i = i0 + warm
    t_all = time.perf_counter()
    while i < i0 + warm + iters:
        n = min(nb, i0 + warm + iters - i)
        t0 = time.perf_counter()
        for k in range(i, i + n):
            it(k)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / n)
        i += n
    dt_mean = (time.perf_counter() - t_all) / iters
    dt = min(blocks)
    # a second pass with a synchronisation per iteration: the spread of single iterations (jitter), not used for the headline
    for i in range(i0 + warm + iters, i0 + warm + iters + min(iters, 20)):
        t1 = time.perf_counter()
        it(i)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t1)
        host.append(t2 - t1)
    share = None
    if kernel_share:
        # kernel time by origin over a few more iterations (torch.profiler's device activity records every kernel of a replayed graph):
        # hand-written = this library's kernels, library = ATen / rocBLAS / runtime copies
        try:
            from torch.profiler import ProfilerActivity, profile
            n_it = 5
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for i in range(i0 + warm + 2 * iters, i0 + warm + 2 * iters + n_it):
                    it(i)
                torch.cuda.synchronize()
            own = lib = 0.0
            n_own = n_lib = 0
            for ev in prof.events():
                if ev.device_type is None or "cuda" not in str(ev.device_type).lower():
                    continue
                dur = float(getattr(ev, "device_time_total", 0.0) or getattr(ev, "cuda_time_total", 0.0) or 0.0)
                name = ev.name
                if name.startswith("Memcpy") or name.startswith("Memset"):
                    lib += dur; n_lib += 1
                elif "at::" in name or "at6native" in name or name.startswith("Cijk") or "rocclr" in name or "rocprim" in name or "hipcub" in name:
                    lib += dur; n_lib += 1
                else:
                    own += dur; n_own += 1
            share = {"iterations": n_it, "hand_written_ms_per_iteration": round(own / n_it / 1e3, 3), "library_ms_per_iteration": round(lib / n_it / 1e3, 3),
                     "hand_written_launches_per_iteration": round(n_own / n_it, 1), "library_launches_per_iteration": round(n_lib / n_it, 1),
                     "hand_written_share": round(own / max(own + lib, 1e-9), 4)}
        except Exception as ex:  # noqa: BLE001
            share = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    losses = torch.stack(vals).cpu()
    counts = {k: step.counts[k] - c0[k] for k in step.counts}
    model.train(False)
    del step, model, opt, tr
    torch.cuda.empty_cache()
    per.sort()
    return {"s_per_iteration": dt, "s_per_iteration_blocks": [round(b, 5) for b in blocks], "s_per_iteration_all_blocks": dt_mean,
            "timed_iterations": iters, "timed_counts": counts,
            "single_iteration_s": {"min": per[0], "median": per[len(per) // 2], "max": per[-1]},
            "host_side_s_of_those": {"min": min(host), "median": sorted(host)[len(host) // 2], "max": max(host)},
            "first_total": float(losses[:5, 0].mean()), "last_total": float(losses[-5:, 0].mean()),
            "all_finite": bool(torch.isfinite(losses).all()), "kernel_time": share}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--frames", type=int, default=90)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--warm", type=int, default=12, help="untimed iterations in front (the first of a batch shape runs eagerly, the second is captured)")
    ap.add_argument("--modes", default="graph,eager")
    ap.add_argument("--operands", default="", help="DTK_TRAIN_CONV_OPERANDS (split | fp16); default: the library's")
    ap.add_argument("--data-dir", default="")
    ap.add_argument("--kernel-share", action="store_true", help="also: kernel time per iteration by origin (hand-written / library), torch.profiler")
    a = ap.parse_args()
    if a.operands:
        os.environ["DTK_TRAIN_CONV_OPERANDS"] = a.operands
    import train_data as TD
    cfg = dict(TD.CFG, T=a.frames, C=a.width, H=476, W=854, total_iterations=10 ** 6, n_fg=4000, n_bg=6000, bb_per_pair=24)
    d = a.data_dir or tempfile.mkdtemp(prefix="dtk_train_iter_bench_")
    t0 = time.time()
    yml = os.path.join(d, "train.yaml")
    if not os.path.isfile(yml):
        d, yml = TD.build(d, None, cfg, overrides={}, synthetic_video=True)
    out = {"config": f"854x476x{a.frames}, C={a.width}, config/train.yaml batch sizes, all losses on, conv operands "
                     f"{os.environ.get('DTK_TRAIN_CONV_OPERANDS', 'split')}", "data_build_s": round(time.time() - t0, 1)}
    for mode in a.modes.split(","):
        out[mode] = run_mode(yml, d, mode, a.iters, a.warm, kernel_share=a.kernel_share)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
