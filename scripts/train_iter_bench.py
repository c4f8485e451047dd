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


def run_mode(yml, data, mode, iters, warm, seed=2):
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
    c0 = dict(step.counts)
    per = []
    t0 = time.perf_counter()
    for i in range(i0 + warm, i0 + warm + iters):
        it(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    # a second pass with a synchronisation per iteration: the spread of single iterations (jitter), not used for the headline
    for i in range(i0 + warm + iters, i0 + warm + iters + min(iters, 20)):
        t1 = time.perf_counter()
        it(i)
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t1)
    losses = torch.stack(vals).cpu()
    counts = {k: step.counts[k] - c0[k] for k in step.counts}
    model.train(False)
    del step, model, opt, tr
    torch.cuda.empty_cache()
    per.sort()
    return {"s_per_iteration": dt, "timed_iterations": iters, "timed_counts": counts,
            "single_iteration_s": {"min": per[0], "median": per[len(per) // 2], "max": per[-1]},
            "first_total": float(losses[:5, 0].mean()), "last_total": float(losses[-5:, 0].mean()),
            "all_finite": bool(torch.isfinite(losses).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--frames", type=int, default=90)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--warm", type=int, default=12, help="untimed iterations in front (the first of a batch shape runs eagerly, the second is captured)")
    ap.add_argument("--modes", default="graph,eager")
    ap.add_argument("--operands", default="", help="DTK_TRAIN_CONV_OPERANDS (split | fp16); default: the library's")
    ap.add_argument("--data-dir", default="")
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
        out[mode] = run_mode(yml, d, mode, a.iters, a.warm)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
