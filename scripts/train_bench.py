"""BASELINE.json config 5: per-video test-time training at full size (854 x 476 x T frames, config/train.yaml's batch sizes,
every loss term on), the reference's train.py UN-MODIFIED:

    --side hip        through `python -m dino_tracker_amd.run` on the device (this implementation)
    --side reference  the reference's own PyTorch code on the host cores (HIP_VISIBLE_DEVICES="")

Prints one JSON line: seconds per iteration (median over the iterations after the first two, which carry the library's
kernel selection), iterations, embedding width.  Synthetic video, embeddings, trajectories and best buddies
(tests/golden/train_data.py); the reference checkout comes from $DTK_REFERENCE_ROOT.

The reference hard-codes a 1024-wide Delta-DINO (models/networks/delta_dino.py:9), so its side only runs at --width 1024.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", choices=["hip", "reference"], default="hip")
    ap.add_argument("--frames", type=int, default=90)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--profile-dir", default="")
    ap.add_argument("--rng-shim", action="store_true", help="keep train_driver's host-side random draws (parity runs; slow)")
    ap.add_argument("--trainer", choices=["device", "reference"], default="device",
                    help="hip side: dino_tracker_amd/trainer.py's iteration (default) or the reference's own loop on this implementation's models")
    ap.add_argument("--keep-stderr", default="", help="write the training process's stderr here")
    ap.add_argument("--keep-losses", default="", help="copy the per-iteration loss log (JSON) here")
    ap.add_argument("--data-dir", default="", help="reuse / create the synthetic inputs here (shared between the two sides)")
    a = ap.parse_args()
    ref = os.environ.get("DTK_REFERENCE_ROOT", "/root/reference")
    import train_data as TD
    cfg = dict(TD.CFG, T=a.frames, C=a.width, H=476, W=854, total_iterations=1 + a.iters, n_fg=4000, n_bg=6000, bb_per_pair=24)
    tmp = tempfile.mkdtemp(prefix="dtk_train_bench_")
    t0 = time.time()
    d = a.data_dir or os.path.join(tmp, "data")
    yml = os.path.join(d, "train.yaml")
    if not os.path.isfile(yml):
        d, yml = TD.build(d, ref, cfg, overrides={}, synthetic_video=True)
    else:  # a previous run's checkpoints would change the start iteration
        ck = os.path.join(d, "models", "dino_tracker")
        for f in os.listdir(ck):
            if not f.endswith(f"_{TD.CFG['start_iter']}.pt"):
                os.remove(os.path.join(ck, f))
        import yaml
        with open(yml) as fh:
            conf = yaml.safe_load(fh.read())
        conf["total_iterations"] = 1 + a.iters
        with open(yml, "w") as fh:
            yaml.safe_dump(conf, fh)
    t_build = time.time() - t0
    log = os.path.join(tmp, "losses.json")
    drv = os.path.join(ROOT, "tests", "golden", "train_driver.py")
    shims = os.path.join(ROOT, "oracle", "shims")
    tail = [os.path.join(ref, "train.py"), "--config", yml, "--data-path", d, "--seed", "2"]
    env = dict(os.environ, DTK_TRAIN_LOG=log)
    if not a.rng_shim:
        env["DTK_TRAIN_NO_RNG_SHIM"] = "1"  # time the loop with torch's own random generators (see train_driver.py)
    if a.side == "hip":
        cmd = [sys.executable, "-m", "dino_tracker_amd.run", "--path", shims, "--path", ref, drv] + tail
        env["PYTHONPATH"] = ROOT
        env["DTK_TRAINER"] = a.trainer
        if a.trainer == "device":
            env["DTK_TRAIN_ASYNC_LOG"] = "1"
    else:
        cmd = [sys.executable, drv] + tail
        env["PYTHONPATH"] = os.pathsep.join([shims, ref, ROOT])
        env["HIP_VISIBLE_DEVICES"] = ""
        env["CUDA_VISIBLE_DEVICES"] = ""
    if a.profile_dir:
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", a.profile_dir, "-o", "train", "--"] + cmd
    t0 = time.time()
    r = subprocess.run(cmd, env=env, cwd=ref, capture_output=True, text=True)
    wall = time.time() - t0
    if a.keep_stderr:
        with open(a.keep_stderr, "w") as fh:
            fh.write(r.stderr)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        raise SystemExit(r.returncode)
    with open(log) as fh:
        rec = json.load(fh)
    if a.keep_losses:
        with open(a.keep_losses, "w") as fh:
            json.dump(rec, fh)
    st = rec["seconds"]
    per = [b - a_ for a_, b in zip(st[:-1], st[1:])]
    steady = per[1:] if len(per) > 2 else per
    median = statistics.median(steady)
    timing = "median of per-iteration wall clock (every iteration ends with host reads of its losses)"
    if len(rec.get("synced", [])) == 2:  # device-side trainer: no host read per iteration; synchronised clock at both ends
        (n0, t0s), (n1, t1s) = rec["synced"]
        median = (t1s - t0s) / (n1 - n0)
        timing = f"(synchronised clock after iteration {n1} - after iteration {n0}) / {n1 - n0}; no host read in between"
    import torch
    print(json.dumps({"config": f"train.py un-modified, 854x476x{a.frames}, C={a.width}, config/train.yaml batch sizes, all losses on",
                      "side": a.side, "trainer": a.trainer if a.side == "hip" else "reference", "timing": timing,
                      "iterations": len(st), "s_per_iteration_median": median,
                      "s_per_iteration_all": [round(p, 4) for p in per], "wall_s": round(wall, 1),
                      "data_build_s": round(t_build, 1), "host_threads": torch.get_num_threads(),
                      "final_losses": dict(zip(rec["names"], rec["losses"][-1]))}))


if __name__ == "__main__":
    main()
