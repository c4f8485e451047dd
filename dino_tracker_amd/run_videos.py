"""Video-parallel launcher for the reference's per-video scripts (train.py, inference_grid.py, inference_benchmark.py):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m dino_tracker_amd.run_videos \\
        [--path DIR]... /path/to/dino-tracker/train.py --data-paths D0 D1 ... D29 -- --config config/train.yaml

Videos (and their test-time training runs) are independent (SURVEY.md section 8e level 1, BASELINE.json configs 4 and 5):
rank r takes data paths v = r (mod world) and runs `python -m dino_tracker_amd.run <script> --data-path Dv <args>` for
each, one fresh process per video, pinned to GPU LOCAL_RANK through HIP_VISIBLE_DEVICES (the reference's scripts address
"cuda:0").  Results are the files the scripts write under each data path; nothing is exchanged between ranks, so there
is no collective and no process group -- RANK / LOCAL_RANK / WORLD_SIZE are only read for the partition.  Without a
launcher (no WORLD_SIZE) it runs every video on one GPU.  Exit status: non-zero if any video failed.
"""
from __future__ import annotations

import os
import subprocess
import sys

from .sharding import videos_of_rank


def parse(argv):
    extra = []
    while argv and argv[0] == "--path":
        extra += argv[:2]
        argv = argv[2:]
    if not argv or argv[0].startswith("-") or "--data-paths" not in argv:
        raise SystemExit("usage: python -m dino_tracker_amd.run_videos [--path DIR]... script.py --data-paths D0 D1 ... "
                         "[-- script arguments]")
    script, rest = argv[0], argv[1:]
    i = rest.index("--data-paths")
    j = rest.index("--") if "--" in rest else len(rest)
    paths = rest[i + 1:j]
    script_args = rest[:i] + rest[j + 1:]
    if not paths:
        raise SystemExit("dino_tracker_amd.run_videos: --data-paths is empty")
    return extra, script, paths, script_args


def pin_local_gpu(env: dict, local: int) -> None:
    """One GPU per rank: entry `local` of whatever device list the job was already restricted to (a scheduler's
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES), or plain index `local` when there is none."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        cur = [d for d in env.get(var, "").split(",") if d.strip()]
        if cur:
            if local >= len(cur):
                raise SystemExit(f"dino_tracker_amd.run_videos: LOCAL_RANK {local} but {var}={env[var]}")
            env[var] = cur[local].strip()
            for other in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
                if other != var:
                    env.pop(other, None)
            return
    # ROCR_VISIBLE_DEVICES narrows what HIP enumerates: HIP indices are relative to it, so `local` is already right
    env["HIP_VISIBLE_DEVICES"] = str(local)


def main(argv) -> int:
    extra, script, paths, script_args = parse(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    env = dict(os.environ)
    if world > 1:
        pin_local_gpu(env, local)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK",
              "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)  # the per-video process is a plain single-GPU run
    failed = []
    for v in videos_of_rank(len(paths), rank, world):
        cmd = [sys.executable, "-m", "dino_tracker_amd.run"] + extra + [script, "--data-path", paths[v]] + script_args
        print(f"[rank {rank}/{world}] video {v}: {paths[v]}", flush=True)
        if subprocess.run(cmd, env=env).returncode != 0:
            failed.append(paths[v])
    if failed:
        print(f"[rank {rank}] failed: {failed}", file=sys.stderr, flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
