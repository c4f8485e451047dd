"""TEST INFRASTRUCTURE -- runs the reference's train.py UNCHANGED (runpy) with two pieces of instrumentation, the same on
both sides of the comparison (the reference's own modules on CPU when make_golden.py builds the fixture; this
implementation on the device under `python -m dino_tracker_amd.run` in tests/test_gpu_train.py):

  * every random draw comes from torch's CPU generator: randperm / randint with a device argument and Tensor.multinomial
    are computed on the host and moved, so that the CPU and the device run see the same indices ($DTK_TRAIN_NO_RNG_SHIM=1
    leaves torch's generators alone: scripts/train_bench.py times the loop that way -- the shim's host-side randperm of
    ~400 k elements, 56 times per iteration, was 165 ms of round 2's "0.20 s per iteration");
  * DINOTracker.update_losses also appends its arguments (the seven loss values of the iteration) to a list that is written
    to $DTK_TRAIN_LOG as JSON at exit -- train() itself only logs every 100th iteration.
On a machine without a GPU `Tensor.cuda()` is the identity (dino_tracker.py and models/utils.py call it unconditionally)
and the reference's RangeNormalizer gets 'cpu' as its default device (data/dataset.py:15; the same patch as
oracle/ref_harness.py), as does models/utils.py:87's grid helper.

    python train_driver.py <reference>/train.py --config C --data-path D --seed S
"""
import atexit
import json
import os
import runpy
import sys
import time

import torch

_randperm, _randint, _multinomial = torch.randperm, torch.randint, torch.Tensor.multinomial


def randperm(n, *args, device=None, **kw):
    out = _randperm(n, *args, **kw)
    return out if device is None else out.to(device)


def randint(*args, device=None, **kw):
    out = _randint(*args, **kw)
    return out if device is None else out.to(device)


def multinomial(self, *args, **kw):
    return _multinomial(self.cpu(), *args, **kw).to(self.device)


if not os.environ.get("DTK_TRAIN_NO_RNG_SHIM"):  # timing runs keep torch's own generators (device draws stay on the device)
    torch.randperm, torch.randint, torch.Tensor.multinomial = randperm, randint, multinomial
if not torch.cuda.is_available():
    torch.Tensor.cuda = lambda self, *a, **k: self

LOSSES = []
STAMPS = []  # wall clock at the end of every iteration (after its .item() reads: the device is idle)
SYNCED = []  # (iteration count, wall clock behind a device synchronisation) -- $DTK_TRAIN_ASYNC_LOG runs
ASYNC = bool(os.environ.get("DTK_TRAIN_ASYNC_LOG"))


def _dump():
    path = os.environ.get("DTK_TRAIN_LOG")
    if path:
        with open(path, "w") as fh:
            losses = [v.tolist() if isinstance(v, torch.Tensor) else v for v in LOSSES]
            json.dump({"names": ["total", "of", "cl_dino_bb", "cl_refiner", "emb_norm_reg", "angle_reg", "cyc"],
                       "losses": losses, "seconds": STAMPS, "synced": SYNCED}, fh)


atexit.register(_dump)

if __name__ == "__main__":
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    if not torch.cuda.is_available():
        import data.dataset
        if data.dataset.RangeNormalizer.__init__.__defaults__ == ("cuda",):
            data.dataset.RangeNormalizer.__init__.__defaults__ = ("cpu",)
        import models.utils  # get_vit_feature_coords_from_mask(..., device="cuda") (models/utils.py:87)
        models.utils.get_vit_feature_coords_from_mask.__defaults__ = (7, 14, "cpu")
    import dino_tracker  # the reference's trainer, or overlay/dino_tracker.py (its subclass with the device-side iteration)

    _update = dino_tracker.DINOTracker.update_losses

    tprof_path = os.environ.get("DTK_TRAIN_TORCHPROF")
    PROF = {}

    def update_losses(self, *vals):
        if os.environ.get("DTK_TRAIN_SYNC_DEBUG") and torch.cuda.is_available() and len(LOSSES) == 3:
            import warnings
            warnings.simplefilter("always")
            torch.cuda.set_sync_debug_mode("warn")  # every synchronising call of the following iterations, with its location
        if ASYNC and isinstance(vals[0], torch.Tensor) and vals[0].is_cuda:
            # timing runs of the device-side trainer: its loss values stay device scalars (no read per iteration); the clock is
            # read behind a device synchronisation after iteration 2 and after the last one only
            LOSSES.append(torch.stack([torch.as_tensor(v, dtype=torch.float32, device=vals[0].device).detach() for v in vals]))
            n_total = self.config["total_iterations"] - getattr(self, "init_iter", 0)
            if len(LOSSES) in (2, n_total):
                torch.cuda.synchronize()
                SYNCED.append((len(LOSSES), time.time()))
        else:
            LOSSES.append([float(v) for v in vals])
        STAMPS.append(time.time())
        if tprof_path:  # per-operator host / device time of iterations 3 .. 6 only (set-up and warm-up excluded)
            if len(LOSSES) == 2:
                from torch.profiler import ProfilerActivity, profile
                acts = [ProfilerActivity.CPU] + ([ProfilerActivity.CUDA] if torch.cuda.is_available() else [])
                PROF["p"] = profile(activities=acts, with_stack=bool(os.environ.get("DTK_TRAIN_TORCHPROF_STACK")))
                PROF["p"].start()
            elif len(LOSSES) == 6 and "p" in PROF:
                PROF["p"].stop()
                with open(tprof_path, "w") as fh:
                    fh.write("iterations 3..6 (4 iterations)\n")
                    fh.write(PROF["p"].key_averages().table(sort_by="self_cpu_time_total", row_limit=70, max_name_column_width=70))
                    if torch.cuda.is_available():
                        fh.write("\n\nsorted by device time\n")
                        fh.write(PROF["p"].key_averages().table(sort_by="self_cuda_time_total", row_limit=50, max_name_column_width=70))
                    if os.environ.get("DTK_TRAIN_TORCHPROF_STACK"):
                        fh.write("\n\nby source location, sorted by device time (group_by_stack_n = 6)\n")
                        fh.write(PROF["p"].key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=50,
                                                                                  max_name_column_width=50, max_src_column_width=110))
                    if os.environ.get("DTK_TRAIN_TORCHPROF_STACK"):
                        fh.write("\n\nby source location (group_by_stack_n = 6)\n")
                        fh.write(PROF["p"].key_averages(group_by_stack_n=6).table(sort_by="self_cpu_time_total", row_limit=60,
                                                                                  max_name_column_width=50, max_src_column_width=110))
                del PROF["p"]
        return _update(self, *vals)

    dino_tracker.DINOTracker.update_losses = update_losses
    prof_path = os.environ.get("DTK_TRAIN_CPROFILE")
    if False:
        pass
    elif prof_path:  # where does the HOST spend an iteration
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        try:
            runpy.run_path(script, run_name="__main__")
        finally:
            pr.disable()
            with open(prof_path, "w") as fh:
                pstats.Stats(pr, stream=fh).sort_stats("cumulative").print_stats(70)
                pstats.Stats(pr, stream=fh).sort_stats("tottime").print_stats(40)
    else:
        runpy.run_path(script, run_name="__main__")
