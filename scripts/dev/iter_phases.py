"""Development aid: where do the slow iterations of the replayed training loop spend their time?  (sync around every phase)"""
import argparse, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import train_data as TD
from dino_tracker_amd import train as TR, trainer as T
from dino_tracker_amd.train_ops import install_fused_adam
from dino_tracker_amd.dataset import stage_to_device
cfg = dict(TD.CFG, T=90, C=384, H=476, W=854, total_iterations=10 ** 6, n_fg=4000, n_bg=6000, bb_per_pair=24)
d = tempfile.mkdtemp()
d, yml = TD.build(d, None, cfg, overrides={}, synthetic_video=True)
TR.fix_random_seeds(2)
tr = TR.standalone_trainer(argparse.Namespace(config=yml, data_path=d, device="cuda:0"))
tr.load_fg_masks(); tr.load_dino_best_buddies()
sampler = tr.get_sampler()
model, opt, sched = tr.train_setup()
install_fused_adam(opt); tr.set_model_train(model); tr.init_losses(); tr.prepare_tables(model)
step = T.GraphedIteration(tr, model, opt, sampler, enabled=True)
for i in range(1, 16):
    step.run(i); sched.step()
torch.cuda.synchronize()
rows = []
for i in range(16, 76):
    t = [time.perf_counter()]
    def mark():
        torch.cuda.synchronize(); t.append(time.perf_counter())
    host, union = sampler.draw_frame_sets()
    key = step.key(union, i)
    e = step.entries.get(key)
    if e is None:
        step.run(i); sched.step(); continue
    mark()
    e["staged"].copy_(stage_to_device(host, step.device)); e["frames_set_t"].copy_(stage_to_device(torch.tensor(union, dtype=torch.int32), step.device))
    step.adam.refresh(e["params"]); mark()
    e["gA"].replay(); mark()
    f = tr.refined_bb_search(model, e["st"]["prepared"]); mark()
    e["found"].copy_(f); e["gB"].replay(); mark()
    sched.step()
    rows.append((len(union),) + tuple(round((b - a) * 1e3, 2) for a, b in zip(t[:-1], t[1:])))
if os.environ.get("ROWS", "0") == "1":
    print("frames | draw | stage+adam | graph A | search | graph B   (ms)")
    for r in rows:
        print(r, "  total", round(sum(r[1:]), 2))
import statistics
r8 = [r for r in rows if r[0] == 8]
print("medians over", len(r8), "eight-frame iterations (ms): draw %.2f stage+adam %.2f graph A %.2f search %.2f graph B %.2f; A + search + B = %.2f" % (
    *[statistics.median(r[k] for r in r8) for k in range(1, 6)], sum(statistics.median(r[k] for r in r8) for k in (3, 4, 5))))
