"""Development aid: GraphedIteration step by step with a synchronisation and a print after every stage."""
import os, sys, tempfile, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import train_data as TD
from dino_tracker_amd import train as TR, trainer as T
from dino_tracker_amd.train_ops import install_fused_adam

def say(*a):
    torch.cuda.synchronize(); print(*a, flush=True)

d = tempfile.mkdtemp()
c = dict(TD.CFG, C=384)
d, yml = TD.build(d, None, c, overrides=None, synthetic_video=True)
TR.fix_random_seeds(2)
tr = TR.standalone_trainer(argparse.Namespace(config=yml, data_path=d, device="cuda:0"))
tr.load_fg_masks(); tr.load_dino_best_buddies()
sampler = tr.get_sampler()
model, opt, sched = tr.train_setup()
install_fused_adam(opt); tr.set_model_train(model); tr.init_losses(); tr.prepare_tables(model)
fixed = sampler.draw_frame_sets()
sampler.draw_frame_sets = lambda generator=None: (fixed[0].clone(), list(fixed[1]))
step = T.GraphedIteration(tr, model, opt, sampler, enabled=True)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
v = step.run(1); say("eager ok", v.tolist())
host, union = sampler.draw_frame_sets()
e = step.capture(step.key(union, 2), host, union, 2); say("captured")
from dino_tracker_amd.dataset import stage_to_device
e["staged"].copy_(stage_to_device(host, step.device)); e["frames_set_t"].copy_(stage_to_device(torch.tensor(union, dtype=torch.int32), step.device))
step.adam.refresh(e["params"]); say("refreshed")
e["gA"].replay(); say("A replayed", float(e["st"]["tracking"]))
if e["found"] is not None:
    e["found"].copy_(tr.refined_bb_search(model, e["st"]["prepared"])); say("search ok")
e["gB"].replay(); say("B replayed", e["values"].tolist())
mode = os.environ.get("DBG", "A2")
if mode == "A2":       # graph A alone, again and again
    for k in range(3):
        e["gA"].replay(); say("A again", k, float(e["st"]["tracking"]))
elif mode == "AB":     # both, without the search
    for k in range(3):
        step.adam.refresh(e["params"]); e["gA"].replay(); say("A", k); e["gB"].replay(); say("B", k, e["values"].tolist())
elif mode == "ASB":
    for k in range(3):
        step.adam.refresh(e["params"]); e["gA"].replay(); say("A", k)
        e["found"].copy_(tr.refined_bb_search(model, e["st"]["prepared"])); say("S", k)
        e["gB"].replay(); say("B", k, e["values"].tolist())
elif mode == "run":
    def stat():
        ps = [p for g in opt.param_groups for p in g["params"]]
        return ("params", [round(float(p.abs().max()), 4) for p in ps][:6] + [round(float(p.abs().max()), 4) for p in ps][-4:],
                "grads", [float(p.grad.abs().max()) for p in ps][:6] + [float(p.grad.abs().max()) for p in ps][-4:])
    say(*stat())
    for i in range(3, 6):
        v = step.run(i); sched.step(); say("replay", i, v.tolist(), step.counts); say(*stat())
