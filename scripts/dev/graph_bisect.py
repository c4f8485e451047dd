"""Development aid: which part of trainer.iteration_front does not survive a second replay?  STAGE=1..6."""
import os, sys, tempfile, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import train_data as TD
from dino_tracker_amd import train as TR, trainer as T
from dino_tracker_amd.train_ops import install_fused_adam
from dino_tracker_amd.dataset import stage_to_device

def say(*a):
    torch.cuda.synchronize(); print(*a, flush=True)

STAGE = int(os.environ.get("STAGE", "6"))
d = tempfile.mkdtemp()
d, yml = TD.build(d, None, dict(TD.CFG, C=384), overrides=None, synthetic_video=True)
TR.fix_random_seeds(2)
tr = TR.standalone_trainer(argparse.Namespace(config=yml, data_path=d, device="cuda:0"))
tr.load_fg_masks(); tr.load_dino_best_buddies()
sampler = tr.get_sampler()
model, opt, sched = tr.train_setup()
install_fused_adam(opt); tr.set_model_train(model); tr.init_losses(); tr.prepare_tables(model)
fixed = sampler.draw_frame_sets()
sampler.draw_frame_sets = lambda generator=None: (fixed[0].clone(), list(fixed[1]))
step = T.GraphedIteration(tr, model, opt, sampler, enabled=True)
v = step.run(1); say("eager ok")
host, union = fixed
dev = step.device
staged = torch.zeros(host.shape, dtype=torch.long, device=dev); staged.copy_(stage_to_device(host, dev))
fs = torch.tensor(union, dtype=torch.int32).to(dev)
opt.zero_grad(set_to_none=True)
model.frame_embeddings = model.raw_embeddings = model.residual_embeddings = None
from dino_tracker_amd import train_ops
train_ops._PACKED.clear()
torch.cuda.synchronize()
gA, gD = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
keep = {}
with torch.cuda.graph(gA, pool=step.pool, stream=step.stream):
    inputs, labels, valid = tr._batch(sampler.batch_from_frame_sets(staged, fs, None))
    keep["out"] = inputs[0].sum()
    if STAGE >= 2:
        coords = model(inputs); keep["out"] = coords.sum()
    if STAGE >= 3:
        keep["tracking"] = T.weighted_mean(T.huber(coords, labels), valid)
    SUB = int(os.environ.get("SUB", "9"))
    if STAGE == 4 and SUB < 9:
        pts, src_idx, tgt_idx, vld = model._cycle_point_sets_static(fs, tr.fg_masks)
        keep["pts"] = pts
        if SUB >= 2:
            src_tgt = model.get_point_predictions((pts, src_idx, tgt_idx, fs), model.frame_embeddings); keep["st"] = src_tgt
        if SUB >= 3:
            unnorm = lambda c: model.range_normalizer.unnormalize(c, src=(-1, 1), dims=[0, 1])
            with torch.no_grad():
                tgt_pts = torch.cat([unnorm(src_tgt.detach()), fs.float()[tgt_idx][:, None]], dim=1)
            keep["tp"] = tgt_pts
        if SUB >= 4:
            keep["ts"] = model.get_point_predictions((tgt_pts, tgt_idx, src_idx, fs), model.frame_embeddings)
    elif STAGE >= 4:
        keep["cyc"] = tr.cycle_terms(model, fs)
    if STAGE >= 5:
        keep["bb"] = tr.dino_bb_selection(fs)
    if STAGE >= 6:
        keep["prep"] = tr.refined_bb_prepare(model, fs)
if os.environ.get("DROP", "1") == "1":
    del inputs, labels, valid
    if STAGE >= 2: del coords
with torch.cuda.graph(gD, pool=step.pool, stream=step.stream):
    junk = torch.full((1 << 28,), -1, dtype=torch.int64, device=dev)     # 2 GB of 0xff over whatever the pool considers free
    junk2 = [torch.full((1 << 16,), -1, dtype=torch.int64, device=dev) for _ in range(64)]
for k in range(3):
    gA.replay(); say("A", k, float(keep["out"]), {n: (float(v.float().abs().max()), bool(torch.isfinite(v.float()).all())) for n, v in keep.items() if torch.is_tensor(v)})
    gD.replay(); say("D", k)
