"""Development aid: which part of trainer.iteration_back (+ backward + Adam) does not survive replays?  STAGEB=1..5."""
import os, sys, tempfile, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import train_data as TD
from dino_tracker_amd import train as TR, trainer as T, train_ops
from dino_tracker_amd.train_ops import install_fused_adam
from dino_tracker_amd.dataset import stage_to_device

def say(*a):
    torch.cuda.synchronize(); print(*a, flush=True)

STAGE = int(os.environ.get("STAGEB", "5"))
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
train_ops._PACKED.clear()
train_ops.RETAIN_REPLACED_WORKSPACES = True
torch.cuda.synchronize()
gA, gB, gD = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
keep = {}
hooked = {}
def wrap(mod, name):
    orig = getattr(mod, name)
    cnt = [0]
    def f(*a, **k):
        out = orig(*a, **k)
        if torch.is_tensor(out) and out.requires_grad:
            tag = f"{name}#{cnt[0]}"; cnt[0] += 1
            out.register_hook(lambda g, tag=tag: hooked.__setitem__("g:" + tag, g))
            hooked["y:" + tag] = out
        return out
    setattr(mod, name, f)
if os.environ.get("HOOKS", "0") == "1":
    _orig_p2 = train_ops._pow2_scale
    _cnt = [0]
    def _p2(t):
        r = _orig_p2(t)
        hooked[f"s:pow2#{_cnt[0]}"] = r
        hooked[f"s:pow2in#{_cnt[0]}"] = t
        _cnt[0] += 1
        return r
    train_ops._pow2_scale = _p2
    for nm in ("conv2d_gemm", "batchnorm_train", "blurpool", "align_cnn_to_vit", "attach_grad_sink", "sample_bilinear", "track_points"):
        wrap(train_ops, nm)
i = 2
with torch.cuda.graph(gA, pool=step.pool, stream=step.stream):
    inputs, labels, valid = tr._batch(sampler.batch_from_frame_sets(staged, fs, None))
    st = tr.iteration_front(model, inputs, labels, valid, i)
pre = st["prepared"][2]
found = torch.zeros(2 * pre["P"] * pre["n"], dtype=torch.int32, device=dev)
cfg = tr.config
with torch.cuda.graph(gB, pool=step.pool, stream=step.stream):
    ref_sel = tr.refined_bb_finish(st["frames_set_t"], st["prepared"], found)
    keep["sel"] = ref_sel[2]
    loss = st["tracking"]
    if STAGE >= 2:
        cl_bb, cl_ref = tr.contrastive_losses(model, st["bb_sel"], ref_sel)
        keep["cl_bb"], keep["cl_ref"] = cl_bb, cl_ref
        loss = loss + cl_bb + cl_ref
    if STAGE >= 3:
        a, b = T.emb_regularization_terms(model.frame_embeddings, model.raw_embeddings)
        keep["reg"] = a + b
        loss = loss + a + b + st["cyc"]
    if STAGE >= 4:
        loss.backward()
    if STAGE >= 5:
        params, grads = step.adam.launch()
    keep["loss"] = loss.detach()
del inputs, labels, valid, loss
with torch.cuda.graph(gD, pool=step.pool, stream=step.stream):
    junk = torch.full((1 << 28,), -1, dtype=torch.int64, device=dev)
    junk2 = [torch.full((1 << 16,), -1, dtype=torch.int64, device=dev) for _ in range(64)]
for k in range(int(os.environ.get('REPS','4'))):
    if STAGE >= 5:
        step.adam.refresh(params)
    gA.replay(); say("A", k)
    found.copy_(tr.refined_bb_search(model, st["prepared"])); say("S", k)
    gB.replay(); say("B", k, {n: (float(v.float().abs().max()), bool(torch.isfinite(v.float()).all())) for n, v in keep.items()})
    for n_, v_ in hooked.items():
        f_ = v_.float()
        print("   ", n_, tuple(v_.shape), "finite" if bool(torch.isfinite(f_).all()) else f"NONFINITE {int((~torch.isfinite(f_)).sum())}", float(torch.nan_to_num(f_).abs().max()), flush=True)
    if STAGE >= 5:
        say("  scalars", step.adam.scalars[:4].tolist(), step.adam.scalars[32:36].tolist(), "lr", [g["lr"] for g in opt.param_groups],
            "params", [round(float(p.abs().max()), 4) for p in params][:8], "grads", [float(g.abs().max()) for g in grads][:8],
            "steps", [float(opt.state[p]["step"]) for p in params][:3])
    if os.environ.get("JUNK", "1") == "1":
        gD.replay(); say("D", k)
