"""Per-layer backward check of the train-mode Delta-DINO on the device: every layer is given the HOST run's input
activation and output gradient, so the numbers isolate one library kernel each (no amplification along the chain).
Prints max |device - host| / max |host| for the input gradient and the parameter gradients of every layer, then the same
for the whole chain in float32 and with the host chain in float64 as the arbiter."""
import copy
import sys

import torch

sys.path.insert(0, ".")
from dino_tracker_amd import synth  # noqa: E402
from dino_tracker_amd.networks import DeltaDINO  # noqa: E402

C, H, W, n = 64, 126, 210, 4
HAVE_GPU = torch.cuda.is_available()
DEVS = (("cpu", torch.float64), ("cpu", torch.float32), ("cuda" if torch.cuda.is_available() else "cpu", torch.float32))
dd = DeltaDINO(channels=[3, 64, 128, 256, C])
dd.load_state_dict(synth.synth_delta_dino_weights(C, 6))
dd.train()
g = torch.Generator().manual_seed(0)
x = torch.rand(n, 3, H, W, generator=g)
acts = [x]
with torch.no_grad():
    for layer in dd.layers:
        acts.append(layer(acts[-1]))
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))
for i, layer in enumerate(dd.layers):
    cot = torch.randn(acts[i + 1].shape, generator=g)
    res = {}
    for slot, (dev, dt) in enumerate(DEVS):
        l2 = copy.deepcopy(layer).to(dev, dt)
        l2.train()
        xi = acts[i].detach().clone().to(dev, dt).requires_grad_()
        y = l2(xi)
        (y * cot.to(dev, dt)).sum().backward()
        res[slot] = [y.detach(), xi.grad] + [p.grad for p in l2.parameters()]
    if HAVE_GPU and isinstance(layer, torch.nn.BatchNorm2d):  # the hand-written BatchNorm (csrc/train.hip)
        from dino_tracker_amd import train_ops
        l2 = copy.deepcopy(layer).to("cuda")
        l2.train()
        xi = acts[i].detach().clone().cuda().requires_grad_()
        y = train_ops.batchnorm_train(l2, xi, False)
        (y * cot.cuda()).sum().backward()
        res[3] = [y.detach(), xi.grad] + [p.grad for p in l2.parameters()]
    ref = res[0]
    names = ["out", "dx"] + [n_ for n_, _ in layer.named_parameters()]
    print(f"layer {i:2d} {type(layer).__name__:16s}", "  ".join(
        f"{nm}: host {rel(a, r):.1e} lib {rel(b, r):.1e}" for nm, a, b, r in zip(names, res[1], res[2], ref)))
    if 3 in res:
        print("         csrc/train.hip  ", "  ".join(f"{nm}: {rel(a, r):.1e}" for nm, a, r in zip(names, res[3], ref)))

# whole chain
vit = torch.zeros(n, C, (H - 14) // 7 + 1, (W - 14) // 7 + 1)
cot = torch.randn(vit.shape, generator=g)
grads = {}
for slot, (dev, dt) in enumerate(DEVS[:2]):
    m = copy.deepcopy(dd).to(dev, dt)
    m.train()
    (m(x.to(dev, dt), vit.to(dev, dt)) * cot.to(dev, dt)).sum().backward()
    grads[slot] = dict((k, p.grad) for k, p in m.named_parameters())
grads[2] = grads[1]
if HAVE_GPU:  # the product path: DeltaDINO.forward in training mode on the device (library convs + csrc/train.hip BatchNorm)
    m = copy.deepcopy(dd).cuda()
    m.train()
    (m(x.cuda(), vit.cuda()) * cot.cuda()).sum().backward()
    grads[3] = dict((k, p.grad) for k, p in m.named_parameters())
    m = copy.deepcopy(dd).cuda()
    m.train()
    xx = x.cuda()
    for layer in m.layers:  # everything on the library kernels (torch.nn.BatchNorm2d), for comparison
        xx = layer(xx)
    from dino_tracker_amd import train_ops
    (train_ops.align_cnn_to_vit(xx, vit.shape[-2], vit.shape[-1], 7, 14, 8) * cot.cuda()).sum().backward()
    grads[2] = dict((k, p.grad) for k, p in m.named_parameters())
print("whole chain, gradient error relative to the float64 chain (max-normalised):")
for k, r in grads[0].items():
    if r.abs().max() < 1e-12:
        continue
    print(f"  {k:20s} host fp32 {rel(grads[1][k], r):.1e}   device, library BatchNorm {rel(grads[2][k], r):.1e}" +
          (f"   device, csrc/train.hip BatchNorm {rel(grads[3][k], r):.1e}" if 3 in grads else ""))
