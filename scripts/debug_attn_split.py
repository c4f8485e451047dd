import sys, torch
sys.path.insert(0, "/root/repo")
from dino_tracker_amd import ops
from dino_tracker_amd._lib import OPERAND_F16, check, lib
tdt = torch.float16
F, Hh, S, Sp = 2, 3, 300, 320
g = torch.Generator().manual_seed(5)
q = torch.zeros(F, Hh, Sp, 64); k = torch.zeros(F, Hh, Sp, 64); v = torch.zeros(F, Hh, Sp, 64)
q[:, :, :S] = torch.randn(F, Hh, S, 64, generator=g) * 0.6
k[:, :, :S] = torch.randn(F, Hh, S, 64, generator=g)
v[:, :, :S] = torch.randn(F, Hh, S, 64, generator=g)
if len(sys.argv) > 1 and sys.argv[1] == "crafted":
    q[0, 0, 5] *= 12.0; k[0, 1, 200, 3] += 30.0; q[0, 1, 17, 3] = 4.0; q[1, 2, 299] *= 20.0
def split(x):
    hi = x.to(tdt); lo = (x - hi.float()).to(tdt); return hi, lo
for zero in ("none", "q", "k", "v", "qkv"):
    planes = []
    for name, x, tr in (("q", q, False), ("k", k, False), ("v", v, True)):
        xx = x.transpose(2, 3).contiguous() if tr else x
        hi, lo = split(xx)
        if name in zero: lo = torch.zeros_like(lo)
        planes += [hi.cuda().contiguous(), lo.cuda().contiguous()]
    oh = torch.empty(F, S, Hh * 64, dtype=tdt, device="cuda"); ol = torch.empty_like(oh)
    check(lib().dtk_vit_attention_split(*[ops._p(t) for t in planes], ops._p(oh), ops._p(ol), F, Hh, S, Sp, OPERAND_F16, ops._stream()))
    qd, kd, vd = [(planes[2 * i].double() + planes[2 * i + 1].double()).cpu() for i in range(3)]
    vd = vd.transpose(2, 3)
    s = qd[:, :, :S] @ kd[:, :, :S].transpose(2, 3)
    p = torch.softmax(s * 0.6931471805599453, dim=-1)
    ref = (p @ vd[:, :, :S]).permute(0, 2, 1, 3).reshape(F, S, Hh * 64)
    for label, got in (("hi+lo", (oh.double() + ol.double()).cpu()), ("hi only", oh.double().cpu())):
        err = (got - ref).abs()
        print(f"lo zeroed: {zero:5s} out {label:8s}: max abs err {err.max().item():.3e} median abs {err.median().item():.3e} rms {err.pow(2).mean().sqrt().item():.3e}; "
              f"ref rms {ref.pow(2).mean().sqrt().item():.3e}")
    e = (oh.double() + ol.double()).cpu() - ref
    # per head / per d structure of the error
    eh = e.reshape(F, S, Hh, 64).abs().amax(dim=(0, 1))
    print("   max abs err per head x d-block(8):", [[f"{x:.1e}" for x in eh[h].reshape(8, 8).amax(1).tolist()] for h in range(Hh)][0])
    print("   max abs err by query % 32 (first 8):", [f"{x:.1e}" for x in e.abs().amax(dim=(0, 2))[:S // 32 * 32].reshape(-1, 32).amax(0)[:8].tolist()])
    if zero == "none":
        got = (oh.double() + ol.double()).cpu()
        top = torch.topk((got - ref).abs().flatten(), 8)
        for idx, val in zip(top.indices.tolist(), top.values.tolist()):
            f, rem = divmod(idx, S * Hh * 64); qq, rem = divmod(rem, Hh * 64); hh, d = divmod(rem, 64)
            pr = p[f, hh, qq]
            print(f"   err {val:.3e} at f {f} q {qq} head {hh} d {d}: ref {ref[f, qq, hh * 64 + d].item():.6f} hi {oh[f, qq, hh * 64 + d].item():.6f} lo {ol[f, qq, hh * 64 + d].item():.3e}"
                  f" | row: max p {pr.max().item():.3f} argmax key {pr.argmax().item()} max|s| {s[f, hh, qq].abs().max().item():.1f}; neighbours d-1..d+1 err "
                  f"{[(got - ref)[f, qq, hh * 64 + dd].item() for dd in range(max(0, d - 1), min(64, d + 2))]}")
        oh2 = torch.empty_like(oh); ol2 = torch.empty_like(ol)
        check(lib().dtk_vit_attention_split(*[ops._p(t) for t in planes], ops._p(oh2), ops._p(ol2), F, Hh, S, Sp, OPERAND_F16, ops._stream()))
        print("   deterministic:", torch.equal(oh, oh2), torch.equal(ol, ol2))
