"""Micro-benchmark of the training convolutions at the Delta-DINO layer shapes (8 frames of 854 x 476): forward, data gradient and
weight gradient of _ConvMfma, per layer, timed with device events.  $DTK_WGRAD_WORKGROUPS varies the pixel split of the weight
gradient."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dino_tracker_amd import ops, train_ops  # noqa: E402

LAYERS = [dict(cin=64, cout=128, h=238, w=427, dil=1), dict(cin=128, cout=256, h=119, w=214, dil=1),
          dict(cin=256, cout=384, h=60, w=107, dil=2)]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = 8
    out = {}
    for L in LAYERS:
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.rand(n, L["cin"], L["h"], L["w"], device="cuda", generator=g)
        w = torch.randn(L["cout"], L["cin"], 5, 5, device="cuda", generator=g) * 0.02
        dy = torch.randn(n, L["cout"], L["h"], L["w"], device="cuda", generator=g) * 1e-4
        s = train_ops._pow2_scale(dy)
        flop = 2.0 * n * L["h"] * L["w"] * 25 * L["cin"] * L["cout"]
        dw = torch.zeros_like(w)
        t_f = timed(lambda: train_ops._implicit_conv(x, w, L["dil"], False, False, 0, False, None))
        t_d = timed(lambda: train_ops._implicit_conv(dy, w, L["dil"], True, True, 2 * L["dil"], True, s))
        t_w = timed(lambda: ops.conv_wgrad_split(x, dy, dw, L["dil"], True, s, train_ops.CONV_OPERANDS == "fp16"))
        key = f"{L['cin']}->{L['cout']} {L['h']}x{L['w']} d{L['dil']}"
        out[key] = {"gflop": flop / 1e9, "forward_ms": t_f, "dgrad_ms": t_d, "wgrad_ms": t_w,
                    "forward_tflops": flop / t_f / 1e9, "dgrad_tflops": flop / t_d / 1e9, "wgrad_tflops": flop / t_w / 1e9}
    print(json.dumps({"operands": train_ops.CONV_OPERANDS, "wgrad_workgroups": os.environ.get("DTK_WGRAD_WORKGROUPS", "default"),
                      "layers": out}))


if __name__ == "__main__":
    main()
