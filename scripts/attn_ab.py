"""Same-box A / B of the stand-alone attention stage between two builds of libdtk.so (ctypes handles of both in one process):
    python scripts/attn_ab.py <lib A> <lib B> [frames]
Benchmark shapes (frames x 6 heads, S = 8108, random fp16 operands), alternating blocks of launches, hipEvents around each block."""
import ctypes
import sys

import torch

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from dino_tracker_amd._lib import SIGNATURES  # noqa: E402


def load(path):
    h = ctypes.CDLL(path)
    res, args = SIGNATURES["dtk_vit_attention"]
    h.dtk_vit_attention.restype, h.dtk_vit_attention.argtypes = res, args
    return h


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 90
    heads, S = 6, 67 * 121 + 1
    Sp = (S + 127) // 128 * 128
    g = torch.Generator(device="cuda").manual_seed(5)
    q = (torch.randn(T, heads, Sp, 64, device="cuda", generator=g) * (0.125 * 1.4426950408889634)).half()
    k = torch.randn(T, heads, Sp, 64, device="cuda", generator=g).half()
    v = torch.randn(T, heads, 64, Sp, device="cuda", generator=g).half()
    k[:, :, S:] = 0
    v[:, :, :, S:] = 0
    outs = [torch.empty(T, S, heads * 64, dtype=torch.float16, device="cuda") for _ in range(2)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(h, o):
        rc = h.dtk_vit_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), T, heads, S, Sp, 0, st)
        assert rc == 0, rc
    for h, o in ((a, outs[0]), (b, outs[1])):
        for _ in range(3):
            run(h, o)
    torch.cuda.synchronize()
    print("outputs bit-identical:", bool(torch.equal(outs[0], outs[1])))
    res = {0: [], 1: []}
    for rnd in range(6):
        for i, (h, o) in enumerate(((a, outs[0]), (b, outs[1]))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(h, o)
            e1.record()
            torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / 10)
    fl = 4.0 * S * S * 384 * T
    for i, name in ((0, sys.argv[1]), (1, sys.argv[2])):
        ms = sorted(res[i])
        med = ms[len(ms) // 2]
        print(f"{name}: median {med:.4f} ms per launch ({fl / med / 1e9:.1f} TFLOP/s), blocks {[round(x, 3) for x in res[i]]}")


if __name__ == "__main__":
    main()
