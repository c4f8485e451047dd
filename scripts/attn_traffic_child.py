"""Child of bench.py's live `roofline.traffic` leg: a few launches of the stand-alone attention stage (dtk_vit_attention) on random
16-bit operands of the benchmark's shape, to be run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE`.  The kernel's
HBM-side bytes do not depend on the operand values; the shapes are the timed step's (frames x heads x S x 64).
    python scripts/attn_traffic_child.py <frames> <heads> <S> <fp16|bf16> [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dino_tracker_amd import ops  # noqa: E402
from dino_tracker_amd._lib import OPERAND_BF16, OPERAND_F16, check, lib  # noqa: E402

T, heads, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
odt = torch.float16 if sys.argv[4] == "fp16" else torch.bfloat16
n = int(sys.argv[5]) if len(sys.argv) > 5 else 3
Sp = (S + 127) // 128 * 128
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(5)
q = (torch.randn(T, heads, Sp, 64, device=dev, generator=g) * (0.125 * 1.4426950408889634)).to(odt)
k = torch.randn(T, heads, Sp, 64, device=dev, generator=g).to(odt)
v = torch.randn(T, heads, 64, Sp, device=dev, generator=g).to(odt)
k[:, :, S:] = 0
v[:, :, :, S:] = 0
o = torch.empty(T, S, heads * 64, dtype=odt, device=dev)
for _ in range(n):
    check(lib().dtk_vit_attention(ops._p(q), ops._p(k), ops._p(v), ops._p(o), T, heads, S, Sp,
                                  OPERAND_F16 if odt == torch.float16 else OPERAND_BF16, ops._stream()))
torch.cuda.synchronize()
