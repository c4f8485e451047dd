"""What the GEMM library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the ViT's shapes: M = 90 x 8108 tokens, fp16 operands, fp32
accumulation, plain product without the epilogues the hand-written kernels fuse (bias, GELU, LayerScale, the q / k / V^T layouts)."""
import torch
M = 90 * 8108
for name, K, N in (("S qkv", 384, 1152), ("S fc1", 384, 1536), ("S fc2", 1536, 384), ("S proj", 384, 384),
                   ("L qkv", 1024, 3072), ("L fc1", 1024, 4096), ("L fc2", 4096, 1024), ("L proj", 1024, 1024)):
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16)
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.matmul(a, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:7s} K={K:5d} N={N:5d}: {ms:7.3f} ms  {2.0 * M * K * N / ms / 1e9:7.1f} TFLOP/s  {(M * K + M * N) * 2 / ms / 1e6:6.0f} GB/s in+out", flush=True)
    del a, w, out
