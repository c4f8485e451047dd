"""Does a hipMemsetAsync captured into a graph clear its range on every replay?  (round 6: the suspect behind two replay failures)"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for nbytes in (8, 256, 4096, 1 << 20, (1 << 20) + 512):
    buf = torch.full((nbytes,), 7, dtype=torch.uint8, device="cuda")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        out = buf.sum(dtype=torch.int64)
    res = []
    for k in range(3):
        buf.fill_(7 + k)
        g.replay()
        torch.cuda.synchronize()
        res.append((int(out), int(buf.sum(dtype=torch.int64))))
    print(nbytes, "rc", rc, "after replays (sum seen inside graph, sum after):", res, flush=True)
