"""Ad-hoc timing of ModelInference.infer on synthetic features (development aid, not the benchmark)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from dino_tracker_amd import ops, synth
from gpu_util import make_inference, make_tracker

T, N, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
method = int(sys.argv[4]) if len(sys.argv) > 4 else 0
H, W = 476, 854
feats = synth.synth_features(T, C, 67, 121, seed=0)
head = synth.synth_head_weights(3)
nx = int(round(N ** 0.5)); ny = N // nx
queries = synth.grid_queries(nx, ny, H, W, 0).cuda()
trk = make_tracker(torch.zeros(T, 3, H, W), feats, head, method=method)
mi = make_inference(trk, H, W, T)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    traj, occ = mi.infer(queries)
    torch.cuda.synchronize(); dt = time.time() - t0
    P = int(mi.last_counts[0])
    maps = queries.shape[0] * T + P * T
    print(f"method {method} T={T} N={queries.shape[0]} C={C} pairs={P} maps={maps} time={dt:.3f}s  maps/s={maps/dt:.0f} qpf/s={queries.shape[0]*T/dt:.1f}", flush=True)
