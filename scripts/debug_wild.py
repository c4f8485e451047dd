import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests"); sys.path.insert(0, R + "/tests/golden")
import numpy as np, torch
import make_golden as MG
from dino_tracker_amd import ops
from gpu_util import make_inference, make_tracker
name = sys.argv[1] if len(sys.argv) > 1 else "p3_small_wild"
cfg = MG.CASES[name]
gold = np.load(f"{R}/tests/golden/{name}.npz")
video, dino, head, queries, delta = MG.build_inputs(cfg)
res = {}
for method in (0, 1):
    trk = make_tracker(video, dino, head, method=method)
    mi = make_inference(trk, cfg["H"], cfg["W"], cfg["T"])
    t3 = mi.compute_trajectories(queries.cuda())
    cs = mi.compute_trajectory_cos_sims(t3, queries.cuda())
    buf, green = mi._anchor_stage(t3, cs)
    P = int(buf.counts[0])
    occ = mi.compute_occlusion(t3, cs, mi.compute_anchor_trajectories(t3, cs))
    res[method] = (t3.cpu(), cs.cpu(), green[:P].cpu(), occ.cpu())
d_t = (res[0][0] - res[1][0]).abs().max().item()
d_g = (res[0][2] - res[1][2]).abs()
print("traj diff", d_t, "green max diff", d_g.max().item(), "n green > 1e-3:", int((d_g > 1e-3).sum()), "of", d_g.numel())
idx = torch.nonzero(d_g.max(dim=-1).values > 1e-3)
for i in idx[:10]:
    p, t = int(i[0]), int(i[1])
    print("pair", p, "t", t, "exact", res[0][2][p, t].tolist(), "mfma", res[1][2][p, t].tolist())
print("occ equal exact-vs-mfma", torch.equal(res[0][3], res[1][3]), "exact-vs-gold", np.array_equal(res[0][3].numpy(), gold["occ"]), "mfma-vs-gold", np.array_equal(res[1][3].numpy(), gold["occ"]))
from oracle import ref_algo as A
rt, ro, rcs, greens = A.infer(dino, queries, head, cfg["H"], cfg["W"], return_aux=True)
print("cs diff exact-vs-mfma", (res[0][1]-res[1][1]).abs().max().item(), "vs oracle", (res[1][1]-rcs).abs().max().item())
print("anchor sets equal", torch.equal(res[0][1] >= 0.7, res[1][1] >= 0.7), torch.equal(res[1][1] >= 0.7, rcs >= 0.7))
off = 0
for n in range(queries.shape[0]):
    A_n = int((rcs[n] >= 0.7).sum())
    md, mc = A.occlusion_margins(greens[n], rt[n], rcs[n], 0.7, 0.6)
    g1 = res[1][2][off:off + A_n]
    dg = (g1 - greens[n]).abs().max().item()
    bad = torch.nonzero(res[1][3][n] != ro[n])[:, 0].tolist()
    if bad or dg > 1e-3:
        print("query", n, "A", A_n, "green diff vs oracle", dg, "bad t", bad, "md", md[bad].tolist(), "mc", mc[bad].tolist(), "vis", (rcs[n] >= 0.7)[bad].tolist())
        md1, _ = A.occlusion_margins(g1, res[1][0][n, :, :2], res[1][1][n], 0.7, 0.6)
        print("   mfma-side md", md1[bad].tolist())
        dd = (g1 - greens[n]).abs().max(dim=-1).values
        ij = torch.nonzero(dd > 1e-3)
        for a_, t_ in ij[:6].tolist():
            print("   green[", a_, t_, "] oracle", greens[n][a_, t_].tolist(), "mfma", g1[a_, t_].tolist(), "exact", res[0][2][off + a_, t_].tolist())
    off += A_n
print("---- isolate occlusion kernel on mfma inputs")
trk = make_tracker(video, dino, head, method=1)
mi = make_inference(trk, cfg["H"], cfg["W"], cfg["T"])
traj, occ = mi.infer(queries.cuda())
print("infer occ vs oracle mismatches:", torch.nonzero(occ.cpu() != ro).tolist())
t3 = mi.compute_trajectories(queries.cuda()); cs = mi.compute_trajectory_cos_sims(t3, queries.cuda())
buf, green = mi._anchor_stage(t3, cs)
torch.cuda.synchronize()
P = int(buf.counts[0]); off = buf.pair_off.cpu().tolist(); pf = buf.pair_frame[:P].cpu().tolist()
occ_k = ops.occlusion(green, buf.pair_off, buf.pair_frame, t3[..., :2].contiguous(), cs, 0.7, 0.6).cpu()
g = green[:P].cpu(); tr = t3[..., :2].cpu(); c = cs.cpu()
for n in range(queries.shape[0]):
    o = A.occlusion_for_query(g[off[n]:off[n+1]], tr[n], c[n], 0.7, 0.6)
    if not torch.equal(o, occ_k[n]):
        print("kernel != formula at query", n, o.tolist(), occ_k[n].tolist(), "pair frames", pf[off[n]:off[n+1]], "anchors by cs", torch.nonzero(c[n] >= 0.7)[:,0].tolist())
print("redo count last chunk etc: counts", buf.counts.cpu().tolist())
n = 5
torch.save(dict(g=g[off[n]:off[n+1]], tr=tr[n], c=c[n], pf=pf[off[n]:off[n+1]], occ_k=occ_k[n]), R + "/gpurun_out/dbg.pt")
