"""TransformerController forward + backward alone at the cfg #5 shape (1 x 32 tracks, 12 layers): torch eager / hipGraph replay /
csrc/mst_ctrl.hip.  python tools/ctrl_bench.py [bs=1] [tracks=32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench
from mst.modules import TransformerController

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
torch.manual_seed(0)
te, me = torch.randn(bs, T, 512, device=dev), torch.randn(bs, 2, 512, device=dev)
mask = torch.zeros(bs, T, dtype=torch.bool, device=dev)
w = torch.randn(bs, T, 27, device=dev)
for label, kw in (("torch eager", dict(native=False)), ("torch kernels replayed as two hipGraphs", dict(graphed=True, native=False)), ("csrc/mst_ctrl.hip (native=True)", dict(native=True))):
    torch.manual_seed(1)
    ctrl = TransformerController(512, 27, 25, 26, num_layers=12, nhead=8, **kw).to(dev).train()

    def step():
        for p in ctrl.parameters():
            p.grad = None
        a, b = te.clone().requires_grad_(True), me.clone().requires_grad_(True)
        tp, fp, mp = ctrl(a, b, mask)
        ((tp * w).sum() + mp.sum()).backward()

    med, mean = bench.time_steps(step, 30, 5)
    print(f"controller fwd+bwd, bs {bs} x {T} tracks, 12 layers - {label:42s}: {med:.3f} ms median, {mean:.3f} mean")
