"""MR-STFT forward + backward at small batches (rows = 2 bs): time per call.  python tools/loss_small_batch.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch, bench
from mst.loss import MultiResolutionSTFTLoss
dev = torch.device("cuda:0")
f = MultiResolutionSTFTLoss(**bench.RESOLUTIONS)
for bs in (1, 2, 4, 8):
    x = (0.1 * torch.randn(bs, 2, 262144)).to(dev).requires_grad_(True)
    y = (0.1 * torch.randn(bs, 2, 262144)).to(dev)
    def step():
        x.grad = None
        f(x, y).backward()
    med, mean = bench.time_steps(step, 100, 20)
    print(f"bs {bs}: {med * 1000:.1f} us per forward + backward")
