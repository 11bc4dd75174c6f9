"""Developer timing of the AudioFeatureLoss fwd+bwd alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
from mst.loss import AudioFeatureLoss
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
torch.manual_seed(0)
f = AudioFeatureLoss([0.1, 0.001, 1.0, 1.0, 0.1], 44100)
x = torch.randn(bs, 2, n, device=dev, requires_grad=True); y = torch.randn(bs, 2, n, device=dev)
def step():
    x.grad = None
    sum(f(x, y).values()).backward()
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): step()
e1.record(); torch.cuda.synchronize()
print(f"afloss bs={bs} n={n} fwd+bwd ms/step: {e0.elapsed_time(e1)/iters*1e3:.1f} us")
