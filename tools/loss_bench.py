"""Developer timing of the MR-STFT loss fwd+bwd alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
from mst.loss import MultiResolutionSTFTLoss
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
torch.manual_seed(0)
f = MultiResolutionSTFTLoss(fft_sizes=[512, 2048, 8192], hop_sizes=[256, 1024, 4096], win_lengths=[512, 2048, 8192])
x = torch.randn(bs, 2, n, device=dev, requires_grad=True); y = torch.randn(bs, 2, n, device=dev)
for _ in range(3):
    x.grad = None; f(x, y).backward()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    x.grad = None; f(x, y).backward()
e1.record(); torch.cuda.synchronize()
print(f"mrstft bs={bs} n={n} fwd+bwd ms/step: {e0.elapsed_time(e1)/iters*1e3:.1f} us")
