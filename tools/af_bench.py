"""Developer timing of AudioFeatureLoss forward + backward: python tools/af_bench.py [bs ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
from mst.loss import AudioFeatureLoss
dev = torch.device("cuda:0")
n = 262144
for bs in [int(a) for a in sys.argv[1:]] or [8, 32]:
    torch.manual_seed(0)
    loss = AudioFeatureLoss([0.1, 0.001, 1.0, 1.0, 0.1], 44100)
    pred = (0.1 * torch.randn(bs, 2, n, device=dev)).requires_grad_()
    target = 0.1 * torch.randn(bs, 2, n, device=dev)
    def step():
        pred.grad = None
        sum(loss(pred, target).values()).backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    print(f"bs={bs}: AudioFeatureLoss fwd+bwd {e0.elapsed_time(e1) / 10:.3f} ms/step")
