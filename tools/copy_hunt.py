"""Developer probe: which part of the bench step issues the device-to-device copy kernels (run under rocprofv3 --kernel-trace).
python tools/copy_hunt.py fwd|loss|full|full_direct"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench
from mst.loss import MultiResolutionSTFTLoss
from mst.modules import AdvancedMixConsole
mode = sys.argv[1]
dev = torch.device("cuda:0")
step = bench.make_workload(dev, 8, 8, bench.N, "mrstft", seed=1)
console = step.console
tracks = (0.1 * torch.randn(8, 8, bench.N)).to(dev)
ref = torch.randn(8, 2, bench.N, device=dev)
tp = torch.rand(8, 8, 27, device=dev).requires_grad_(True)
fp = torch.rand(8, 25, device=dev)
mp = torch.rand(8, 26, device=dev).requires_grad_(True)
loss_fn = MultiResolutionSTFTLoss(**bench.RESOLUTIONS)
seed = torch.ones((), device=dev)
for _ in range(20):
    tp.grad = None; mp.grad = None
    if mode == "fwd":
        with torch.no_grad():
            console(tracks, tp, fp, mp, **bench.FLAGS)
    elif mode == "loss":
        _, mix, *_ = console(tracks, tp, fp, mp, **bench.FLAGS)
        loss_fn(mix, ref)
    elif mode == "full":
        _, mix, *_ = console(tracks, tp, fp, mp, **bench.FLAGS)
        loss = loss_fn(mix, ref)
        torch.autograd.backward(loss, grad_tensors=seed.expand_as(loss))
    elif mode == "full_direct":
        _, mix, *_ = console(tracks, tp, fp, mp, **bench.FLAGS)
        loss = loss_fn(mix, ref)
        torch.autograd.backward(loss, grad_tensors=seed)
    elif mode == "lossbwd":
        mix = torch.randn(8, 2, bench.N, device=dev, requires_grad=True) if _ == 0 else mix
        mix.grad = None
        loss = loss_fn(mix, ref)
        torch.autograd.backward(loss, grad_tensors=seed)
torch.cuda.synchronize()
