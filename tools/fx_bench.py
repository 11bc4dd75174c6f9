"""Developer timing of the console with the fx bus on (reference default flags) vs off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
from mst.modules import AdvancedMixConsole
bs, T, n = 8, 8, 262144
dev = torch.device("cuda:0")
torch.manual_seed(0)
c = AdvancedMixConsole(44100, materialize_mixed_tracks=False, validate="deferred", param_dicts="lazy")
c.fx_noise = torch.randn(bs * 2, 12, 65536 + 1022, device=dev)
tracks = (0.1 * torch.randn(bs, T, n)).to(dev)
tp = torch.rand(bs, T, 27, device=dev, requires_grad=True); fp = torch.rand(bs, 25, device=dev, requires_grad=True)
mp = torch.rand(bs, 26, device=dev, requires_grad=True); g = torch.randn(bs, 2, n, device=dev)
for fx in (False, True):
    def step():
        tp.grad = None; mp.grad = None; fp.grad = None
        _, mix, *_ = c(tracks, tp, fp, mp, use_fx_bus=fx)
        mix.backward(g)
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): step()
    e1.record(); torch.cuda.synchronize()
    print(f"use_fx_bus={fx}: console fwd+bwd {e0.elapsed_time(e1) / 10:.3f} ms/step")
