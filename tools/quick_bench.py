"""Developer timing of the console fwd+bwd (not the contract bench; see bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
from mst.modules import AdvancedMixConsole

bs, T, n = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 8, 262144)))
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
c = AdvancedMixConsole(44100, materialize_mixed_tracks=False, validate="deferred")
tracks = (0.1 * torch.randn(bs, T, n)).to(dev)
tp = torch.rand(bs, T, 27, device=dev, requires_grad=True)
fp = torch.rand(bs, 25, device=dev)
mp = torch.rand(bs, 26, device=dev, requires_grad=True)
g = torch.randn(bs, 2, n, device=dev)
flags = dict(use_fx_bus=False)
def step():
    tp.grad = None; mp.grad = None
    _, mix, *_ = c(tracks, tp, fp, mp, **flags)
    mix.backward(g)
for _ in range(5): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.time(); e0.record()
for _ in range(iters): step()
e1.record(); torch.cuda.synchronize(); t1 = time.time()
ms = e0.elapsed_time(e1) / iters
print(f"console fwd+bwd bs={bs} T={T} n={n}: {ms:.3f} ms/step (wall {1e3*(t1-t0)/iters:.3f})  -> {bs/ms*1e3:.1f} mixes/s; lean bytes {8*n*(T+2)*bs/1e6:.1f} MB -> {8*n*(T+2)*bs/ms/1e9:.3f} TB/s algorithmic")
with torch.no_grad():
    for _ in range(3): c(tracks, tp, fp, mp, **flags)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): c(tracks, tp, fp, mp, **flags)
    e1.record(); torch.cuda.synchronize()
print(f"console fwd only (no_grad): {e0.elapsed_time(e1)/iters:.3f} ms")
c.check_parameters()
