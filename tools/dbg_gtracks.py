import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone"), os.path.join(ROOT, "tests")]
from mst.modules import AdvancedMixConsole
from oracle import console_restated as oc
from util import FULL, rel
dev = torch.device("cuda:0")
torch.manual_seed(21)
bs, T, n = 1, 4, 262144
tracks = 0.1 * torch.randn(bs, T, n)
tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
gmix = torch.randn(bs, 2, n)
c = AdvancedMixConsole(44100)
res = {}
for name, env in (("inwave", None), ("multipass", "1")):
    if env: os.environ["MST_MULTIPASS_EQ"] = env
    tr = tracks.to(dev).requires_grad_(True)
    a, b = tp.to(dev).requires_grad_(True), mp.to(dev).requires_grad_(True)
    _, mix, *_ = c(tr, a, fp.to(dev), b, **FULL)
    (mix * gmix.to(dev)).sum().backward()
    res[name] = (mix.detach().cpu(), tr.grad.cpu(), a.grad.cpu(), b.grad.cpu())
outs = {}
for dt in (torch.float32, torch.float64):
    tr = tracks.clone().to(dt).detach().clone().requires_grad_(True)
    a, b = tp.clone().to(dt).detach().clone().requires_grad_(True), mp.clone().to(dt).detach().clone().requires_grad_(True)
    _, mix, *_ = oc.console_forward(tr, a, fp.to(dt), b, **FULL)
    (mix * gmix.to(dt)).sum().backward()
    outs[dt] = (mix.detach(), tr.grad, a.grad, b.grad)
for name in res:
    print(name, "mix vs f64 %.2e | g_tracks vs f64 %.2e vs ref32 %.2e | g_tp vs f64 %.2e | g_mp vs f64 %.2e" % (
        rel(res[name][0], outs[torch.float64][0]), rel(res[name][1], outs[torch.float64][1]), rel(res[name][1], outs[torch.float32][1]),
        rel(res[name][2], outs[torch.float64][2]), rel(res[name][3], outs[torch.float64][3])))
print("ref32 vs f64: mix %.2e g_tracks %.2e g_tp %.2e g_mp %.2e" % tuple(rel(outs[torch.float32][i], outs[torch.float64][i]) for i in range(4)))
print("inwave vs multipass g_tracks %.2e" % rel(res["inwave"][1], res["multipass"][1]))
g64 = outs[torch.float64][1]
for name in res:
    d = (res[name][1].double() - g64)
    # per-tile relative error along time of row 0
    e = d[0, 0].view(64, 4096).norm(dim=1) / g64[0, 0].view(64, 4096).norm(dim=1)
    print(name, "per-tile rel err row0:", " ".join(f"{v:.1e}" for v in e[:8]), "...", " ".join(f"{v:.1e}" for v in e[-4:]))
