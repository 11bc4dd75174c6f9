import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone"), os.path.join(ROOT, "tests")]
import torch
from mst.loss import MultiResolutionSTFTLoss
from oracle import loss_restated as ol
from util import rel
dev = torch.device("cuda:0")
RES = ((512, 256, 512), (2048, 1024, 2048), (8192, 4096, 8192))
corr = sys.argv[1] == "corr"
for seed in range(6):
    torch.manual_seed(1000 * seed + 14 + 131072)
    bs, n = 2, 131072
    x = 0.3 * torch.randn(bs, 2, n); y = (0.5 * x + 0.2 * torch.randn(bs, 2, n)) if corr else 0.25 * torch.randn(bs, 2, n)
    out = []
    for res in RES:
        kw = dict(w_sc=0.0, w_log_mag=1.0)
        xd = x.to(dev).requires_grad_(True)
        f = MultiResolutionSTFTLoss(fft_sizes=[res[0]], hop_sizes=[res[1]], win_lengths=[res[2]], **kw)
        f(xd, y.to(dev)).backward()
        g = {}
        for dt in (torch.float32, torch.float64):
            xo = x.clone().to(dt).requires_grad_(True)
            ol.mrstft_loss(xo, y.to(dt), (res,), **kw).backward()
            g[dt] = xo.grad
        h64, r = rel(xd.grad, g[torch.float64]), rel(g[torch.float32], g[torch.float64])
        out.append(f"{res[0]}: hip {h64:.1e} ref {r:.1e} ratio {h64/r:.2f}")
    print(("corr " if corr else "indep"), seed, " | ".join(out))
