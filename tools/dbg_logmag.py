"""Developer probe: where does the log-magnitude gradient of the MR-STFT lose accuracy (HIP vs float64 vs the fp32 oracle)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone"), os.path.join(ROOT, "tests")]
import torch
from mst.loss import MultiResolutionSTFTLoss
from oracle import loss_restated as ol
from util import rel
dev = torch.device("cuda:0")
RES = ((512, 256, 512), (2048, 1024, 2048), (8192, 4096, 8192))
for bs, n, seeds in ((1, 50000, (57, 1, 2)), (2, 131072, (131086, 3)), (1, 49152, (57, 5))):
    for seed in seeds:
        torch.manual_seed(seed)
        x = 0.3 * torch.randn(bs, 2, n); y = 0.5 * x + 0.2 * torch.randn(bs, 2, n)
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
            e = (xd.grad.cpu().double() - g[torch.float64]).flatten().abs()
            e2 = (g[torch.float32].double() - g[torch.float64]).flatten().abs()
            top = torch.topk(e, 3)
            print(f"bs {bs} n {n} seed {seed} res {res[0]}: hip-f64 {h64:.2e} ref32-f64 {r:.2e} ratio {h64 / r:.2f}; "
                  f"hip max err {top.values[0]:.2e} at {[int(i) % n for i in top.indices]}; ref32 max err {e2.max():.2e} at {int(e2.argmax()) % n}; |g|max {g[torch.float64].abs().max():.2e}")
