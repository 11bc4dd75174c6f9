"""Encoder fwd+bwd at the three precisions: time and distance from the fp32 path (34 signals x 262144)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch, bench
from mst.modules import SpectrogramEncoder
dev = torch.device("cuda:0")
ns, N = int(os.environ.get("NS", "34")), 262144
res = {}
for precision in ("fp32", "bf16x6", "bf16x3", "bf16"):
    torch.manual_seed(3000)
    enc = SpectrogramEncoder(embed_dim=512, precision=precision).to(dev).train()
    x = (0.1 * torch.randn(ns, 1, N)).to(dev)
    g = torch.randn(ns, 512, device=dev)
    def enc_step():
        enc.zero_grad(set_to_none=True)
        e = enc(x); e.backward(g); return e
    med, mean = bench.time_steps(enc_step, 5, 2)
    e = enc_step()
    res[precision] = (e.detach().double(), {k: p.grad.detach().double() for k, p in enc.named_parameters()})
    print(f"{precision:7s} {med:8.2f} ms per step", flush=True)
    del enc, x, g
    torch.cuda.empty_cache()
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
for p in ("bf16x6", "bf16x3", "bf16"):
    ge = {k: rel(res[p][1][k], res["fp32"][1][k]) for k in res["fp32"][1]}
    w = max(ge, key=ge.get)
    print(f"{p} vs fp32: embed {rel(res[p][0], res['fp32'][0]):.2e}; gradients max {ge[w]:.2e} ({w}), median {sorted(ge.values())[len(ge)//2]:.2e}")
