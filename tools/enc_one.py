"""One precision of the encoder fwd+bwd for profiling: PREC=fp32|bf16x3|bf16 (34 signals x 262144, 3 steps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch, bench
from mst.modules import SpectrogramEncoder
dev = torch.device("cuda:0")
ns, N = int(os.environ.get("NS", "34")), 262144
precision = os.environ.get("PREC", "bf16x3")
torch.manual_seed(3000)
enc = SpectrogramEncoder(embed_dim=512, precision=precision).to(dev).train()
x = (0.1 * torch.randn(ns, 1, N)).to(dev)
g = torch.randn(ns, 512, device=dev)
def enc_step():
    enc.zero_grad(set_to_none=True)
    enc(x).backward(g)
med, mean = bench.time_steps(enc_step, 3, 1)
print(f"{precision:7s} {med:8.2f} ms per step", flush=True)
