"""Developer timing of the spectrogram encoder (STFT front end + Cnn14 fwd+bwd): python tools/encoder_bench.py [signals=16] [samples=262144] [precision=bf16] [iters=5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
from mst.modules import SpectrogramEncoder

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = SpectrogramEncoder(embed_dim=512, precision=prec).to(dev).train()
x = 0.1 * torch.randn(ns, 1, n, device=dev)
g = torch.randn(ns, 512, device=dev)


def conv_flops(frames, bins):
    """forward multiply-adds x 2 of the twelve convolutions for one signal (reference mst/panns.py:135-198)"""
    ch = (1, 64, 128, 256, 512, 1024, 2048)
    pools = ((2, 2), (4, 4), (2, 4), (2, 4), (2, 4), (2, 2))  # (frames, bins)
    h, w, f = frames, bins, 0
    for b in range(6):
        f += 2 * 9 * h * w * (ch[b] * ch[b + 1] + ch[b + 1] * ch[b + 1])
        h, w = h // pools[b][0], w // pools[b][1]
    return f


fwd_flops = ns * conv_flops(1 + n // 512, 1025)


def step():
    enc.zero_grad(set_to_none=True)
    e = enc(x)
    e.backward(g)


def fwd():
    with torch.no_grad():
        enc(x)


for fn, name, fl in ((fwd, "forward (no_grad)", fwd_flops), (step, "forward + backward", 3 * fwd_flops)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"encoder {prec} {ns} x {n}: {name}: {ms:.2f} ms -> {ns / ms * 1e3:.1f} signals/s, {fl / ms / 1e9:.1f} TFLOP/s of convolution "
          f"({100 * fl / ms / 1e9 / (2500.0 if prec == 'bf16' else 157.3):.1f} % of the dense {'bf16' if prec == 'bf16' else 'fp32'} MFMA peak)")
print(f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
