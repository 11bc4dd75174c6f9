"""Developer timing of the bench step (console fwd + MR-STFT + bwd) at another batch size: python tools/step_bs.py bs [T] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench
bs = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 8; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
step = bench.make_workload(dev, bs, T, bench.N, "mrstft", seed=1)
med, mean = bench.time_steps(step, steps, 5)
print(f"bs {bs} T {T}: {med:.3f} ms/step median ({mean:.3f} mean) -> {bs / med * 1e3:.0f} mixes/s")
