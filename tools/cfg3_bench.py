"""Developer timing of BASELINE cfg #3 (bs 32, 16 tracks, AudioFeatureLoss) - one workload of bench.secondary_lines alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench
dev = torch.device("cuda:0")
step = bench.make_workload(dev, seed=2000, bs=32, n_tracks=16, n=bench.N, loss_kind="af", lean=True)
med, mean = bench.time_steps(step, 6, 2)
print(f"cfg3: {med:.3f} ms/step median, {32 / (med * 1e-3):.0f} mixes/s")
