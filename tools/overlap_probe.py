"""A/B of AdvancedMixConsole(overlap_backward_prepare=...) on the bench step (no profiler attached)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench
dev = torch.device("cuda:0")
step = bench.make_workload(dev, bench.BS, bench.T, bench.N, "mrstft", seed=1000, lean=True)
for flag in (True, False, True, False):
    step.console.overlap_backward_prepare = flag
    med, mean = bench.time_steps(step, 100, 20)
    print(f"overlap_backward_prepare={flag}: median {med:.4f} ms, mean {mean:.4f} ms")
