"""Developer probe: does the bench step get faster once the GPU has been busy for a while (clock ramp)?  python tools/clock_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench
dev = torch.device("cuda:0")
step = bench.make_workload(dev, 8, 8, bench.N, "mrstft", seed=1)
for label, steps, warm in (("cold: 5 warm-up + 20 timed", 20, 5), ("again 20", 20, 0), ("200 timed", 200, 0), ("20 after those", 20, 0), ("1000 timed", 1000, 0), ("20 after those", 20, 0)):
    med, mean = bench.time_steps(step, steps, warm)
    print(f"{label:30s}: median {med:.4f} ms, mean {mean:.4f} ms")
time.sleep(2.0)
med, mean = bench.time_steps(step, 20, 0)
print(f"{'20 after a 2 s pause':30s}: median {med:.4f} ms, mean {mean:.4f} ms")
