"""Developer: host time of the cfg #2 step (cProfile over 200 steps; the device is ~0.5 ms per step, the host must stay ahead)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench
dev = torch.device("cuda:0")
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
step = bench.make_workload(dev, bs, 8, 16384, "mrstft", seed=1)   # tiny device work: the loop below is host-bound
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host {1e6*(t1-t0)/200:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 35)
