"""Developer: where the host time of the cfg #1 eager step goes (cProfile over 300 steps, the device far ahead)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench
dev = torch.device("cuda:0")
step = bench.make_workload(dev, seed=1, bs=2, n_tracks=4, n=65536, loss_kind="none", lean=True, basic=True)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host {1e6*(t1-t0)/300:.1f} us/step, with drain {1e6*(t2-t0)/300:.1f} us/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "cumulative").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 35)
