"""Which host-side op issues the device-to-device copies of the cfg #2 step?  (torch.profiler, three steps)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
step = bench.make_workload(dev, bench.BS, bench.T, bench.N, "mrstft", seed=1000, lean=True)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
for e in prof.events():
    n = e.name
    if "copy" in n.lower() or "clone" in n.lower() or "Memcpy" in n or "fill" in n.lower() or "zero" in n.lower():
        print(f"{n[:60]:60s} dev={e.device_type} cpu_t={e.cpu_time_total:.1f} shapes={getattr(e,'input_shapes',None)} stack={e.stack[:3] if e.stack else None}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
