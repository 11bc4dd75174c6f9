"""Developer probe: (1) which torch op issues the device-to-device copies of a bench step; (2) console fwd+bwd time by batch size at
16 tracks (does walking cfg #3 in mix groups pay?).  python tools/probe_groups.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
step = bench.make_workload(dev, 8, 8, bench.N, "mrstft", seed=1)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if "copy" in e.key.lower() or "clone" in e.key.lower() or "fill" in e.key.lower() or "Memcpy" in e.key or "zero" in e.key.lower()]
for e in rows:
    print(f"{e.key[:70]:70s} n={e.count:4d} cpu={e.cpu_time_total:9.1f}us dev={getattr(e, 'device_time_total', 0):9.1f}us")
# every event whose name is a copy, with the op that encloses it
evs = sorted(prof.events(), key=lambda e: e.time_range.start)
for e in evs:
    if e.name in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_"):
        par, chain = e.cpu_parent, []
        while par is not None and len(chain) < 6:
            chain.append(par.name)
            par = par.cpu_parent
        print("  ", e.name, [tuple(s) for s in (e.input_shapes or [])][:2], "<-", " <- ".join(chain))

for T, kinds in ((16, ("dot", "af")), (8, ("dot",))):
    for kind in kinds:
        for bs in (2, 4, 8, 16, 32):
            st = bench.make_workload(dev, bs, T, bench.N, kind, seed=3)
            med, mean = bench.time_steps(st, 8, 3)
            print(f"T {T} loss {kind}: bs {bs:2d}: {med:.3f} ms/step = {med / bs * 1e3:.1f} us per mix; 32 mixes in groups of {bs}: {med * 32 / bs:.3f} ms")
            del st
            torch.cuda.empty_cache()
