"""cfg #1 (BasicMixConsole 4 x 65536, batch 2, fwd+bwd) step time, eager and replayed as one hipGraph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch, bench
dev = torch.device("cuda:0")
step = bench.make_workload(dev, seed=2000, bs=2, n_tracks=4, n=65536, loss_kind="none", lean=True, basic=True)
med, mean = bench.time_steps(step, 200, 20)
print(f"cfg #1 step, eager: median {1e3*med:.1f} us, mean {1e3*mean:.1f} us")
ref = step().clone(); gref = step.params[0].grad.clone()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
med, mean = bench.time_steps(g.replay, 200, 20)
torch.cuda.synchronize()
print(f"cfg #1 step, one hipGraph: median {1e3*med:.1f} us, mean {1e3*mean:.1f} us; bit-equal {torch.equal(out, ref) and torch.equal(step.params[0].grad, gref)}")
