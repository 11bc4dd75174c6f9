"""Does replaying the cfg #2 step (console fwd + MR-STFT fwd + bwd + console bwd, ~40 launches) as ONE hipGraph beat the eager launch
stream?  Captures bench.py's own step with torch.cuda.CUDAGraph and times eager vs replay with HIP events; checks the replayed gradients
against the eager ones.  usage: python tools/graph_probe.py [bs] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch

import bench

bs = int(sys.argv[1]) if len(sys.argv) > 1 else bench.BS
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
step = bench.make_workload(dev, bs, bench.T, bench.N, "mrstft", seed=1000, lean=True)
tp, mp = step.params

med, mean = bench.time_steps(step, steps, 10)
print(f"eager : {med:.4f} ms median, {mean:.4f} mean per step of {bs} mixes")
loss_e = step().clone()
g_e = (tp.grad.clone(), mp.grad.clone())

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss_g = step()
torch.cuda.synchronize()


def replay():
    graph.replay()


med, mean = bench.time_steps(replay, steps, 10)
print(f"graph : {med:.4f} ms median, {mean:.4f} mean per step of {bs} mixes")
torch.cuda.synchronize()
rel = lambda a, b: float((a - b).norm() / b.norm())
print(f"replayed vs eager: loss {abs(float(loss_g) - float(loss_e)):.3e}  g_tp {rel(tp.grad, g_e[0]):.3e}  g_mp {rel(mp.grad, g_e[1]):.3e}")
step.console.check_parameters()
