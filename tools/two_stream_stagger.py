"""Probe: two half-batches (4 mixes each) of the cfg #2 step on two streams, the second one offset in time - does the latency-bound
stretches of one half (k_prep, master chain, finish) hide under the execution-bound kernels of the other?  Steady state over many
steps; reports ms per FULL step (8 mixes).  usage: python tools/two_stream_stagger.py [offset_us ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench

dev = torch.device("cuda:0")
full = bench.make_workload(dev, 8, 8, bench.N, "mrstft", seed=1000)
halves = [bench.make_workload(dev, 4, 8, bench.N, "mrstft", seed=1000 + i) for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def run_full(steps):
    for _ in range(10): full()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): full()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3

def run_two(steps, offset_us):
    for s, h in zip(streams, halves):
        with torch.cuda.stream(s):
            for _ in range(5): h()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[1]):
        if offset_us: torch.cuda._sleep(int(offset_us * 2100))  # cycles at ~2.1 GHz
    for _ in range(steps):
        for s, h in zip(streams, halves):
            with torch.cuda.stream(s):
                h()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

print(f"full batch, one stream: {run_full(100):.4f} ms/step")
for off in [float(a) for a in sys.argv[1:]] or [0.0, 60.0, 110.0, 170.0]:
    print(f"two halves, two streams, second offset {off:.0f} us: {run_two(100, off):.4f} ms per full step")
# host-only cost of the two-half loop
t0 = time.perf_counter()
for _ in range(50):
    for s, h in zip(streams, halves):
        with torch.cuda.stream(s):
            h()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host enqueue of two half-steps: {(t1 - t0) / 50 * 1e3:.4f} ms")

# device-side potential: each half-step captured as a hipGraph, the two graphs replayed on two streams with an offset
graphs = []
for s, h in zip(streams, halves):
    with torch.cuda.stream(s):
        for _ in range(3): h()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        h()
    graphs.append(g)
torch.cuda.synchronize()
def run_graphs(steps, offset_us):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(streams[1]):
        if offset_us: torch.cuda._sleep(int(offset_us * 2100))
    for _ in range(steps):
        for s, g in zip(streams, graphs):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
gf = torch.cuda.CUDAGraph()
for _ in range(3): full()
torch.cuda.synchronize()
with torch.cuda.graph(gf):
    full()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): gf.replay()
torch.cuda.synchronize(); print(f"full batch as ONE graph: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step")
for off in (0.0, 40.0, 80.0, 120.0, 160.0, 200.0):
    print(f"two half-step graphs on two streams, offset {off:.0f} us: {run_graphs(200, off):.4f} ms per full step")
