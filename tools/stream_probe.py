"""Developer probe (GPU box): what does mix-group concurrency buy on the cfg #2 step, and what does a HIP
stream fork/join cost here?  Not the contract bench.
  python tools/stream_probe.py [steps=40]
1. whole step (console fwd + MR-STFT + bwd) at bs 8 on one stream  vs  G groups of 8/G mixes on G torch streams
2. fork/join latency: chain of tiny kernels on one stream vs ping-pong between two streams through events"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def wall(fn, steps, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


base = bench.make_workload(dev, 8, 8, bench.N, "mrstft", seed=1)
print(f"one stream, bs 8: {wall(base, steps):.3f} ms/step", flush=True)
# host-only cost of issuing one step (GPU idle-side check): enqueue without waiting, then sync
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    base()
t_host = 1e3 * (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
print(f"host enqueue time per step (bs 8): {t_host:.3f} ms", flush=True)
del base

for G in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(G)]
    parts = []
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            parts.append(bench.make_workload(dev, 8 // G, 8, bench.N, "mrstft", seed=10 + g))
    torch.cuda.synchronize()

    def grouped():
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                parts[g]()

    print(f"{G} streams x bs {8 // G}: {wall(grouped, steps):.3f} ms/step", flush=True)
    one = parts[0]
    print(f"   (a single group of bs {8 // G} alone on one stream: {wall(one, steps):.3f} ms)", flush=True)
    del parts, one

# ---- fork/join latency with tiny kernels
x = torch.zeros(256, device=dev)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
K = 200


def chain_one():
    with torch.cuda.stream(s0):
        for _ in range(K):
            x.add_(1.0)


def chain_pingpong():
    for i in range(K // 2):
        with torch.cuda.stream(s0):
            x.add_(1.0)
            e = torch.cuda.Event(); e.record(s0)
        with torch.cuda.stream(s1):
            s1.wait_event(e)
            x.add_(1.0)
            e2 = torch.cuda.Event(); e2.record(s1)
        s0.wait_event(e2)


for name, fn in (("one stream", chain_one), ("ping-pong two streams", chain_pingpong)):
    print(f"{K} tiny kernels, {name}: {1e3 * wall(fn, 5, 2) / K:.2f} us per kernel", flush=True)
