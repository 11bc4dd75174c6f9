"""Do two HIP streams of this process run kernels side by side?  Two latency-bound launches (one long row cumsum each), same stream vs two streams."""
import torch, statistics, os
dev = torch.device("cuda:0")
x = torch.randn(4, 1 << 22, device=dev)
y = torch.randn(4, 1 << 22, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
main = torch.cuda.current_stream()
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    e[0].record()
    for i in range(n):
        fn(); e[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(e[i].elapsed_time(e[i + 1]) for i in range(n)) * 1000
def same():
    torch.cumsum(x, 1); torch.cumsum(y, 1)
def two(sa, sb):
    def f():
        sa.wait_stream(main); sb.wait_stream(main)
        with torch.cuda.stream(sa): torch.cumsum(x, 1)
        with torch.cuda.stream(sb): torch.cumsum(y, 1)
        main.wait_stream(sa); main.wait_stream(sb)
    return f
def main_plus(sb):
    def f():
        sb.wait_stream(main)
        with torch.cuda.stream(sb): torch.cumsum(y, 1)
        torch.cumsum(x, 1)
        main.wait_stream(sb)
    return f
print("env GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "main is default stream:", main == torch.cuda.default_stream())
print("one cumsum      ", t(lambda: torch.cumsum(x, 1)))
print("same stream x2  ", t(same))
print("two side streams", t(two(s1, s2)))
print("main + side     ", t(main_plus(s2)))
