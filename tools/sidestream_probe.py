"""Probe: how much of a forward MR-STFT launch hides under the console forward when it runs on a side stream from the step's start?
(cfg #2; the side launch is the existing fused forward on dummy rows, forward only.)  Prints ms per step for both arrangements."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
import bench

dev = torch.device("cuda:0")
step = bench.make_workload(dev, 8, 8, 262144, "mrstft", 0)
from mst.loss import MultiResolutionSTFTLoss
loss2 = MultiResolutionSTFTLoss(**bench.RESOLUTIONS)
a = (0.1 * torch.randn(8, 2, 262144)).to(dev)
b = (0.1 * torch.randn(8, 2, 262144)).to(dev)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print("priority range: least", lo.value, "greatest", hi.value)
PRI = int(os.environ.get("SIDE_PRI", "0"))
side = torch.cuda.Stream(priority=PRI)
print("side priority", side.priority)
main = torch.cuda.current_stream()

from diffmst_hip import _hip
L = _hip.lib()
ev = torch.cuda.Event()
ev.record()
torch.cuda.synchronize()
has_probe = hasattr(L, "mst_debug_set_mid_event")
if has_probe:
    L.mst_debug_set_mid_event.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.mst_debug_set_mid_event.restype = None

console, (tp, mp) = step.console, step.params
ref = (0.1 * torch.randn(8, 2, 262144)).to(dev)
fxp = torch.rand(8, 25).to(dev)
tracks = (0.1 * torch.randn(8, 8, 262144)).to(dev)
seed_grad = torch.ones((), device=dev)
loss_main = MultiResolutionSTFTLoss(**bench.RESOLUTIONS)

def step_mid(where):
    """own step: console forward (records the event in the middle), side launch gated on it, main waits for the side before its loss"""
    def f():
        tp.grad = None; mp.grad = None
        if has_probe:
            L.mst_debug_set_mid_event(ctypes.c_void_p(ev.cuda_event), where)
        _, mix, *_ = console(tracks, tp, fxp, mp, **bench.FLAGS)
        if where >= 0:
            side.wait_event(ev)
            with torch.cuda.stream(side), torch.no_grad():
                loss2(a, b)
            main.wait_stream(side)
        loss = loss_main(mix, ref)
        torch.autograd.backward(loss, grad_tensors=seed_grad.expand_as(loss))
        return loss.detach()
    return f

def step_side(where):
    def f():
        if where == "start":
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                loss2(a, b)
        out = step()
        main.wait_stream(side)
        return out
    return f

def solo():
    with torch.no_grad():
        loss2(a, b)

runs = [("base", step), ("side@start", step_side("start"))]
if has_probe:
    runs += [("own step, no side", step_mid(-1)), ("side after k_prep", step_mid(0)), ("side after apply_tracks", step_mid(1)),
             ("own step, no side", step_mid(-1)), ("side after k_prep", step_mid(0)), ("side after apply_tracks", step_mid(1))]
runs += [("loss fwd alone", solo)]
for name, fn in runs:
    med, mean = bench.time_steps(fn, 200, 20)
    print(f"{name:24s} median {med*1000:.1f} us  mean {mean*1000:.1f} us", flush=True)
