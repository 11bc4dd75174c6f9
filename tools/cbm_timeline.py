"""Developer timeline of k_comp_bwd_mix (needs the -DMST_CBR_STAMPS build, MST_HIP_LIB=.../stamps.so): per-phase wall-clock durations of
the second track iteration of every workgroup + workgroup lifetimes + occupancy over time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import numpy as np
import torch
from diffmst_hip import _hip
from mst.modules import AdvancedMixConsole
bs, T, n = 8, 8, 262144
dev = torch.device("cuda:0")
torch.manual_seed(0)
c = AdvancedMixConsole(44100, materialize_mixed_tracks=False, validate="deferred")
tracks = (0.1 * torch.randn(bs, T, n)).to(dev)
tp = torch.rand(bs, T, 27, device=dev, requires_grad=True)
fp = torch.rand(bs, 25, device=dev)
mp = torch.rand(bs, 26, device=dev, requires_grad=True)
g = torch.randn(bs, 2, n, device=dev)
for _ in range(4):
    tp.grad = None; mp.grad = None
    _, mix, *_ = c(tracks, tp, fp, mp, use_fx_bus=False)
    mix.backward(g)
torch.cuda.synchronize()
SLOTS, WGS = 12, 16384
buf = np.zeros(WGS * SLOTS, dtype=np.uint64)
fn = _hip.lib().mst_debug_read_cbr_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]; fn.restype = ctypes.c_int
assert fn(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(WGS, SLOTS).astype(np.int64)[: 128 * 16]
t0, tend = st[:, 9].min(), st[:, 10].max()
life = (st[:, 10] - st[:, 9]) / 100.0
print(f"launch span {(tend - t0) / 100:.1f} us; workgroup life median {np.median(life):.2f} us p10 {np.percentile(life, 10):.2f} p90 {np.percentile(life, 90):.2f}; "
      f"prologue + ride (start -> first track) median {np.median(st[:, 11] - st[:, 9]) / 100:.2f} us")
names = ["loads issued -> zq phase done", "scan + barrier", "sw/publish/flush/prefetch issue/fwd", "tile store + static curve", "granule wait + wave_sum",
         "finish du, sums, tile store", "walk", "partial sums + sync"]
for k, nm in enumerate(names):
    d = (st[:, k + 1] - st[:, k]) / 100.0
    print(f"  phase {k}->{k + 1} {nm:42s} median {np.median(d):6.2f} us  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}")
d = (st[:, 8] - st[:, 0]) / 100.0
print(f"  one track iteration: median {np.median(d):.2f} us p10 {np.percentile(d, 10):.2f} p90 {np.percentile(d, 90):.2f}")
edges = np.arange(0, (tend - t0) / 100 + 5, 5.0)
print("alive workgroups per 5 us bin:", [int((((st[:, 9] - t0) / 100 < e + 5) & ((st[:, 10] - t0) / 100 > e)).sum()) for e in edges[:-1]])
print("start times (us) pct 0/25/50/75/100:", [round(float(np.percentile((st[:, 9] - t0) / 100, p)), 1) for p in (0, 25, 50, 75, 100)])
