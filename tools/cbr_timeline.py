"""Developer timeline of k_comp_bwd_run<tracks> (needs a -DMST_CBR_STAMPS build: make BUILD=build_stamps OUT=../lib/stamps.so
EXTRA=-DMST_CBR_STAMPS; run with MST_HIP_LIB=.../stamps.so).  Wave 0 of every workgroup stamps the 100 MHz wall clock at its phase
boundaries; this prints the per-phase durations (median / p90 over the track workgroups) and the launch's occupancy over time."""
import ctypes
import os
import sys

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
    tp.grad = None
    mp.grad = None
    _, mix, *_ = c(tracks, tp, fp, mp, use_fx_bus=False)
    mix.backward(g)
torch.cuda.synchronize()
SLOTS, WGS = 12, 16384
buf = np.zeros(WGS * SLOTS, dtype=np.uint64)
lib = _hip.lib()
fn = lib.mst_debug_read_cbr_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
fn.restype = ctypes.c_int
assert fn(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(WGS, SLOTS).astype(np.int64)
nblk = 128
light = st[: 16 * nblk]
heavy = st[16 * nblk: (16 + 64) * nblk]
t0 = min(light[:, 10].min(), heavy[:, 0].min())
tend = heavy[:, 9].max()
print(f"launch span {(tend - t0) / 100:.1f} us; light workgroups: first start {0:.1f}, last end {(light[:, 11].max() - t0) / 100:.1f} us, "
      f"median life {np.median(light[:, 11] - light[:, 10]) / 100:.2f} us")
names = ["load gy/xd0/gs -> zq loop", "aggregate + publish", "load gF/gyF -> fwd0", "load x0 + block_enter_split", "static-curve loop", "granule wait (block_carry_g)",
         "finish du / sums", "coefgrad (transpose, states, walk)", "partial-sum reduction + store"]
print(f"heavy workgroups: first start {(heavy[:, 0].min() - t0) / 100:.1f} us, median life {np.median(heavy[:, 9] - heavy[:, 0]) / 100:.2f} us, "
      f"p90 {np.percentile(heavy[:, 9] - heavy[:, 0], 90) / 100:.2f}")
for k, nm in enumerate(names):
    d = (heavy[:, k + 1] - heavy[:, k]) / 100.0
    print(f"  phase {k}->{k + 1} {nm:42s} median {np.median(d):6.2f} us  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}")
# occupancy over time: number of heavy workgroups alive in 5 us bins
edges = np.arange(0, (tend - t0) / 100 + 5, 5.0)
alive = [(((heavy[:, 0] - t0) / 100 < e + 5) & ((heavy[:, 9] - t0) / 100 > e)).sum() for e in edges[:-1]]
print("alive heavy workgroups per 5 us bin:", alive)
start_order = np.argsort(heavy[:, 0])
print("start time of heavy workgroups (us) at percentiles 0/25/50/75/100:", [round(float(np.percentile((heavy[:, 0] - t0) / 100, p)), 1) for p in (0, 25, 50, 75, 100)])
