"""Developer check of k_comp_bwd_mix against the row-wise kernel: run with MST_HIP_LIB=<lib> and a tag; saves the gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")]
import torch
from mst.modules import AdvancedMixConsole
tag = sys.argv[1]
bs, T, n = (int(v) for v in sys.argv[2:5])
dev = torch.device("cuda:0")
torch.manual_seed(0)
c = AdvancedMixConsole(44100, materialize_mixed_tracks=False, validate="deferred")
tracks = (0.1 * torch.randn(bs, T, n)).to(dev)
tp = torch.rand(bs, T, 27, device=dev, requires_grad=True)
fp = torch.rand(bs, 25, device=dev)
mp = torch.rand(bs, 26, device=dev, requires_grad=True)
g = torch.randn(bs, 2, n, device=dev)
_, mix, *_ = c(tracks, tp, fp, mp, use_fx_bus=False)
mix.backward(g)
torch.cuda.synchronize()
c.check_parameters()
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", f"dbg_cbm_{tag}_{bs}x{T}x{n}.pt")
torch.save(dict(g_tp=tp.grad.cpu(), g_mp=mp.grad.cpu()), out)
if len(sys.argv) > 5:
    ref = torch.load(out.replace(f"_{tag}_", f"_{sys.argv[5]}_"))
    for k in ("g_tp", "g_mp"):
        a, b = (tp.grad if k == "g_tp" else mp.grad).cpu().double(), ref[k].double()
        d = (a - b).abs()
        print(k, "rel", ((a - b).norm() / b.norm()).item(), "finite", bool(torch.isfinite(a).all()))
        if k == "g_tp":
            per = d.amax(dim=(0,)) / (b.abs().amax(dim=(0,)) + 1e-30)  # (T, 27)
            torch.set_printoptions(precision=2, linewidth=250, sci_mode=True)
            print("per (track, param) max rel diff:\n", per)
        else:
            print("per master param:", (d.amax(0) / (b.abs().amax(0) + 1e-30)))
