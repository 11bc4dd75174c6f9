"""Where does the HIP parameter gradient of a console fixture differ from the float64 one?  usage: dbg_golden_grad.py <fixture.npz>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from mst.modules import AdvancedMixConsole
g = np.load(sys.argv[1], allow_pickle=True)
flags = {k: v == "True" for k, v in g["flags"]}
t = lambda k: torch.from_numpy(g[k]).float()
dev = torch.device("cuda:0")
c = AdvancedMixConsole(44100)
tp = t("track_params").to(dev).requires_grad_(True); mp = t("master_bus_params").to(dev).requires_grad_(True)
_, mix, *_ = c(t("tracks").to(dev), tp, t("fx_bus_params").to(dev), mp, **flags)
(mix * t("grad_mix").to(dev)).sum().backward()
h = tp.grad.double().cpu(); r = t("grad_track_params").double(); f = torch.from_numpy(g["grad_track_params_f64"])
tot = f.norm()
print("overall hip-f64", ((h - f).norm() / tot).item(), "ref32-f64", ((r - f).norm() / tot).item())
eh, er = (h - f).abs() / tot, (r - f).abs() / tot
idx = torch.argsort(eh.flatten(), descending=True)[:12]
names = ["gain"] + [f"{b}.{p}" for b in ("ls", "b0", "b1", "b2", "b3", "hs") for p in ("g", "f", "q")] + ["thr", "ratio", "att", "rel", "knee", "mk", "pan", "send"]
for i in idx:
    b, tr, p = np.unravel_index(i.item(), f.shape)
    print(f"mix {b} track {tr} {names[p]:6s} f64 {f[b,tr,p].item():+.5e}  hip err {eh[b,tr,p].item():.2e}  ref32 err {er[b,tr,p].item():.2e}   param {g['track_params'][b,tr,p]:.4f}")
