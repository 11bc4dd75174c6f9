"""Developer check: native encoder stack / torch fp32 / torch fp64 three-way on the controller (python tools/dbg_ctrl.py)."""
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone"), os.path.join(ROOT, "tests")]
import torch
from mst.modules import TransformerController
from test_controller_gpu import _run
dev = torch.device("cuda:0")
for (bs, T, layers, mk, seed) in [(2, 6, 2, True, 13), (2, 6, 2, True, 1), (2, 6, 2, True, 2), (2, 6, 2, True, 3), (2, 6, 3, True, 4), (4, 6, 2, True, 5)]:
    torch.manual_seed(seed)
    ctrl = TransformerController(512, 27, 25, 26, num_layers=layers, nhead=8).to(dev).train()
    with torch.no_grad():
        for n, p in ctrl.named_parameters():
            if "norm" in n or n.endswith("bias"):
                p.add_(0.1 * torch.randn_like(p))
    te = torch.randn(bs, T, 512, device=dev); me = torch.randn(bs, 2, 512, device=dev)
    mask = torch.zeros(bs, T, dtype=torch.bool, device=dev); mask[0, T // 2:] = True; mask[bs - 1, 1] = True
    w = [torch.randn(bs, T, 27, device=dev), None, torch.randn(bs, 26, device=dev)]
    ctrl.native = True; on, gn, pn = _run(ctrl, te, me, mask, w)
    ctrl.native = False; oe, ge, pe = _run(ctrl, te, me, mask, w)
    c64 = copy.deepcopy(ctrl).double()
    o64, g64, p64 = _run(c64, te.double(), me.double(), mask, [w[0].double(), None, w[2].double()])
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    print((bs, T, layers), "grad_in  native-fp64", [rel(a, b) for a, b in zip(gn, g64)], " torch32-fp64", [rel(a, b) for a, b in zip(ge, g64)])
    print("   params native-fp64 %.2e  torch32-fp64 %.2e" % (max(rel(pn[n], p64[n]) for n in pe), max(rel(pe[n], p64[n]) for n in pe)))
    bad = sorted(((rel(pn[n], p64[n]), n) for n in pe), reverse=True)[:8]
    print("   worst params:", [(f"{e:.1e}", n.replace("transformer_encoder.layers.", "L")) for e, n in bad])
    for a, b, nm in zip(gn, g64, ("te", "me")):
        print("   grad", nm, ((a.double() - b).abs().amax(-1) / b.abs().max()).cpu().numpy().round(5).tolist())
