"""``TransformerController(graphed=True)`` (reference mst/modules.py:809-914): the hipGraph replay of the training-mode forward and
backward against the eager evaluation of the same module - same kernels, so the bound is rounding-level - over several calls
(static buffers reused), with and without the padding mask, and with the parameters updated in place between calls."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(ctrl, te, me, mask, w):
    te = te.clone().requires_grad_(True)
    me = me.clone().requires_grad_(True)
    for p in ctrl.parameters():
        p.grad = None
    tp, fp, mp = ctrl(te, me, mask)
    loss = (tp * w[0]).sum() + (mp * w[2]).sum()  # the fx-bus output stays unused, as in the reference's configs
    loss.backward()
    return ([tp.detach().clone(), fp.detach().clone(), mp.detach().clone()], [te.grad.clone(), me.grad.clone()],
            {n: p.grad.clone() for n, p in ctrl.named_parameters() if p.grad is not None})


@pytest.mark.parametrize("with_mask", [False, True])
def test_graphed_controller_equals_eager(with_mask, record):
    from mst.modules import TransformerController

    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    ctrl = TransformerController(512, 27, 25, 26, num_layers=3, nhead=8, graphed=True).to(dev).train()
    keys = set(ctrl.state_dict().keys())
    bs, T = 2, 6
    worst = 0.0
    for it in range(3):
        te = torch.randn(bs, T, 512, device=dev)
        me = torch.randn(bs, 2, 512, device=dev)
        mask = None
        if with_mask:
            mask = torch.zeros(bs, T, dtype=torch.bool, device=dev)
            mask[0, T - 1 - it] = True
        w = [torch.randn(bs, T, 27, device=dev), None, torch.randn(bs, 26, device=dev)]
        ctrl.graphed = True
        out_g, gin_g, gp_g = _run(ctrl, te, me, mask, w)
        ctrl.graphed = False
        out_e, gin_e, gp_e = _run(ctrl, te, me, mask, w)
        assert set(gp_g) == set(gp_e) or set(gp_e) <= set(gp_g)
        for a, b in list(zip(out_g, out_e)) + list(zip(gin_g, gin_e)) + [(gp_g[n], gp_e[n]) for n in gp_e]:
            err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            worst = max(worst, err)
            assert err <= 1e-5, err
        with torch.no_grad():  # an in-place optimizer step between replays: the graphs read the updated weights
            for p in ctrl.parameters():
                p.add_(0.01 * torch.randn_like(p))
    assert len(ctrl._graphs) == 1
    assert set(ctrl.state_dict().keys()) == keys  # the captured wrapper did not register itself under the controller
    record(max_rel_err=worst)
