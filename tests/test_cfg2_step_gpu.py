"""BASELINE cfg #2 exactly as `bench.py` times it (round-5 review, weak 4 / item 5a): the step object `bench.make_workload` builds - lean
console (no `mixed_tracks`, `validate="deferred"`, lazy parameter dictionaries) at 8 mixes x 8 tracks x 262144 samples chained into
`MultiResolutionSTFTLoss` (512 / 2048 / 8192) and back through both backward launches to the 27 + 26 parameter tensors - against the
oracle (fp32 = the reference's algorithm, float64 = truth) driven with the SAME tensors: loss and both parameter gradients, three-way.

The oracle runs mix by mix on the host (the loss is a mean of per-example terms with `sc_per_example=True`, the default:
loss = mean_b loss_b and d loss / d params_b = (1 / bs) d loss_b / d params_b), which bounds its memory to one mix of 2^19-point
frequency-sampling filters in float64."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from util import rel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(tracks, ref, tp, fp, mp, flags, res, dtype):
    from oracle import console_restated as oc
    from oracle import loss_restated as ol

    bs = tracks.shape[0]
    loss, g_tp, g_mp = 0.0, [], []
    for b in range(bs):
        t_b = tp[b:b + 1].detach().clone().to(dtype).requires_grad_(True)
        m_b = mp[b:b + 1].detach().clone().to(dtype).requires_grad_(True)
        _, mix, *_ = oc.console_forward(tracks[b:b + 1].to(dtype), t_b, fp[b:b + 1].to(dtype), m_b, **flags)
        l_b = ol.mrstft_loss(mix, ref[b:b + 1].to(dtype), res)
        l_b.backward()
        loss += l_b.item() / bs
        g_tp.append(t_b.grad / bs)
        g_mp.append(m_b.grad / bs)
    return loss, torch.cat(g_tp), torch.cat(g_mp)


def test_cfg2_step_as_benchmarked(record):
    assert torch.cuda.is_available()
    sys.path.insert(0, ROOT)
    import bench

    dev = torch.device("cuda:0")
    step = bench.make_workload(dev, bench.BS, bench.T, bench.N, "mrstft", seed=1000, lean=True)  # bench.py main(): rank 0's workload
    console = step.console
    assert console.materialize_mixed_tracks is False and console.validate == "deferred" and console.param_dicts == "lazy"
    loss = step()
    loss_again = step()  # the timed loop calls it back to back on the same tensors: bit-identical
    torch.cuda.synchronize()
    console.check_parameters()
    tp, mp = step.params
    g_tp, g_mp = tp.grad.detach().cpu(), mp.grad.detach().cpu()
    assert torch.equal(loss, loss_again)
    step()
    assert torch.equal(tp.grad.cpu(), g_tp) and torch.equal(mp.grad.cpu(), g_mp)

    cpu = lambda t: t.detach().cpu()
    tracks, ref, fp = cpu(step.inputs["tracks"]), cpu(step.inputs["ref"]), cpu(step.inputs["fx_params"])
    res = tuple(zip(bench.RESOLUTIONS["fft_sizes"], bench.RESOLUTIONS["hop_sizes"], bench.RESOLUTIONS["win_lengths"]))
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 16))  # the measured optimum of these 2^19-point FFT batches (bench.py: cpu_baseline)
    try:
        l32, t32, m32 = _oracle(tracks, ref, cpu(tp), fp, cpu(mp), bench.FLAGS, res, torch.float32)
        l64, t64, m64 = _oracle(tracks, ref, cpu(tp), fp, cpu(mp), bench.FLAGS, res, torch.float64)
    finally:
        torch.set_num_threads(threads)
    e_loss, r_loss = abs(loss.item() - l64) / abs(l64), abs(l32 - l64) / abs(l64)
    rep = dict(loss=(abs(loss.item() - l32) / abs(l32), e_loss, r_loss),
               g_tp=(rel(g_tp, t32), rel(g_tp, t64), rel(t32, t64)), g_mp=(rel(g_mp, m32), rel(g_mp, m64), rel(m32, m64)))
    print("\n[cfg #2 step as benchmarked] (hip vs ref32, hip vs f64, ref32 vs f64):", rep)
    record(**rep)
    # the loss is a mean over > 1e7 terms of a mix that meets 1e-4: north-star tolerance on the scalar, and no further from float64 than
    # the reference's own fp32 evaluation (+ 1e-5)
    assert e_loss < 1e-4 and e_loss <= 2 * r_loss + 1e-5, rep["loss"]
    # parameter gradients three-way, the bound of every other gradient test of the suite
    for k in ("g_tp", "g_mp"):
        h32, h64, r = rep[k]
        assert h64 <= 2 * r + 1e-4, (k, rep[k])
        assert h32 < 1e-2, (k, rep[k])
