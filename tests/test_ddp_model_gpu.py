"""cfg #5 on more than one rank (reference configs/config.yaml:40-41: ``strategy: ddp_find_unused_parameters_true``,
``sync_batchnorm: true``; optimiser over ``self.model.parameters()`` mst/system.py:419-423), on ONE GPU: two processes on cuda:0,
gloo collectives.  Each rank runs the System step (``CommonStep``) on its shard with the parameter-estimation model wrapped the way
Lightning wraps it - ``SyncBatchNorm.convert_sync_batchnorm`` + ``DistributedDataParallel(find_unused_parameters=True)`` (the
fx-bus projection never reaches the loss) - and the gradients DDP leaves on every rank must be the single-process gradients of
the global batch.  That only holds if the fused Cnn14 shares its BatchNorm statistics AND their adjoints across ranks
(``mst_cnn14_forward_sync`` / ``_backward_sync``, reference mst/panns.py:27-85 under SyncBatchNorm)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from util import rel

GBS, T, N, SR = int(os.environ.get('MST_DDP_GBS', 4)), 3, 131072, 44100
AF_WEIGHTS = [0.1, 0.001, 1.0, 1.0, 0.1]


def _batch():
    torch.manual_seed(4321)
    tracks = 0.1 * torch.randn(GBS, T, N)
    ref = 0.2 * torch.randn(GBS, 2, N)
    return tracks, ref


def _model(dev, sync=False):
    from mst.modules import MixStyleTransferModel, SpectrogramEncoder, TransformerController

    torch.manual_seed(99)  # same weights in every process
    model = MixStyleTransferModel(SpectrogramEncoder(embed_dim=128, precision="fp32"), SpectrogramEncoder(embed_dim=128, precision="fp32"),
                                  TransformerController(128, 27, 25, 26, num_layers=2, nhead=8, native=True)).to(dev).train()
    if sync:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    return model


def _step(model, lo, hi):
    """One System step on mixes [lo, hi): loss, {name: gradient} on the CPU, running statistics of the first BatchNorm."""
    from mst.loss import AudioFeatureLoss
    from mst.mixing import naive_random_mix
    from mst.modules import AdvancedMixConsole
    from mst.system import CommonStep

    dev = torch.device("cuda:0")
    tracks, ref = (t[lo:hi].to(dev) for t in _batch())
    step = CommonStep(model, AdvancedMixConsole(SR, materialize_mixed_tracks=False), naive_random_mix, AudioFeatureLoss(AF_WEIGHTS, SR),
                      generate_mix=False, active_eq_epoch=0, active_compressor_epoch=0, active_fx_bus_epoch=1000, active_master_bus_epoch=0)
    batch = (tracks, None, None, torch.zeros(hi - lo, T, dtype=torch.bool, device=dev), ref, ["x"] * (hi - lo))
    model.zero_grad(set_to_none=True)
    loss, _ = step(batch, train=True)
    loss.backward()
    torch.cuda.synchronize()
    core = model.module if hasattr(model, "module") else model
    grads = {n: p.grad.detach().cpu() for n, p in core.named_parameters() if p.grad is not None}
    bn = core.track_encoder.model.conv_block1.bn1
    return loss.detach().cpu(), grads, (bn.running_mean.detach().cpu(), bn.running_var.detach().cpu())


def _worker(rank, world, port, ret, sync):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from torch.nn.parallel import DistributedDataParallel

    model = DistributedDataParallel(_model(dev, sync=sync), device_ids=[0], find_unused_parameters=True)
    per = GBS // world
    loss, grads, run = _step(model, rank * per, (rank + 1) * per)
    ret[rank] = (loss, grads, run)
    dist.barrier()
    dist.destroy_process_group()


def _spawn(sync):
    world = 2
    port = 29500 + ((os.getpid() + (7 if sync else 0)) % 2000)
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret, sync), nprocs=world, join=True)
        return dict(ret)


def test_ddp_model_step_equals_single_process(record):
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    loss1, g1, run1 = _step(_model(dev), 0, GBS)  # single process, global batch, plain BatchNorm over the whole batch
    ret = _spawn(sync=True)
    (l_a, g_a, run_a), (l_b, g_b, run_b) = ret[0], ret[1]
    # DDP has averaged: every rank holds the same gradients
    assert set(g_a) == set(g_b) == set(g1)
    assert all(torch.equal(g_a[k], g_b[k]) for k in g_a)
    assert not any(k.startswith("controller.fx_bus_projection") for k in g_a)  # unused: no gradient, and DDP did not hang on it
    e_loss = abs(0.5 * (l_a + l_b) - loss1).item() / abs(loss1.item())
    errs = {k: rel(g_a[k], g1[k]) for k in g1}
    worst = max(errs, key=errs.get)
    bn = {k: v for k, v in errs.items() if ".bn" in k}
    conv = {k: v for k, v in errs.items() if "conv" in k and ".bn" not in k}
    ctrl = {k: v for k, v in errs.items() if k.startswith("controller.")}
    e_run = max(rel(run_a[0], run1[0]), rel(run_a[1], run1[1]))
    print(f"\n[DDP x2 + SyncBatchNorm vs single process] loss {e_loss:.2e}; worst gradient {worst} {errs[worst]:.2e}; BatchNorm weights <= "
          f"{max(bn.values()):.2e}; conv weights <= {max(conv.values()):.2e}; controller <= {max(ctrl.values()):.2e}; running statistics {e_run:.2e}")
    record(loss=e_loss, worst=errs[worst], bn=max(bn.values()), conv=max(conv.values()), controller=max(ctrl.values()), running=e_run)
    assert e_loss < 1e-5
    assert e_run < 1e-5 and torch.equal(run_a[0], run_b[0])
    # Two fp32 evaluations whose BatchNorm statistics come from different summation trees (per-rank slice sums all-reduced in fp64 vs one
    # pass over the batch) agree to ~1e-7 on the statistics; wherever that moves ONE pre-activation across zero the ReLU mask flips and every
    # layer upstream of it sees a gradient that differs by ~1/sqrt(elements) = 1e-3..1e-2 (either path is equally far from float64:
    # tests/test_encoder_gpu.py).  So: what lies downstream of every mask - the controller, the encoders' final Linear - must
    # agree tightly, everything must agree to the flip scale, and the exchange must matter (below).
    head = {k: v for k, v in errs.items() if ".fc." in k}  # behind the last ReLU: no mask between it and the loss
    if os.environ.get("MST_DDP_VERBOSE"):
        for k in sorted(errs, key=errs.get, reverse=True)[:24]:
            print(f"   {k:56s} {errs[k]:.2e}  |g| {g1[k].norm().item():.2e}")
    assert max(ctrl.values()) < 1e-4 and max(head.values()) < 1e-4, (max(ctrl.values()), max(head.values()))
    assert errs[worst] < 2e-2, (worst, errs[worst])

    # and the exchange is doing something: plain per-rank BatchNorm under the same DDP is a different model
    ret = _spawn(sync=False)
    p_loss = abs(0.5 * (ret[0][0] + ret[1][0]) - loss1).item() / abs(loss1.item())
    p_head = max(rel(ret[0][1][k], g1[k]) for k in head)
    print(f"[DDP x2, per-rank BatchNorm] loss off by {p_loss:.2e} (shared statistics: {e_loss:.2e}), head gradients by {p_head:.2e} "
          f"(shared: {max(head.values()):.2e})")
    record(per_rank_bn_loss=p_loss, per_rank_bn_head=p_head, shared_head=max(head.values()))
    assert p_loss > 100 * e_loss and p_head > 100 * max(head.values())


def _count_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from mst.modules import SpectrogramEncoder

    torch.manual_seed(5)
    enc = torch.nn.SyncBatchNorm.convert_sync_batchnorm(SpectrogramEncoder(embed_dim=32, precision="fp32")).to(dev).train()
    x = 0.1 * torch.randn(3, 1, 65536, device=dev)
    out = []
    # call 1: both ranks feed 2 signals.  call 2: rank 0 feeds 2 again (an `n` it has already seen), rank 1 feeds 3 - the case in which a
    # check cached per (group, n) let rank 0 skip the exchange rank 1 was waiting in (advisor, round 5), and in which the layers'
    # statistics exchanges themselves differ in size.  Every call of every rank runs the same count exchange first and reads it: both
    # ranks raise at call 2, in front of the statistics exchange; call 3 (equal counts again) works
    for n in (2, 2 if rank == 0 else 3, 2):
        try:
            enc(x[:n])
            out.append("ok")
        except RuntimeError as e:
            out.append(str(e))
    torch.cuda.synchronize()
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_signal_count_mismatch_raises_on_every_rank():
    """`diffmst_hip/panns.py: _sync_count_check` - never gates a collective on rank-local state."""
    world, port = 2, 29500 + ((os.getpid() + 13) % 2000)
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_count_worker, args=(world, port, ret), nprocs=world, join=True)
        ret = dict(ret)
    for rank in range(world):
        r = ret[rank]
        assert r[0] == "ok" and r[2] == "ok", r
        assert "same number of signals" in r[1], r
