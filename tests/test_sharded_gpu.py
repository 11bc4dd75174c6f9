"""The HIP path under batch sharding (reference: DDP over the batch axis, configs/config.yaml:34-42), on ONE GPU:
two processes on cuda:0, gloo for the collectives.  Each rank owns a contiguous slice of the mixes (bench.shard_batch), runs
the HIP console + HIP MR-STFT on it and all-reduces the loss scalar only (bench.reduce_loss); the result must be the
single-process HIP run over the global batch."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from util import FULL, rel

GBS, T, N = 4, 4, 131072
RES = dict(fft_sizes=[512, 2048, 8192], hop_sizes=[256, 1024, 4096], win_lengths=[512, 2048, 8192])


def _inputs():
    torch.manual_seed(1234)  # every rank can rebuild the global batch; it only touches its shard
    tracks = 0.1 * torch.randn(GBS, T, N)
    ref = 0.2 * torch.randn(GBS, 2, N)
    return tracks, ref, torch.rand(GBS, T, 27), torch.rand(GBS, 25), torch.rand(GBS, 26)


def _step(lo, hi, loss_kw):
    """HIP console + MR-STFT on mixes [lo, hi) -> (loss, grad track params, grad master params), all on the CPU."""
    from mst.loss import MultiResolutionSTFTLoss
    from mst.modules import AdvancedMixConsole

    dev = torch.device("cuda:0")
    tracks, ref, tp, fp, mp_ = (t[lo:hi].to(dev) for t in _inputs())
    tp.requires_grad_(True)
    mp_.requires_grad_(True)
    console = AdvancedMixConsole(44100, materialize_mixed_tracks=False)
    _, mix, *_ = console(tracks, tp, fp, mp_, **FULL)
    loss = MultiResolutionSTFTLoss(**RES, **loss_kw)(mix, ref)
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().cpu(), tp.grad.cpu(), mp_.grad.cpu()


def _worker(rank, world, port, ret):
    import bench

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard_batch(GBS, rank, world)
    out = {}
    for name, kw in (("per_example", {}), ("global_sc", dict(sc_per_example=False, sync_group=True))):
        loss, gtp, gmp = _step(lo, hi, kw)
        out[name] = (bench.reduce_loss(loss.clone(), world).item(), loss.item(), gtp, gmp)
        # the bench's own reduction (queued on a side stream, joined once) must give the same number
        red = bench.AsyncLossReduce(world, torch.device("cuda:0"), 4)
        red.push(loss.to("cuda:0"))
        assert abs(red.join().item() - out[name][0]) <= 1e-7 * abs(out[name][0])
    ret[rank] = (lo, hi, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hip_step_equals_single_process(record):
    assert torch.cuda.is_available()
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        ret = dict(ret)
    assert (ret[0][0], ret[0][1], ret[1][0], ret[1][1]) == (0, 2, 2, 4)
    rep = {}
    for name, kw in (("per_example", {}), ("global_sc", dict(sc_per_example=False))):
        gl, gtp, gmp = _step(0, GBS, kw)  # single process, global batch
        reduced = [ret[r][2][name][0] for r in range(world)]
        assert reduced[0] == reduced[1]
        e_loss = abs(reduced[0] - gl.item()) / gl.item()
        # rank gradients are d(rank loss)/d(own parameters); averaged over ranks (DDP) they are the global-batch gradients
        sh_tp = torch.cat([ret[r][2][name][2] for r in range(world)]) / world
        sh_mp = torch.cat([ret[r][2][name][3] for r in range(world)]) / world
        bit = bool(torch.equal(sh_tp, gtp) and torch.equal(sh_mp, gmp))
        rep[name] = (e_loss, rel(sh_tp, gtp), rel(sh_mp, gmp), float(bit))
        print(f"\n[sharded x{world}, {name}] loss rel err {e_loss:.2e}; grads tp {rep[name][1]:.2e} mp {rep[name][2]:.2e}; bitwise {bit}")
        assert e_loss < 1e-6
        assert rep[name][1] < 1e-5 and rep[name][2] < 1e-5
    record(**rep)
    # per-example terms only: no cross-rank arithmetic, and 1/world is a power of two - the sharded gradients are the
    # single-process ones bit for bit
    assert rep["per_example"][3] == 1.0
    # without the norm exchange the batch-global ratio of a shard is NOT the global one (the exchange is doing something)
    local_only = [ret[r][2]["global_sc"][1] for r in range(world)]
    assert abs(local_only[0] - local_only[1]) > 0
