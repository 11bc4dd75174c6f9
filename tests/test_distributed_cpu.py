"""world_size-2 gloo test of the batch-axis sharding the multi-GPU bench uses: every rank
owns a contiguous slice of the mixes, computes its local loss, and ONE all-reduce of the scalar gives
the global-batch loss (per-example terms), with no data-path collective."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench
from oracle import loss_restated as ol

RES = ((512, 256, 512), (2048, 1024, 2048))


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)  # every rank can rebuild the global batch; it only touches its shard
    gbs = 4
    x, y = torch.randn(gbs, 2, 8192), torch.randn(gbs, 2, 8192)
    lo, hi = bench.shard_batch(gbs, rank, world)
    local = ol.mrstft_loss(x[lo:hi], y[lo:hi], RES)
    out = bench.reduce_loss(local.detach().clone(), world)
    ret[rank] = (lo, hi, out.item())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_loss_matches_global_batch():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        ret = dict(ret)
    assert (ret[0][0], ret[0][1], ret[1][0], ret[1][1]) == (0, 2, 2, 4)
    torch.manual_seed(0)
    x, y = torch.randn(4, 2, 8192), torch.randn(4, 2, 8192)
    glob = ol.mrstft_loss(x, y, RES).item()  # per-example spectral convergence => shard-mean == global
    assert abs(ret[0][2] - glob) / glob < 1e-6 and abs(ret[1][2] - ret[0][2]) < 1e-9
