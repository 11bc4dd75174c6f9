"""Encoder alone under SyncBatchNorm on two processes (one GPU, gloo) against the single-process global batch: per-parameter error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone"), os.path.join(ROOT, "tests")]
import torch, torch.distributed as dist, torch.multiprocessing as mp

NS, N = int(os.environ.get("NS", 4)), 65536

def data():
    torch.manual_seed(7)
    return 0.1 * torch.randn(NS, 1, N), torch.randn(NS, 128)

def enc(sync):
    from mst.modules import SpectrogramEncoder
    torch.manual_seed(3)
    e = SpectrogramEncoder(embed_dim=128, precision=os.environ.get("PREC", "fp32")).cuda().train()
    return torch.nn.SyncBatchNorm.convert_sync_batchnorm(e) if sync else e

def run(e, lo, hi, scale):
    x, w = data()
    out = e(x[lo:hi].cuda())
    (out * w[lo:hi].cuda()).sum().mul(scale).backward()
    torch.cuda.synchronize()
    return out.detach().cpu(), {n: p.grad.cpu() for n, p in e.named_parameters()}

def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = NS // world
    ret[rank] = run(enc(True), rank * per, (rank + 1) * per, 1.0 / per)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()
    o1, g1 = run(enc(False), 0, NS, 1.0 / NS)
    o1b, g1b = run(enc(False), 0, NS, 1.0 / NS)
    with mp.Manager() as m:
        ret = m.dict(); mp.spawn(worker, args=(2, 29611, ret), nprocs=2, join=True); ret = dict(ret)
    o2 = torch.cat([ret[0][0], ret[1][0]])
    print("embed", rel(o2, o1), "rerun", rel(o1b, o1))
    for k in g1:
        avg = 0.5 * (ret[0][1][k] + ret[1][1][k])
        print(f"{k:40s} ddp-avg vs single {rel(avg, g1[k]):.2e}   |g| {g1[k].norm().item():.2e}")
