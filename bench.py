#!/usr/bin/env python3
"""Headline benchmark of the MI355X-native Diff-MST mix-console hot path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one synthetic batch per GPU (BASELINE config #2):
AdvancedMixConsole forward on 8 mixes x 8 tracks x 262144 samples @ 44.1 kHz, MR-STFT loss
(512/2048/8192) against a peak-normalised random reference mix, backward to the 27/26 parameter
tensors.  Inputs are resident in HBM before the timed region.  N > 1 shards the batch axis (weak
scaling: 8 mixes per GPU) and all-reduces the loss scalar only (leaf parameters stay local) - RCCL.

Prints ONE JSON line on rank 0 (see the fields at the bottom).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "diff-mst_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

BS, T, N, SR = 8, 8, 262144, 44100
RESOLUTIONS = dict(fft_sizes=[512, 2048, 8192], hop_sizes=[256, 1024, 4096], win_lengths=[512, 2048, 8192])
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# algorithmic bytes per mix, SURVEY 8(d): console fwd+bwd (lean) 8*N*(T+2) + MR-STFT fwd+bwd 24*N
BYTES_PER_MIX = 8 * N * (T + 2) + 24 * N


def measured_traffic():
    """HBM bytes per step from the committed rocprofv3 PMC passes of this same command (profiles/), or None.
    bench.py cannot run the profiler on itself; the figure is refreshed whenever the kernels change."""
    path = os.path.join(ROOT, "profiles", "round1_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_bytes_per_step_raw"])
    except (OSError, KeyError, ValueError):
        return None


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous batch-axis shard owned by `rank`: mixes [lo, hi)."""
    per = global_batch // world
    assert per * world == global_batch, "global batch must divide evenly"
    return rank * per, (rank + 1) * per


def reduce_loss(loss: torch.Tensor, world: int):
    """Mean of the per-rank losses == single-process loss over the global batch (per-example terms)."""
    if world > 1:
        dist.all_reduce(loss, op=dist.ReduceOp.SUM)
        loss = loss / world
    return loss


def cpu_baseline(seconds_budget: float = 20.0):
    """Oracle (PyTorch-CPU restatement of the reference algorithm: frequency-sampling IIR via torch.fft,
    torch.stft loss) on a bounded sample of the same workload: 1 mix of 8 tracks x 262144, fwd+bwd."""
    from oracle import console_restated as oc
    from oracle import loss_restated as ol

    torch.manual_seed(0)
    tracks = 0.1 * torch.randn(1, T, N)
    fp = torch.rand(1, 25)
    flags = dict(use_track_input_fader=True, use_track_eq=True, use_track_compressor=True, use_track_panner=True,
                 use_fx_bus=False, use_master_bus=True, use_output_fader=True)
    with torch.no_grad():
        _, ref, *_ = oc.console_forward(tracks, torch.rand(1, T, 27), fp, torch.rand(1, 26), **flags)
        ref = oc.batch_stereo_peak_normalize(ref)
    res = tuple(zip(RESOLUTIONS["fft_sizes"], RESOLUTIONS["hop_sizes"], RESOLUTIONS["win_lengths"]))

    def one():
        tp = torch.rand(1, T, 27, requires_grad=True)
        mp = torch.rand(1, 26, requires_grad=True)
        _, mix, *_ = oc.console_forward(tracks, tp, fp, mp, **flags)
        ol.mrstft_loss(mix, ref, res).backward()

    one()  # warm-up
    times = []
    t_start = time.time()
    while len(times) < 5 and (time.time() - t_start < seconds_budget or len(times) < 2):
        t0 = time.time()
        one()
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "mixes/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 mix (8 tracks x 262144) console fwd+bwd + MR-STFT, median of {len(times)} runs, fp32, "
                      f"{os.cpu_count()} host cpus"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl")  # RCCL over xGMI
    assert world == args.gpus or world == 1, (world, args.gpus)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from mst.loss import MultiResolutionSTFTLoss
    from mst.mixing import naive_random_mix
    from mst.modules import AdvancedMixConsole
    from mst.utils import batch_stereo_peak_normalize

    console = AdvancedMixConsole(SR, materialize_mixed_tracks=False, validate="deferred")
    loss_fn = MultiResolutionSTFTLoss(**RESOLUTIONS)
    flags = dict(use_track_input_fader=True, use_track_eq=True, use_track_compressor=True, use_track_panner=True,
                 use_fx_bus=False, use_master_bus=True, use_output_fader=True)

    lo, hi = shard_batch(BS * world, rank, world)  # this rank's mixes of the global batch
    torch.manual_seed(1000 + rank)
    tracks = (0.1 * torch.randn(hi - lo, T, N)).to(dev)
    ref = naive_random_mix(tracks, console, **{k: v for k, v in flags.items() if k != "use_output_fader"})[1]
    ref = batch_stereo_peak_normalize(ref)  # reference mst/system.py:149-176
    track_params = torch.rand(hi - lo, T, 27).to(dev).requires_grad_(True)
    fx_params = torch.rand(hi - lo, 25).to(dev)
    master_params = torch.rand(hi - lo, 26).to(dev).requires_grad_(True)

    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("s", "fwd", "loss", "lbwd", "e")}

    def step(stamp=False):
        track_params.grad = None
        master_params.grad = None
        if stamp:
            ev["s"].record()
        _, mix, *_ = console(tracks, track_params, fx_params, master_params, **flags)
        if stamp:
            ev["fwd"].record()
            mix.register_hook(lambda g: ev["lbwd"].record())
        loss = loss_fn(mix, ref)
        if stamp:
            ev["loss"].record()
        loss.backward()
        out = reduce_loss(loss.detach(), world)
        if stamp:
            ev["e"].record()
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        last = step()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1)
    console.check_parameters()
    assert torch.isfinite(last).all() and torch.isfinite(track_params.grad).all()

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # stage split of one step on the launch stream (HIP events; kernels run on torch's current stream).
    # Two un-stamped steps are enqueued first so that the host is ahead of the GPU, as in the timed loop.
    step()
    step()
    step(stamp=True)
    torch.cuda.synchronize()
    stages = {"console_fwd_ms": ev["s"].elapsed_time(ev["fwd"]), "loss_fwd_ms": ev["fwd"].elapsed_time(ev["loss"]),
              "loss_bwd_ms": ev["loss"].elapsed_time(ev["lbwd"]), "console_bwd_ms": ev["lbwd"].elapsed_time(ev["e"])}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * BS * args.steps / elapsed
        gpu_ms_per_step = gpu_ms / args.steps
        achieved = BS * BYTES_PER_MIX / (gpu_ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "mixes/sec (8-track x 262144-sample fwd+bwd)",
            "value": value, "unit": "mixes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "BASELINE cfg #2: AdvancedMixConsole 8 tracks x 262144 @44.1kHz, batch 8 per GPU, "
                            "fwd+bwd + MR-STFT loss (512/2048/8192)",
                "per_gpu_batch": BS, "global_batch": BS * world, "tracks": T, "samples": N,
                "parallelism": f"batch-sharded x{world}, loss all-reduce only (RCCL)",
                "mixed_tracks": "not materialised (lean variant, SURVEY 8d)", "range_check": "deferred flag readback",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(),
                "kernel": "whole step = the ~33 back-to-back kernels of console fwd+bwd + MR-STFT fwd+bwd (no single kernel "
                          "exceeds 13 % of the step, see profiles/round1_summary.md); HIP-event time per step",
                "algorithmic_bytes_per_step": BS * BYTES_PER_MIX, "gpu_ms_per_step": gpu_ms_per_step, "stages": stages,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
