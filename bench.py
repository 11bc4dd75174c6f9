#!/usr/bin/env python3
"""Headline benchmark of the MI355X-native Diff-MST mix-console hot path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one synthetic batch per GPU (BASELINE config #2):
AdvancedMixConsole forward on 8 mixes x 8 tracks x 262144 samples @ 44.1 kHz, MR-STFT loss
(512/2048/8192) against a peak-normalised random reference mix, backward to the 27/26 parameter
tensors.  Inputs are resident in HBM before the timed region.  N > 1 shards the batch axis (weak
scaling: 8 mixes per GPU) and all-reduces the loss scalar only (leaf parameters stay local) - RCCL.

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline      HBM roofline of the whole step on ALGORITHMIC bytes + a VALU cross-check
  cpu_baseline  the oracle (CPU restatement of the reference algorithm) timed on this box's host cores
  secondary     (N = 1 only, --no-secondary skips) the API-faithful variant of cfg #2, cfg #3 and cfg #1
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

BS, T, N, SR = 8, 8, 262144, 44100
RESOLUTIONS = dict(fft_sizes=[512, 2048, 8192], hop_sizes=[256, 1024, 4096], win_lengths=[512, 2048, 8192])
AF_WEIGHTS = [0.1, 0.001, 1.0, 1.0, 0.1]  # reference configs/models/unpaired+feat.yaml:55-60
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 vector peak
FLAGS = dict(use_track_input_fader=True, use_track_eq=True, use_track_compressor=True, use_track_panner=True,
             use_fx_bus=False, use_master_bus=True, use_output_fader=True)


def bytes_per_mix(n_tracks, n, materialised=False, loss=True):
    """Algorithmic bytes per mix, SURVEY 8(d): console fwd+bwd (lean) 8*N*(T+2) [+ 8*N*T when mixed_tracks is
    written] + loss fwd+bwd 24*N (MR-STFT and AudioFeatureLoss alike)."""
    return 8 * n * (n_tracks + 2) + (8 * n * n_tracks if materialised else 0) + (24 * n if loss else 0)


BYTES_PER_MIX = bytes_per_mix(T, N)
# algorithmic flops per mix (SURVEY 8d estimate: console ~0.63 GF at T = 8, MR-STFT ~0.26 GF) - for the VALU cross-check
FLOPS_PER_MIX = 0.63e9 + 0.26e9


def committed_profile():
    """Numbers that need the profiler (bench.py cannot run rocprofv3 on itself): HBM bytes per step from the PMC passes
    and VALU instructions per step from the SQ passes of this same command, committed under profiles/."""
    for name in ("round6_traffic.json", "round5_traffic.json", "round4_traffic.json", "round3_traffic.json", "round2_traffic.json", "round1_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            # round 4 on: corrected bytes (2 x FETCH_SIZE + WRITE_SIZE, profiles/round4_hbm_calibration.md); earlier files hold the raw sum
            return float(d.get("hbm_bytes_per_step", d["hbm_bytes_per_step_raw"])), d.get("valu_lane_instructions_per_step"), name
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous batch-axis shard owned by `rank`: mixes [lo, hi)."""
    per = global_batch // world
    assert per * world == global_batch, "global batch must divide evenly"
    return rank * per, (rank + 1) * per


def reduce_loss(loss: torch.Tensor, world: int):
    """Mean of the per-rank losses == single-process loss over the global batch (per-example terms).  Blocking form (tests)."""
    if world > 1:
        dist.all_reduce(loss, op=dist.ReduceOp.SUM)
        loss = loss / world
    return loss


class AsyncLossReduce:
    """The loss scalar is logging output: nothing on the device waits for its all-reduce.  Each step's value is copied into a slot of a
    resident ring on the compute stream; a side stream waits for that copy, all-reduces the slot (RCCL, async_op) and is joined ONCE,
    before the line is printed - the compute stream never waits for the collective ('async, off the critical path')."""

    def __init__(self, world: int, dev, slots: int):
        self.world, self.dev = world, dev
        self.ring = torch.zeros(max(slots, 1), device=dev)
        self.side = torch.cuda.Stream(device=dev) if world > 1 else None
        self.i = 0
        self.works = []

    def push(self, loss: torch.Tensor):
        if self.world == 1:  # nothing to reduce: no copy, no launch
            self.last = loss.detach()
            return self.last
        slot = self.ring[self.i % self.ring.numel()].view(())
        self.i += 1
        slot.copy_(loss.detach())
        if self.world > 1:
            self.side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self.side):
                self.works.append(dist.all_reduce(slot, op=dist.ReduceOp.SUM, async_op=True))
        return slot

    def join(self):
        """Wait for every queued all-reduce (host + compute stream) and return the mean loss of the last step."""
        if self.world == 1:
            return self.last
        for w in self.works:
            w.wait()
        self.works.clear()
        if self.side is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
        last = self.ring[(self.i - 1) % self.ring.numel()]
        return last / self.world


# ---------------------------------------------------------------------------------------------- CPU baseline
def _oracle_step(n_tracks, n, loss_kind, flags):
    from oracle import console_restated as oc
    from oracle import loss_restated as ol

    torch.manual_seed(0)
    tracks = 0.1 * torch.randn(1, n_tracks, n)
    fp = torch.rand(1, 25)
    with torch.no_grad():
        _, ref, *_ = oc.console_forward(tracks, torch.rand(1, n_tracks, 27), fp, torch.rand(1, 26), **flags)
        ref = oc.batch_stereo_peak_normalize(ref)
    res = tuple(zip(RESOLUTIONS["fft_sizes"], RESOLUTIONS["hop_sizes"], RESOLUTIONS["win_lengths"]))

    def one():
        tp = torch.rand(1, n_tracks, 27, requires_grad=True)
        mp = torch.rand(1, 26, requires_grad=True)
        _, mix, *_ = oc.console_forward(tracks, tp, fp, mp, **flags)
        if loss_kind == "mrstft":
            ol.mrstft_loss(mix, ref, res).backward()
        elif loss_kind == "af":
            sum(v.mean() for v in ol.audio_feature_loss(mix, ref, AF_WEIGHTS).values()).backward()
        else:
            (mix * ref).sum().backward()

    return one


def _time_cpu(one, runs, budget_s, warmups=2, min_runs=3):
    """2 warm-ups, median of 5 timed iterations - fewer only when the time budget runs out (an over-subscribed
    thread setting can take minutes per iteration: then the warm-ups themselves are the sample)."""
    t_start, warm = time.time(), []
    for _ in range(warmups):
        t0 = time.time()
        one()
        warm.append(time.time() - t0)
        if time.time() - t_start > budget_s:
            return warm[-1], 1
    times, t_start = [], time.time()
    while len(times) < runs and (time.time() - t_start < budget_s or len(times) < min_runs):
        t0 = time.time()
        one()
        times.append(time.time() - t0)
    return statistics.median(times), len(times)


def cpu_baseline():
    """Oracle (PyTorch-CPU restatement of the reference ALGORITHM: frequency-sampling IIR via torch.fft, torch.stft
    losses, autograd backward) on bounded samples of the bench workloads.  Three thread settings are timed on cfg #2 -
    every host cpu, 16 and 1 (more threads are NOT faster for these batched 2^19-point FFTs: the measured optimum on the
    256-cpu EPYC box is a handful of threads) - and `value` is the best of them, with `cores` = the threads it used."""
    host = os.cpu_count()
    try:
        model = [l.split(":", 1)[1].strip() for l in subprocess.run(["lscpu"], capture_output=True, text=True).stdout.splitlines()
                 if l.startswith("Model name")][0]
    except Exception:
        model = "unknown"
    default_threads = torch.get_num_threads()
    legs = {}
    # all cores AND one thread; 16 threads is the measured optimum of these batched 2^19-point FFTs on the EPYC
    # hosts and is timed as well
    # "all cores" = torch's own default thread count = the physical cores (128 on the 256-cpu EPYC hosts; the other 128 cpus are their SMT
    # siblings, and on all 256 these 2^19-point FFT batches over-subscribe to 0.01 mixes/s = 100 s per run, measured in round 5)
    allc = default_threads
    for threads in (allc, 16, 1):
        if threads > host or threads in legs:
            continue
        torch.set_num_threads(threads)
        med, k = _time_cpu(_oracle_step(T, N, "mrstft", FLAGS), 5, 30.0 if threads == allc else 15.0, min_runs=1 if threads == allc else 3)
        legs[threads] = (1.0 / med, k)
    best = max(legs, key=lambda t: legs[t][0])
    torch.set_num_threads(best)
    # cfg #3 (16 tracks, AudioFeatureLoss) and cfg #1 (gain + pan only, 4 x 65536), one mix each, at the best thread count
    med3, _ = _time_cpu(_oracle_step(16, N, "af", FLAGS), 3, 12.0, warmups=1)
    basic = dict(FLAGS, use_track_eq=False, use_track_compressor=False, use_master_bus=False, use_output_fader=False)
    med1, _ = _time_cpu(_oracle_step(4, 65536, "none", basic), 5, 3.0, warmups=1)
    torch.set_num_threads(default_threads)
    return {
        "value": legs[best][0], "unit": "mixes/s", "cores": best, "kind": "port",
        "sample": f"cfg #2: 1 mix (8 tracks x 262144) console fwd+bwd + MR-STFT, fp32, median of {legs[best][1]} runs after 2 warm-ups, "
                  f"best of the thread settings below on {host} host cpus ({model})",
        "by_threads": {str(t): {"value": v, "unit": "mixes/s", "runs": k} for t, (v, k) in legs.items()},
        "all_cores": {"value": legs[allc][0], "unit": "mixes/s", "cores": allc, "runs": legs[allc][1],
                      "note": "torch's default thread count = the physical cores of the host (BASELINE.md section 3 'all cores'; os.cpu_count() "
                              "counts their SMT siblings too, where this workload over-subscribes to 0.01 mixes/s)"},
        "cfg3": {"value": 1.0 / med3, "unit": "mixes/s", "cores": best,
                 "sample": "1 mix of 16 tracks x 262144, console fwd+bwd + AudioFeatureLoss"},
        "cfg1": {"value": 1.0 / med1, "unit": "mixes/s", "cores": best,
                 "sample": "1 mix of 4 tracks x 65536, gain + pan + bus sum fwd+bwd"},
        "host": {"os_cpu_count": host, "lscpu_model": model, "torch_default_threads": default_threads,
                 "parallel_info": " | ".join(l.strip() for l in torch.__config__.parallel_info().splitlines() if l.strip())[:400]},
    }


# ---------------------------------------------------------------------------------------------- GPU workloads
def make_workload(dev, bs, n_tracks, n, loss_kind, seed, lean=True, flags=FLAGS, basic=False):
    """Returns step() -> loss for one configuration; every tensor is resident on `dev` before it is called."""
    from mst.loss import AudioFeatureLoss, MultiResolutionSTFTLoss
    from mst.mixing import naive_random_mix
    from mst.modules import AdvancedMixConsole, BasicMixConsole
    from mst.utils import batch_stereo_peak_normalize

    kw = dict(materialize_mixed_tracks=False, validate="deferred", param_dicts="lazy") if lean else {}
    console = (BasicMixConsole if basic else AdvancedMixConsole)(SR, **kw)
    torch.manual_seed(seed)
    tracks = (0.1 * torch.randn(bs, n_tracks, n)).to(dev)
    if basic:
        with torch.no_grad():
            ref = console(tracks, torch.rand(bs, n_tracks, 27).to(dev))[1]
    else:
        ref = naive_random_mix(tracks, AdvancedMixConsole(SR, materialize_mixed_tracks=False),
                               **{k: v for k, v in flags.items() if k != "use_output_fader"})[1]
    ref = batch_stereo_peak_normalize(ref)  # reference mst/system.py:149-176
    track_params = torch.rand(bs, n_tracks, 27).to(dev).requires_grad_(True)
    fx_params = torch.rand(bs, 25).to(dev)
    if flags.get("use_fx_bus"):
        fx_params.requires_grad_(True)
    master_params = torch.rand(bs, 26).to(dev).requires_grad_(True)
    if loss_kind == "mrstft":
        loss_fn = MultiResolutionSTFTLoss(**RESOLUTIONS)
    elif loss_kind == "af":
        af = AudioFeatureLoss(AF_WEIGHTS, SR)
        loss_fn = lambda a, b: sum(v.mean() for v in af(a, b).values())  # reference mst/system.py:334-336
    else:
        loss_fn = lambda a, b: (a * b).mean()

    seed_grad = torch.ones((), device=dev)  # dL/dL, resident: `loss.backward()` would launch a ones-fill for it every step

    def step(marks=None):
        track_params.grad = None
        master_params.grad = None
        fx_params.grad = None
        if basic:
            _, mix, *_ = console(tracks, track_params)
        else:
            _, mix, *_ = console(tracks, track_params, fx_params, master_params, **flags)
        if marks:
            marks["fwd"].record()
            mix.register_hook(lambda g: marks["lbwd"].record())
        loss = loss_fn(mix, ref)
        if marks:
            marks["loss"].record()
        torch.autograd.backward(loss, grad_tensors=seed_grad.expand_as(loss))
        return loss.detach()

    step.console = console
    step.params = (track_params, master_params)
    step.inputs = dict(tracks=tracks, ref=ref, fx_params=fx_params)  # tests/test_cfg2_step_gpu.py drives the oracle with the same tensors
    return step


def time_steps(step, steps, warmup):
    """Median / mean GPU time per step from per-step HIP events on torch's current stream (the launch stream)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    return statistics.median(per), sum(per) / steps


def graph_replay(step, bs, steps, warmup):
    """The same eager step captured once (torch.cuda.CUDAGraph = hipGraph) and replayed: the device-side cost of the step without the
    host's enqueue work.  Replay must be bit-equal to eager (loss and the first parameter gradient).  Never the headline: `value` is eager."""
    try:
        ref_loss = step().clone()
        ref_grad = step.params[0].grad.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_loss = step()
        torch.cuda.synchronize()
        gmed, gmean = time_steps(graph.replay, steps, warmup)
        torch.cuda.synchronize()
        same = bool(torch.equal(g_loss, ref_loss) and torch.equal(step.params[0].grad, ref_grad))
        assert same, "hipGraph replay differs from the eager step"
        return {"ms_per_step_median": gmed, "ms_per_step_mean": gmean, "steps": steps, "mixes_per_s": bs / (gmed * 1e-3),
                "note": "the same step replayed as one hipGraph (torch.cuda.CUDAGraph), bit-equal to eager; device-side time only - the "
                        "headline value is the eager loop"}
    except Exception as e:  # noqa: BLE001 - a capture failure must not take the contract line down
        torch.cuda.synchronize()
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def conv_flops(frames, bins):
    """Forward multiply-add flops of Cnn14's twelve 3x3 convolutions for ONE signal (reference mst/panns.py:135-198): 124.5 GF
    at 513 x 1025 (262144 samples, STFT 2048 / 512)."""
    ch = (1, 64, 128, 256, 512, 1024, 2048)
    pools = ((2, 2), (4, 4), (2, 4), (2, 4), (2, 4), (2, 2))  # (frames, bins)
    h, w, f = frames, bins, 0
    for b in range(6):
        f += 2 * 9 * h * w * (ch[b] * ch[b + 1] + ch[b + 1] * ch[b + 1])
        h, w = h // pools[b][0], w // pools[b][1]
    return f


MFMA_PEAK_BF16_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak


def encoder_lines(dev):
    """the spectrogram encoder (STFT front end + Cnn14 on MFMA), forward + backward, and the cfg #5 step - each at the
    reference's precision (fp32 operands on the fp32 MFMA, configs/config.yaml:42 `precision: 32`) and with bf16 operand storage."""
    from mst.modules import SpectrogramEncoder

    out = []
    ns = 34  # the 32 tracks + 2 reference-mix channels one cfg #5 mix sends through the encoders
    fl = 3.0 * ns * conv_flops(1 + N // 512, 1025)
    for precision in ("fp32", "bf16x6", "bf16x3", "bf16"):
        torch.manual_seed(3000)
        enc = SpectrogramEncoder(embed_dim=512, precision=precision).to(dev).train()
        x = (0.1 * torch.randn(ns, 1, N)).to(dev)
        g = torch.randn(ns, 512, device=dev)

        def enc_step():
            enc.zero_grad(set_to_none=True)
            enc(x).backward(g)

        med, mean = time_steps(enc_step, 5, 2)
        # fp32 MFMA runs at the fp32 vector rate (MI355X_MICROARCH.md); bf16x3 issues three bf16 MFMAs per block: useful flops against a third of the bf16 peak
        peak = {"bf16": MFMA_PEAK_BF16_TF, "bf16x3": MFMA_PEAK_BF16_TF / 3, "bf16x6": MFMA_PEAK_BF16_TF / 6}.get(precision, VALU_PEAK_TF)
        out.append({"workload": f"SpectrogramEncoder + Cnn14 (embed 512) fwd+bwd, {ns} signals x {N} samples, "
                                + ("fp32 operands on v_mfma_f32_16x16x4_f32 = the reference's precision" if precision == "fp32" else
                                   "bf16x6: fp32 tensors, convolution operands split into bf16 (hi, mid, lo) triples in registers, the six product terms "
                                   "down to 2^-24 on v_mfma_f32_16x16x32_bf16 with fp32 accumulation - passes the fp32 path's parity bounds "
                                   "(tests/test_encoder_gpu.py, test_cfg5_step_as_benchmarked[bf16x6])" if precision == "bf16x6" else
                                   "bf16x3: fp32 tensors, operands split into bf16 (hi, lo) pairs, hi hi + hi lo + lo hi: ~2^-17 per product - between "
                                   "fp32 and bf16 (eval-mode weight gradients 3e-3 from float64; fp32 2e-4, bf16 6e-2)" if precision == "bf16x3" else
                                   "bf16 operand storage / fp32 accumulate (NOT the reference's arithmetic: training-mode weight gradients 20-40 % "
                                   "from fp32, DESIGN 9.3)") + " (reference mst/modules.py:740-806, mst/panns.py:126-209)",
                    "precision": precision, "ms_per_step_median": med, "ms_per_step_mean": mean, "steps": 5, "signals_per_s": ns / (med * 1e-3),
                    "conv_TFLOPs_per_s": fl / (med * 1e-3) / 1e12, "frac_of_dense_mfma_peak_of_that_dtype": fl / (med * 1e-3) / 1e12 / peak,
                    "note": "flops = 3 x forward multiply-adds x 2 of the twelve convolutions; time = whole encoder step incl. STFT, BatchNorm, "
                            "pooling, weight re-layout"})
        del enc, x, g
        torch.cuda.empty_cache()
    # cfg #5 on one GPU, one mix per step: full step with the real model structure (configs/models/naive+feat.yaml sizes); parity of this
    # exact configuration: tests/test_system_gpu.py::test_cfg5_step_as_benchmarked
    for precision in ("fp32", "bf16x6", "bf16x3", "bf16"):
        model, step = build_cfg5(dev, precision)
        torch.manual_seed(3002)
        tracks = (0.05 * torch.randn(1, 32, N)).to(dev)
        batch = (tracks, None, None, torch.zeros(1, 32, dtype=torch.bool, device=dev), None, ["a"])

        def sys_step():
            model.zero_grad(set_to_none=True)
            loss, _ = step(batch, train=True)
            loss.backward()

        med, mean = time_steps(sys_step, 5, 2)
        step.check_finite()
        out.append({"workload": "cfg #5 step on ONE GPU, batch 1: System.common_step order (two naive_random_mix reference mixes, peak normalise, "
                                "A/B split) + MixStyleTransferModel (2 x SpectrogramEncoder/Cnn14 on MFMA for 32 tracks + 2 mix channels of 131072 "
                                "samples, 12-layer TransformerController on csrc/mst_ctrl.hip) + AdvancedMixConsole 32 tracks + AudioFeatureLoss, fwd+bwd "
                                "to every weight; no host readback inside the step (deferred range / NaN checks, checked after the loop)",
                    "encoder_precision": precision, "ms_per_step_median": med, "ms_per_step_mean": mean, "steps": 5, "mixes_per_s": 1.0 / (med * 1e-3)})
        del model, step, tracks
        torch.cuda.empty_cache()
    return out


def build_cfg5(dev, precision, world=1):
    """The cfg #5 step exactly as tests/test_system_gpu.py::test_cfg5_step_as_benchmarked checks it against the oracle."""
    from mst.loss import AudioFeatureLoss
    from mst.mixing import naive_random_mix
    from mst.modules import AdvancedMixConsole, MixStyleTransferModel, SpectrogramEncoder, TransformerController
    from mst.system import CommonStep

    torch.manual_seed(3001)
    model = MixStyleTransferModel(SpectrogramEncoder(embed_dim=512, precision=precision), SpectrogramEncoder(embed_dim=512, precision=precision),
                                  TransformerController(512, 27, 25, 26, num_layers=12, nhead=8, native=True)).to(dev).train()
    wrapped = model
    if world > 1:  # reference configs/config.yaml:40-41: strategy ddp_find_unused_parameters_true, sync_batchnorm true
        from torch.nn.parallel import DistributedDataParallel

        wrapped = DistributedDataParallel(torch.nn.SyncBatchNorm.convert_sync_batchnorm(model), device_ids=[dev.index],
                                          find_unused_parameters=True)
    step = CommonStep(wrapped, AdvancedMixConsole(SR, materialize_mixed_tracks=False, validate="deferred", param_dicts="lazy"), naive_random_mix,
                      AudioFeatureLoss(AF_WEIGHTS, SR), generate_mix=True, active_eq_epoch=0, active_compressor_epoch=0,
                      active_fx_bus_epoch=1000, active_master_bus_epoch=0, nan_check="deferred")
    return model, step


def main_cfg5(args, dev, world, rank, backend):
    """`--config 5 --gpus N`: one 32-track mix per GPU; SyncBatchNorm + DDP over RCCL (weak scaling); same barrier / max-over-ranks timing."""
    model, step = build_cfg5(dev, args.precision, world)
    torch.manual_seed(4000 + rank)
    tracks = (0.05 * torch.randn(1, 32, N)).to(dev)
    batch = (tracks, None, None, torch.zeros(1, 32, dtype=torch.bool, device=dev), None, ["a"])

    def one():
        model.zero_grad(set_to_none=True)
        loss, _ = step(batch, train=True)
        loss.backward()
        return loss.detach()

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = one()
    step.check_finite()  # deferred NaN guard: before an optimizer would step, and at loop end (diffmst_hip/system.py)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert torch.isfinite(last).all()
    if rank == 0:
        print(json.dumps({
            "metric": "mixes/sec (full System step, 32-track x 262144-sample mix, fwd+bwd to every weight)", "value": world * args.steps / elapsed,
            "unit": "mixes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16x6": "f32 tensors, bf16x6 split operands / f32 accumulate", "bf16x3": "f32 tensors, bf16x3 split operands / f32 accumulate"}.get(args.precision, "bf16 operands / f32 accumulate"),
            "data": "synthetic",
            "config": {"workload": "BASELINE cfg #5: System.common_step order (two naive_random_mix reference mixes, peak normalise, A/B split) + "
                                   "MixStyleTransferModel (2 x SpectrogramEncoder/Cnn14 embed 512, 12-layer TransformerController on csrc/mst_ctrl.hip) + "
                                   "AdvancedMixConsole 32 tracks + AudioFeatureLoss; one mix per GPU",
                       "per_gpu_batch": 1, "global_batch": world, "tracks": 32, "samples": N, "encoder_precision": args.precision,
                       "parallelism": f"DistributedDataParallel(find_unused_parameters=True) + SyncBatchNorm x{world}; backend: {backend}"}}))
    if world > 1:
        dist.destroy_process_group()


def secondary_lines(dev):
    out = []
    specs = [
        ("cfg #2 API-faithful: mixed_tracks (bs,2,T,N) materialised, validate='sync' (one flag readback per call), eager "
         "parameter dictionaries", dict(bs=BS, n_tracks=T, n=N, loss_kind="mrstft", lean=False), True, 20, 5),
        ("cfg #2 with the fx bus on (the reference's DEFAULT call flags; its configs keep the bus off): send bus + 65536-tap "
         "noise-shaped reverberation, fresh noise drawn on the device every call, gradients to the 25 fx parameters too; lean console",
         dict(bs=BS, n_tracks=T, n=N, loss_kind="mrstft", lean=True, flags=dict(FLAGS, use_fx_bus=True)), False, 20, 5),
        ("cfg #3: AdvancedMixConsole 16 tracks x 262144, batch 32, AudioFeatureLoss, lean console",
         dict(bs=32, n_tracks=16, n=N, loss_kind="af", lean=True), False, 6, 2),
        ("cfg #1: BasicMixConsole (gain + pan + bus sum) 4 tracks x 65536, batch 2, fwd+bwd",
         dict(bs=2, n_tracks=4, n=65536, loss_kind="none", lean=True, basic=True), False, 50, 10),
    ]
    out.extend(encoder_lines(dev))
    for name, kw, materialised, steps, warm in specs:
        step = make_workload(dev, seed=2000, **kw)
        med, mean = time_steps(step, steps, warm)
        graphed = None
        if kw.get("basic"):
            # cfg #1 is three console launches (20 us of kernels, profiles/round4_cfg1.md) inside ~130 us of host work per eager step
            # (autograd bookkeeping, the loss's own torch kernels): the same step captured once and replayed as ONE hipGraph shows the
            # device-side cost.  Replay is bit-equal to eager (checked inside).
            graphed = graph_replay(step, kw["bs"], 200, 20)
        if kw.get("lean", True):
            step.console.check_parameters()
        b = kw["bs"] * bytes_per_mix(kw["n_tracks"], kw["n"], materialised, loss=kw["loss_kind"] != "none")
        gbs = b / (med * 1e-3) / 1e9
        out.append({"workload": name, "ms_per_step_median": med, "ms_per_step_mean": mean, "steps": steps,
                    "mixes_per_s": kw["bs"] / (med * 1e-3), "algorithmic_bytes_per_step": b, "achieved_GBs": gbs,
                    "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, **({"hipgraph_replay": graphed} if graphed else {})})
        del step
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replay of the headline step (reported beside it, never as `value`)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="2 = BASELINE cfg #2 (the headline; default).  5 = the full System step (cfg #5): one 32-track mix per GPU, model wrapped "
                         "in SyncBatchNorm + DistributedDataParallel(find_unused_parameters=True) when --gpus > 1")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16x6", "bf16x3", "bf16"), help="encoder operand precision of --config 5 (reference: fp32; bf16x6 / bf16x3: fp32 tensors, split operands)")
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
    backend = "single process"
    if world > 1:
        # one process per device: every rank must sit on its own GPU (torchrun sets LOCAL_RANK; a mis-launch would time 8 ranks on one GPU)
        backend = f"{dist.get_backend()} (RCCL), world_size {dist.get_world_size()}"
        ids = [None] * world
        dist.all_gather_object(ids, (os.uname().nodename, torch.cuda.current_device()))
        assert len(set(ids)) == world, f"ranks share a device: {ids}"
    if args.config == 5:
        return main_cfg5(args, dev, world, rank, backend)

    lo, hi = shard_batch(BS * world, rank, world)  # this rank's mixes of the global batch
    step_fn = make_workload(dev, hi - lo, T, N, "mrstft", seed=1000 + rank, lean=True)
    reducer = AsyncLossReduce(world, dev, args.steps + args.warmup + 8)

    def step(marks=None):
        return reducer.push(step_fn(marks))

    for _ in range(args.warmup):
        step()
    reducer.join()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    last = reducer.join()  # the queued loss all-reduces finish inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_step = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    gpu_ms = evs[0].elapsed_time(evs[-1])
    step_fn.console.check_parameters()
    assert torch.isfinite(last).all() and torch.isfinite(step_fn.params[0].grad).all()

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # stage split of one step on the launch stream (HIP events; kernels run on torch's current stream).
    # Two un-stamped steps are enqueued first so that the host is ahead of the GPU, as in the timed loop.
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("s", "fwd", "loss", "lbwd", "e")}
    step()
    step()
    ev["s"].record()
    step(ev)
    ev["e"].record()
    reducer.join()
    torch.cuda.synchronize()
    stages = {"console_fwd_ms": ev["s"].elapsed_time(ev["fwd"]), "loss_fwd_ms": ev["fwd"].elapsed_time(ev["loss"]),
              "loss_bwd_ms": ev["loss"].elapsed_time(ev["lbwd"]), "console_bwd_ms": ev["lbwd"].elapsed_time(ev["e"])}

    replay = None
    if world == 1 and not args.no_graph:
        replay = graph_replay(step_fn, BS, 50, 10)  # outside the timed region; device-only time of the same step (review item 6)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * BS * args.steps / elapsed
        gpu_ms_per_step = gpu_ms / args.steps
        med_ms = statistics.median(per_step)
        achieved = BS * BYTES_PER_MIX / (gpu_ms_per_step * 1e-3) / 1e9
        traffic, lane_insts, prof = committed_profile()
        tf = BS * FLOPS_PER_MIX / (gpu_ms_per_step * 1e-3) / 1e12
        out = {
            "metric": "mixes/sec (8-track x 262144-sample fwd+bwd)",
            "value": value, "unit": "mixes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "BASELINE cfg #2: AdvancedMixConsole 8 tracks x 262144 @44.1kHz, batch 8 per GPU, "
                            "fwd+bwd + MR-STFT loss (512/2048/8192)",
                "per_gpu_batch": BS, "global_batch": BS * world, "tracks": T, "samples": N,
                "parallelism": f"batch-sharded x{world}, loss all-reduce only (async, side stream); backend: {backend}",
                "mixed_tracks": "not materialised (lean variant, SURVEY 8d)", "range_check": "deferred flag readback",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "whole step = the back-to-back kernels of console fwd+bwd + MR-STFT fwd+bwd (per-kernel table under "
                          f"profiles/; traffic from profiles/{prof}); HIP-event time per step on the launch stream",
                "algorithmic_bytes_per_step": BS * BYTES_PER_MIX, "gpu_ms_per_step": gpu_ms_per_step,
                "gpu_ms_per_step_median": med_ms, "stages": stages,
                "valu": {
                    "flops_per_step": BS * FLOPS_PER_MIX, "achieved_TF": tf, "frac_of_157.3": tf / VALU_PEAK_TF,
                    "lane_instructions_per_step": lane_insts,
                    "issue_frac": (lane_insts / 64 * 2 / 1024 / 2.4e9) / (gpu_ms_per_step * 1e-3) if lane_insts else None,
                    "note": "flops = SURVEY 8d algorithmic estimate; lane_instructions = SQ_INSTS_VALU x 64 of the committed "
                            "counter pass; issue_frac = share of the step's SIMD-32 issue slots (2 cycles per wave64 op) they fill",
                },
            },
        }
        if replay is not None:
            out["hipgraph_replay"] = replay
        if world == 1 and not args.no_secondary:
            del step_fn
            torch.cuda.empty_cache()
            out["secondary"] = secondary_lines(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
