"""Developer timing of the cfg #5 step on one GPU (the secondary bench line), for use under rocprofv3: python tools/cfg5_step.py [iters=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
from mst.loss import AudioFeatureLoss
from mst.mixing import naive_random_mix
from mst.modules import AdvancedMixConsole, MixStyleTransferModel, SpectrogramEncoder, TransformerController
from mst.system import CommonStep

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
torch.manual_seed(3001)
model = MixStyleTransferModel(SpectrogramEncoder(embed_dim=512), SpectrogramEncoder(embed_dim=512),
                              TransformerController(512, 27, 25, 26, num_layers=12, nhead=8,
                                                    graphed=os.environ.get("MST_GRAPHED", "0") == "1",
                                                    native=os.environ.get("MST_NATIVE", "1") == "1")).to(dev).train()
step = CommonStep(model, AdvancedMixConsole(bench.SR, materialize_mixed_tracks=False, validate="deferred", param_dicts="lazy"), naive_random_mix,
                  AudioFeatureLoss(bench.AF_WEIGHTS, bench.SR), generate_mix=True, active_eq_epoch=0, active_compressor_epoch=0,
                  active_fx_bus_epoch=1000, active_master_bus_epoch=0, nan_check=os.environ.get("MST_NANCHECK", "deferred"))
tracks = (0.05 * torch.randn(1, 32, bench.N)).to(dev)
batch = (tracks, None, None, torch.zeros(1, 32, dtype=torch.bool, device=dev), None, ["a"])
for i in range(iters + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.zero_grad(set_to_none=True)
    loss, _ = step(batch, train=True)
    loss.backward()
    torch.cuda.synchronize()
    print(f"cfg5 step {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms")

def one():
    model.zero_grad(set_to_none=True)
    loss, _ = step(batch, train=True)
    loss.backward()

med, mean = bench.time_steps(one, max(iters, 5), 1)
print(f"cfg5 pipelined (no host sync between steps): {med:.2f} ms median, {mean:.2f} ms mean")
