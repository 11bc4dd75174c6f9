import sys, os
ROOT = "/root/repo"
sys.path[:0] = [ROOT, ROOT + "/diff-mst_amd", ROOT + "/diff-mst_amd/standalone", ROOT + "/tests"]
import torch
from mst.modules import AdvancedMixConsole
from mst import _desc
from util import FULL, rel
dev = torch.device("cuda:0")
c = AdvancedMixConsole(44100)
torch.manual_seed(17)
bs, T, n = 2, 3, 65536
tracks = (0.1 * torch.randn(bs, T, n)).to(dev)
tp = torch.rand(bs, T, 27, device=dev); mp = torch.rand(bs, 26, device=dev); fp = torch.rand(bs, 25, device=dev)
ORDER = ("use_track_input_fader", "use_track_eq", "use_track_compressor", "use_track_panner", "use_fx_bus", "use_master_bus", "use_output_fader")
for name, flags in (("full", FULL), ("no eq/comp/master", dict(FULL, use_track_eq=False, use_track_compressor=False, use_master_bus=False)),
                    ("eq only", dict(FULL, use_track_compressor=False, use_master_bus=False)), ("comp only", dict(FULL, use_track_eq=False, use_master_bus=False)),
                    ("master only", dict(FULL, use_track_eq=False, use_track_compressor=False))):
    with torch.no_grad():
        mixed, mix, tpd, fpd, mpd = c(tracks, tp, fp, mp, **flags)
        mixed2, mix2 = c.forward_mix_console(tracks, tpd, fpd, mpd, *[flags[k] for k in ORDER])
    print(name, rel(mix2, mix), float((mix2 - mix).abs().max()))
lo, hi = _desc.range_vectors(c.param_ranges, _desc.TRACK_INDEX)
lo_t, hi_t = torch.tensor(lo, device=dev), torch.tensor(hi, device=dev)
a = tp * (hi_t - lo_t) + lo_t
b = torch.addcmul(lo_t, tp, hi_t - lo_t)
print("mul+add vs addcmul differ in", int((a != b).sum()), "of", a.numel())
