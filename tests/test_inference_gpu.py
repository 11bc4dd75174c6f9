"""the inference driver ``run_diffmst`` (reference mst/utils.py:32-173) on the HIP console, against the
fixture written by the REAL function in the build container (tests/golden/make_golden.py run): loudness normalisation with a
dropped silent track, one parameter estimate, Hann-faded overlap-add over 262144-sample windows, a short last window."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import StubModel, rel, simple_lufs


def test_run_diffmst_against_the_real_function(golden_dir, record):
    from mst.modules import AdvancedMixConsole
    from mst.utils import run_diffmst

    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "run_diffmst.npz"))
    T, n = (int(v) for v in g["shape"])
    torch.manual_seed(int(g["seed_tracks"]))
    tracks = (0.05 * torch.randn(1, T, n) * torch.tensor([1.0, 0.3, 2.0, 1e-6, 0.7]).view(1, T, 1)).half().float()
    ref = 0.2 * torch.randn(1, 2, int(g["ref_len"]))
    assert np.array_equal(tracks.numpy()[..., ::4096], g["tracks_sub"]) and np.array_equal(ref.numpy()[..., ::4096], g["ref_sub"])
    model = StubModel(seed=int(g["seed_model"])).to(dev)
    before = tracks.clone()
    pred_mix, tpd, fpd, mpd = run_diffmst(tracks, ref, model, AdvancedMixConsole(44100), track_start_idx=int(g["track_start_idx"]),
                                          ref_start_idx=int(g["ref_start_idx"]), loudness_fn=simple_lufs)
    assert pred_mix.shape == (1, 2, n) and pred_mix.device == tracks.device and torch.equal(tracks, before)
    t = lambda k: torch.from_numpy(g[k])
    rep = dict(mix=rel(pred_mix[..., ::8], t("pred_mix_sub")), seam=rel(pred_mix[..., 262144 - 64:262144 + 64], t("seam")),
               l2=abs(pred_mix.double().pow(2).sum().sqrt().item() - float(g["pred_mix_l2"])) / float(g["pred_mix_l2"]))
    print("\n[run_diffmst vs the real function]", rep)
    record(**rep)
    assert rep["mix"] < 1e-4 and rep["seam"] < 1e-4 and rep["l2"] < 1e-5
    # four of the five tracks survive the -80 LUFS gate; the dictionaries are the console's (denormalised) ones
    assert tpd["compressor"]["ratio"].shape == (1, 4)
    assert torch.allclose(tpd["compressor"]["ratio"].cpu(), t("track_ratio"), rtol=1e-5)
    assert torch.allclose(mpd["compressor"]["threshold_db"].cpu(), t("master_thr"), rtol=1e-5)
    # device tensors in -> device tensor out, same numbers
    again, *_ = run_diffmst(tracks.to(dev), ref.to(dev), model, AdvancedMixConsole(44100), track_start_idx=int(g["track_start_idx"]),
                            ref_start_idx=int(g["ref_start_idx"]), loudness_fn=simple_lufs)
    assert again.is_cuda and torch.equal(again.cpu(), pred_mix)


def test_run_diffmst_short_song_and_missing_meter():
    from mst.modules import AdvancedMixConsole
    from mst.utils import run_diffmst

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    tracks, ref = 0.05 * torch.randn(1, 3, 70000), 0.1 * torch.randn(1, 2, 50000)  # both shorter than the analysis window
    model = StubModel(seed=1).to(dev)
    console = AdvancedMixConsole(44100)
    mix, tpd, _, _ = run_diffmst(tracks, ref, model, console, loudness_fn=simple_lufs)
    # one window, held at 1 over its first half and faded by the 262144-point Hann window after sample 131072: nothing fades
    g = torch.tensor([10 ** ((-48 - simple_lufs(tracks[0, t:t + 1].permute(1, 0).numpy())) / 20) for t in range(3)]).view(1, 3, 1)
    norm = (tracks * g).to(dev)
    tp, fp, mp = model(norm, ref.to(dev))
    with torch.no_grad():
        _, direct, *_ = console(norm, tp, fp, mp, use_fx_bus=False)
    assert torch.equal(mix, direct.cpu())
    try:
        import pyloudnorm  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="loudness_fn"):
            run_diffmst(tracks, ref, model, console)
