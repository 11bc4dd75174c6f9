"""``diffmst_hip.system`` - the body of the reference's training / validation step as a plain ``nn.Module``.

``CommonStep`` restates the CALL ORDER of ``System.common_step`` (reference mst/system.py:102-407) without
PyTorch-Lightning: epoch-gated effect flags (:123-133), the random reference mix under ``no_grad`` (:149-173,
executed twice by the reference, :222-246), peak normalisation and the NaN guard (:176-180 / :249-253), the A/B
split on the LAST dimension (:255-258 - ``tracks_b`` is a strided view, which the console kernels read in place),
the parameter-estimation model (:263-271), the console with gradients (:274-292) and the loss sum (:329-338).
Everything numerical inside it is the HIP console / losses of this package; the model is whatever the caller
passes (the reference's ``MixStyleTransferModel`` or any module with the same call signature).

It exists to prove "API unchanged" end to end on a machine without the reference (tests/test_system_gpu.py);
with a checkout of the reference, ``diffmst_hip.install()`` lets the reference's own ``System`` run instead.
Logging (``self.log``), wandb callbacks, optimisers and the D2H plotting copies are Lightning's business and stay
out (``collect=True`` reproduces the ``data_dict`` of :392-405 when a caller wants it).
"""
from __future__ import annotations

from typing import Callable

import torch

from .utils import batch_stereo_peak_normalize


class CommonStep(torch.nn.Module):
    def __init__(
        self,
        model: torch.nn.Module,
        mix_console: torch.nn.Module,
        mix_fn: Callable,
        loss: torch.nn.Module,
        generate_mix: bool = True,
        use_mix_loss: bool = True,
        active_eq_epoch: int = 0,
        active_compressor_epoch: int = 0,
        active_fx_bus_epoch: int = 0,
        active_master_bus_epoch: int = 0,
        max_epochs: int = 500,
        repeat_reference_mix: bool = True,
        nan_check: str = "sync",
    ) -> None:
        super().__init__()
        if nan_check not in ("sync", "deferred"):
            raise ValueError("nan_check is 'sync' or 'deferred'")
        # "sync" = the reference's `if torch.isnan(ref_mix).any(): raise` (:178-180): one device -> host readback in the middle
        # of every step, after which the GPU idles until the host has issued the encoder's first kernels.  "deferred" keeps the
        # flag on the device: `check_finite()` - called at the top of the NEXT step, when the flag has long been written -
        # raises the same ValueError one step later and the host never waits (cfg #5: DESIGN 9.5).  The reference aborts BEFORE
        # backward; with "deferred" the caller owns that guarantee: call `check_finite()` before `optimizer.step()` (it costs the
        # readback then, but after the backward has been queued) and once more when the loop ends, otherwise the flag of the last
        # step is never looked at and an optimizer step on a NaN batch goes through (bench.py and tools/cfg5_step.py do both).
        self.nan_check = nan_check
        self._nan_flag = None
        self.model = model
        self.mix_console = mix_console
        self.mix_fn = mix_fn
        self.loss = loss
        self.generate_mix = generate_mix
        self.use_mix_loss = use_mix_loss
        self.active_eq_epoch = active_eq_epoch
        self.active_compressor_epoch = active_compressor_epoch
        self.active_fx_bus_epoch = active_fx_bus_epoch
        self.active_master_bus_epoch = active_master_bus_epoch
        # the reference runs the mix_fn block twice per step and keeps the second result (a merge artefact, SURVEY
        # App. C.1); True reproduces its RNG stream and its cost, False runs the block once
        self.repeat_reference_mix = repeat_reference_mix
        self.current_epoch = 0
        # defaults of reference :83-89
        self.use_track_input_fader = True
        self.use_track_panner = True
        self.use_track_eq = False
        self.use_track_compressor = False
        self.use_fx_bus = False
        self.use_master_bus = False
        self.use_output_fader = True
        if active_fx_bus_epoch < max_epochs and not getattr(mix_console, "supports_fx_bus", False):
            # fail at construction, not at epoch `active_fx_bus_epoch` in the middle of a run
            raise NotImplementedError(
                f"active_fx_bus_epoch={active_fx_bus_epoch} switches the fx bus on within max_epochs={max_epochs}, but this "
                "console has no fx bus (no `supports_fx_bus` attribute): set active_fx_bus_epoch >= max_epochs (the reference's "
                "configs use 1000)"
            )

    def _reference_mix(self, tracks, instrument_id, stereo_info):
        return self.mix_fn(
            tracks,
            self.mix_console,
            use_track_input_fader=False,
            use_track_panner=self.use_track_panner,
            use_track_eq=self.use_track_eq,
            use_track_compressor=self.use_track_compressor,
            use_fx_bus=self.use_fx_bus,
            use_master_bus=self.use_master_bus,
            use_output_fader=False,  # lands in naive_random_mix's **kwargs (its keyword is misspelt), as in the reference
            instrument_id=instrument_id,
            stereo_id=stereo_info,
            instrument_number_file=None,
            ke_dict=None,
        )

    def check_finite(self):
        """Deferred NaN guard: raise the reference's ValueError if a reference mix of an earlier step held a NaN.  Runs at the top
        of every forward(); with nan_check="deferred" the training loop calls it again before ``optimizer.step()`` and at loop end."""
        flag, self._nan_flag = self._nan_flag, None
        if flag is not None and bool(flag):
            raise ValueError("Found nan in ref_mix")
        # the same (already synchronous) point reads the console's sticky status word: an in-launch exchange that timed out in a LATER
        # launch of an earlier forward, or in its backward, is reported here at the latest - validate="sync" only waits for the range
        # check's verdict, 20 us into the call (diffmst_hip/modules.py: _note_status)
        if flag is not None and hasattr(self.mix_console, "check_parameters"):
            self.mix_console.check_parameters()

    def forward(self, batch: tuple, train: bool = False, collect: bool = False):
        tracks, instrument_id, stereo_info, track_padding, ref_mix, song_name = batch
        self.check_finite()
        middle_idx = tracks.shape[-1] // 2
        if self.current_epoch >= self.active_eq_epoch:
            self.use_track_eq = True
        if self.current_epoch >= self.active_compressor_epoch:
            self.use_track_compressor = True
        if self.current_epoch >= self.active_fx_bus_epoch:
            self.use_fx_bus = True
        if self.current_epoch >= self.active_master_bus_epoch:
            self.use_master_bus = True

        ref_track_param_dict = ref_fx_bus_param_dict = ref_master_bus_param_dict = None
        if self.generate_mix:
            for _ in range(2 if self.repeat_reference_mix else 1):
                (_, ref_mix, ref_track_param_dict, ref_fx_bus_param_dict, ref_master_bus_param_dict,
                 _, _, _) = self._reference_mix(tracks, instrument_id, stereo_info)
                ref_mix = batch_stereo_peak_normalize(ref_mix)
                found = torch.isnan(ref_mix).any()
                if self.nan_check == "deferred":
                    self._nan_flag = found if self._nan_flag is None else self._nan_flag | found
                elif found:
                    raise ValueError("Found nan in ref_mix")
            ref_mix_a = ref_mix[..., :middle_idx]
            ref_mix_b = ref_mix[..., middle_idx:]
            tracks_b = tracks[..., middle_idx:]
        else:
            ref_mix_a = ref_mix_b = ref_mix
            tracks_b = tracks

        pred_track_params, pred_fx_bus_params, pred_master_bus_params = self.model(
            tracks_b, ref_mix_a, track_padding_mask=track_padding
        )
        (pred_mixed_tracks_b, pred_mix_b, pred_track_param_dict, pred_fx_bus_param_dict,
         pred_master_bus_param_dict) = self.mix_console(
            tracks_b,
            pred_track_params,
            pred_fx_bus_params,
            pred_master_bus_params,
            use_track_input_fader=self.use_track_input_fader,
            use_track_panner=self.use_track_panner,
            use_track_eq=self.use_track_eq,
            use_track_compressor=self.use_track_compressor,
            use_fx_bus=self.use_fx_bus,
            use_master_bus=self.use_master_bus,
            use_output_fader=self.use_output_fader,
        )
        if ref_track_param_dict is None:
            ref_track_param_dict = pred_track_param_dict
            ref_fx_bus_param_dict = pred_fx_bus_param_dict
            ref_master_bus_param_dict = pred_master_bus_param_dict

        loss = 0
        terms = {}
        if self.use_mix_loss:
            mix_loss = self.loss(pred_mix_b, ref_mix_b)
            if type(mix_loss) == dict:
                # reference: `loss += val.mean()` from `loss = 0`.  The mean of a 0-dim value IS the value and 0 + v IS v, to the bit: for
                # the five 0-dim terms of AudioFeatureLoss that is 6 tiny launches forward and 5 backward less per step (cfg #5 is
                # launch-bound at the bf16 precision: profiles/round5_cfg5.md)
                for key, val in mix_loss.items():
                    term = val if val.dim() == 0 else val.mean()
                    loss = term if isinstance(loss, int) else loss + term
                terms = mix_loss
            else:
                loss += mix_loss

        data_dict = {"loss_terms": terms}
        if collect:  # the plotting payload of reference :388-405 (device -> host copies every step)
            sum_mix_b = batch_stereo_peak_normalize(tracks_b.sum(dim=1, keepdim=True).detach().float())
            data_dict.update(
                ref_mix_a=ref_mix_a.detach().float().cpu(),
                ref_mix_b_norm=ref_mix_b.detach().float().cpu(),
                pred_mix_b_norm=pred_mix_b.detach().float().cpu(),
                sum_mix_b=sum_mix_b.cpu(),
                ref_track_param_dict=ref_track_param_dict,
                pred_track_param_dict=pred_track_param_dict,
                ref_fx_bus_param_dict=ref_fx_bus_param_dict,
                pred_fx_bus_param_dict=pred_fx_bus_param_dict,
                ref_master_bus_param_dict=ref_master_bus_param_dict,
                pred_master_bus_param_dict=pred_master_bus_param_dict,
            )
        return loss, data_dict
