"""``mst.mixing`` - mix functions on the hot path (reference mst/mixing.py:35-94)."""
import torch


def naive_random_mix(
    tracks: torch.Tensor,
    mix_console: torch.nn.Module,
    use_track_input_fader: bool = True,
    use_track_eq: bool = True,
    use_track_compressor: bool = True,
    use_track_panner: bool = True,
    use_fx_bus: bool = True,
    use_master_bus: bool = True,
    use_ouput_fader: bool = True,
    **kwargs,
):
    """Uniform-random parameters -> console under ``no_grad`` -> the reference's 8-tuple.

    The misspelt ``use_ouput_fader`` keyword and the ``**kwargs`` sink are part of the reference's
    interface (mixing.py:44; ``System`` passes ``use_output_fader`` which lands in kwargs, SURVEY
    App. C.2) and are kept.  Parameters are drawn with the CPU generator in the same order as the
    reference (mixing.py:61-69) so seeded runs agree, then moved to ``tracks``' device (asynchronously, from pinned memory).
    """
    bs, num_tracks, _ = tracks.size()
    def draw(*shape):
        # same CPU-generator stream as the reference; drawn into pinned memory and copied asynchronously: `.type_as` on a
        # pageable tensor is a copy + stream synchronize, i.e. three host stalls per mix with the GPU drained behind them
        if tracks.is_cuda:
            return torch.rand(*shape, pin_memory=True).to(tracks.device, non_blocking=True).type_as(tracks)
        return torch.rand(*shape).type_as(tracks)

    mix_params = draw(bs, num_tracks, mix_console.num_track_control_params)
    fx_bus_params = draw(bs, mix_console.num_fx_bus_control_params)
    master_bus_params = draw(bs, mix_console.num_master_bus_control_params)
    with torch.no_grad():
        mixed_tracks, mix, track_param_dict, fx_bus_param_dict, master_bus_param_dict = mix_console(
            tracks, mix_params, fx_bus_params, master_bus_params,
            use_track_input_fader=use_track_input_fader, use_track_eq=use_track_eq,
            use_track_compressor=use_track_compressor, use_track_panner=use_track_panner,
            use_master_bus=use_master_bus, use_fx_bus=use_fx_bus, use_output_fader=use_ouput_fader,
        )
    return (mixed_tracks, mix, track_param_dict, fx_bus_param_dict, master_bus_param_dict,
            mix_params, fx_bus_params, master_bus_params)
