"""MI355X-native drop-in for the Diff-MST mix-console hot path.

Same import surface as the reference package for this path
(``mst.modules.AdvancedMixConsole``, ``mst.mixing.naive_random_mix``,
``mst.loss.AudioFeatureLoss`` / ``MultiResolutionSTFTLoss``,
``mst.utils.batch_stereo_peak_normalize``) - see INTEGRATION.md.  All numerical work
runs in hand-written HIP kernels for gfx950 behind the C ABI of
``include/diffmst_hip.h``; there is no CPU fallback.
"""
__version__ = "0.1.0"
