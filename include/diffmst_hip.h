/* diffmst_hip.h - C ABI of the MI355X-native Diff-MST mix-console hot path.
 *
 * The reference (sai-soum/Diff-MST) is pure Python and has no FFI; its seam for this
 * path is the operator ring of six functions imported from dasp_pytorch.functional
 * (reference mst/modules.py:7-14, called at :231-312) plus the loss objects called as
 * loss(pred, target) (reference mst/system.py:331-338).  This header is the C-ABI
 * ring a maintainer binds instead (ctypes stub: INTEGRATION.md): stateless launchers
 * over caller-owned device buffers.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless stated; the caller (PyTorch's caching
 *    allocator) owns all memory; nothing is allocated, freed or retained here;
 *  - all signals are fp32; tracks are (bs, n_tracks, n_samples) with unit sample stride
 *    and `track_row_stride` elements between consecutive (b,t) rows (reference
 *    mst/system.py:258 passes a last-dim slice - SURVEY App. C.7); stereo signals are
 *    dense (bs, 2, n_samples);
 *  - every launcher enqueues on `stream` and returns immediately; return value is a
 *    hipError_t-compatible int (0 = ok), never an exception / abort;
 *  - semantic errors keep the reference's Python exceptions: the launcher only writes
 *    a code to `status` (see mst_console_forward) and the host wrapper raises;
 *  - re-entrant: the library keeps NO process-wide mutable state (no stream pools, no caches, no environment
 *    switches in the default build); everything a call needs arrives through its arguments.
 */
#ifndef DIFFMST_HIP_H
#define DIFFMST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MST_NUM_TRACK_PARAMS 27  /* reference mst/modules.py:182 */
#define MST_NUM_FX_PARAMS 25     /* :183 */
#define MST_NUM_MASTER_PARAMS 26 /* :184 */

/* flag bits = keyword flags of AdvancedMixConsole.forward (reference mst/modules.py:316-329) */
#define MST_USE_TRACK_INPUT_FADER 0x01u
#define MST_USE_TRACK_EQ 0x02u
#define MST_USE_TRACK_COMPRESSOR 0x04u
#define MST_USE_TRACK_PANNER 0x08u
#define MST_USE_FX_BUS 0x10u /* stereo send bus + noise-shaped reverberation (reference mst/modules.py:275-284); needs mst_console_fx */
#define MST_USE_MASTER_BUS 0x20u
#define MST_USE_OUTPUT_FADER 0x40u
#define MST_SAVE_FOR_BACKWARD 0x100u /* keep intermediates in the workspace for mst_console_backward */
#define MST_NO_RANGE_CHECK 0x400u   /* the parameter tensors hold DENORMALISED values (forward_mix_console, reference
                                       mst/modules.py:186-314, which applies whatever it is given): no range check; the
                                       caller passes lo = 0, hi = 1 so that the map v*(hi-lo)+lo is the identity and the
                                       returned gradients are w.r.t. the denormalised values */
#define MST_BWD_PREPARED 0x800u     /* mst_console_backward only: mst_console_backward_prepare has already run on this workspace */
#define MST_SPLIT_BATCH 0x1000u     /* ABI v9, mst_console_forward_overlapped / _backward_overlapped: run the call as two halves of the
                                       batch on two streams (see mst_console_overlap); must be set - or not - in BOTH calls over a
                                       workspace, it selects the workspace layout.  Ignored (both calls alike) when bs < 2, with
                                       MST_USE_FX_BUS and on the gain + pan fast path */
#define MST_DEV_MULTIPASS_EQ 0x200u /* developer/test switch: keep the EQ carry scan in its own kernel (zero-state pass,
                                       carry scan, run) even when a row is short enough (<= 262144 samples) for the
                                       in-wave scans of the two-kernel path; longer rows always take three kernels */

/* *status after mst_console_forward: 0 = all parameters in [0,1]; otherwise 1000 - code with
 * code = 1 + k           track parameter index k out of range   (reference mst/modules.py:353-392)
 *        1 + 27 + k      fx-bus parameter index k out of range  (:394-422)
 *        1 + 52 + k      master-bus parameter index k           (:424-460)
 * the smallest code wins (atomic max of 1000-code), which is the reference's dictionary
 * iteration order (:79-97).  *status is only ever raised (atomic max): the caller zeroes it once and may let
 * several calls accumulate into it before reading (deferred validation).
 *
 * MST_STATUS_EXCHANGE_TIMEOUT (2000, above every range code): a kernel of the call gave up waiting for a value that another
 * workgroup of the SAME launch publishes (the block / tile aggregates of the recurrences are exchanged inside the run launches
 * instead of by a launch of their own).  The outputs of that call are poisoned (NaN) and must not be used; the Python binding
 * raises RuntimeError at its next status read (validate="sync": right after the forward; "deferred": check_parameters()).
 * Forward progress of those waits rests on ONE assumption about the hardware, which HIP does not promise: the workgroups of a launch
 * are dispatched in ascending order of their linear id (observed on gfx950, MI355X_MICROARCH.md) - a waiting workgroup only ever waits
 * for workgroups with lower ids, which are therefore resident or finished.  Every wait is bounded (MST_GRAN_SPINS polls), so a
 * platform that breaks the assumption produces this status, not a hang. */
#define MST_STATUS_EXCHANGE_TIMEOUT 2000

typedef struct mst_console_desc {
    int32_t bs;
    int32_t n_tracks;
    int64_t n_samples;
    int64_t track_row_stride; /* elements between (b,t) rows of `tracks` */
    float sample_rate;
    uint32_t flags;
    int32_t track_lookahead;  /* 2048, reference mst/modules.py:250 */
    int32_t master_lookahead; /* 1024, reference mst/modules.py:304 */
    /* denormalisation ranges v*(hi-lo)+lo, reference mst/modules.py:71-72,121-181 */
    float track_lo[MST_NUM_TRACK_PARAMS], track_hi[MST_NUM_TRACK_PARAMS];
    float master_lo[MST_NUM_MASTER_PARAMS], master_hi[MST_NUM_MASTER_PARAMS];
    /* fx bus (MST_USE_FX_BUS): ranges of the 25 reverberation parameters (band gains, band decays, mix - forward() forces the mix to
     * 1, mst/modules.py:420; under MST_NO_RANGE_CHECK the given value is applied), impulse-response length and band-pass length of
     * dasp_pytorch.functional.noise_shaped_reverberation as the reference calls it (:277-283: 65536, 1023) */
    float fx_lo[MST_NUM_FX_PARAMS], fx_hi[MST_NUM_FX_PARAMS];
    int32_t fx_ir_samples;     /* multiple of 4096 */
    int32_t fx_bandpass_taps;  /* odd, <= 1023 */
} mst_console_desc;

/* Inputs of the fx bus that are not parameters (all device pointers, caller-owned, read-only):
 *   noise    (bs*2, 12, fx_ir_samples + fx_bandpass_taps - 1) standard-normal samples - the reference draws them inside the op
 *            with torch.randn on every call; the caller draws them here (or passes fixed ones: tests)
 *   filters  (12, fx_bandpass_taps) octave-band FIR filterbank (a constant table built by the host wrapper, like the Bark
 *            filterbank of the AudioFeatureLoss)
 *   tables   mst_console_fx_tables_bytes() bytes filled once by mst_console_fx_init_tables() */
typedef struct mst_console_fx {
    const float* noise;
    const float* filters;
    const void* tables;
} mst_console_fx;
size_t mst_console_fx_tables_bytes(void);
int mst_console_fx_init_tables(void* tables, void* stream);

/* Library / ABI version (bumped when a signature changes). */
int mst_abi_version(void);

/* Bytes of scratch the console needs for `d` (includes everything forward saves for backward). */
size_t mst_console_workspace_bytes(const mst_console_desc* d);

/* AdvancedMixConsole.forward_mix_console (reference mst/modules.py:186-314) incl. the
 * denormalise + range check of :79-97: per-track gain -> 6-biquad EQ -> compressor -> pan ->
 * bus sum -> master gain/EQ/stereo-linked compressor -> output fader.
 *   tracks            (bs, n_tracks, n_samples) fp32, see strides above
 *   track_params      (bs, n_tracks, 27) normalised [0,1], dense
 *   fx_bus_params     (bs, 25) - range-checked; used when MST_USE_FX_BUS is set
 *   fx                NULL unless MST_USE_FX_BUS
 *   master_bus_params (bs, 26)
 *   mix               (bs, 2, n_samples) out
 *   mixed_tracks      (bs, 2, n_tracks, n_samples) out, or NULL to skip the API-only copy
 *   status            one int32 on the device, see codes above
 *   workspace         >= mst_console_workspace_bytes(d), 256-byte aligned */
int mst_console_forward(const mst_console_desc* d, const float* tracks, const float* track_params,
                        const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                        float* mix, float* mixed_tracks, int32_t* status, void* workspace, size_t workspace_bytes,
                        void* stream);
/* The same call (ABI v8) with the verdict of the range check MIRRORED to the host as soon as it exists: the parameters are checked by
 * the first launch of the forward; right behind it - ahead of every other launch - *status is copied to `status_host` (pinned host
 * memory, one int32) and `status_event` (a hipEvent_t, may be NULL) is recorded.  A caller that wants the reference's immediate
 * ValueError (mst/modules.py:86-89) enqueues the whole forward, then waits for THAT event - 20 us into the call - instead of for the
 * mix: the device never idles behind the host.  Codes raised by later launches of the call (MST_STATUS_EXCHANGE_TIMEOUT) are not in
 * the mirrored value; the device word is sticky and the next mirrored call (or any read of *status) sees them. */
int mst_console_forward_mirrored(const mst_console_desc* d, const float* tracks, const float* track_params,
                                 const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                 float* mix, float* mixed_tracks, int32_t* status, void* workspace, size_t workspace_bytes,
                                 void* stream, int32_t* status_host, void* status_event);

/* ABI v9: the same forward as two HALVES of the batch that overlap on the device (d->flags & MST_SPLIT_BATCH; without the flag this is
 * mst_console_forward).  The mixes of a call are independent (reference mst/modules.py:186-314 has no cross-batch term), so mixes
 * [0, bs / 2) run on `stream` and the rest on `ov->side_stream`, each over its own part of the workspace: the latency-bound stretches of
 * one half (the fp64 design chains of the first launch, the 2 x bs / 2 master-bus rows at one wave per SIMD) run beside the other half's
 * occupancy-full track kernels.  Same kernels and arithmetic per mix: the results are bit-identical to the unsplit call.
 * The library stays stateless: the caller LENDS the side stream and two events (created once, reused by every call, never shared by two
 * calls in flight); the call records `fork_event` on `stream`, makes the side stream wait for it, and before it returns makes `stream`
 * wait for `join_event` recorded behind the side stream's last launch - to the caller everything is ordered on `stream` as before
 * (memory handed to the call may be reused or freed in stream order on `stream`; the pattern is capturable into a hipGraph). */
typedef struct mst_console_overlap {
    void* side_stream; /* hipStream_t, not `stream` */
    void* fork_event;  /* hipEvent_t */
    void* join_event;  /* hipEvent_t */
} mst_console_overlap;
int mst_console_forward_overlapped(const mst_console_desc* d, const float* tracks, const float* track_params,
                                   const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                   float* mix, float* mixed_tracks, int32_t* status, void* workspace, size_t workspace_bytes,
                                   void* stream, const mst_console_overlap* ov);

/* Reverse-mode of the above (what autograd does through the reference's op graph).
 * `workspace` must be the buffer a forward call with MST_SAVE_FOR_BACKWARD filled, untouched.
 *   grad_mix           (bs, 2, n_samples)
 *   grad_mixed_tracks  (bs, 2, n_tracks, n_samples) or NULL
 *   grad_track_params  (bs, n_tracks, 27) out (w.r.t. the NORMALISED parameters)
 *   grad_master_params (bs, 26) out
 *   grad_fx_params     (bs, 25) out, or NULL (written only when MST_USE_FX_BUS; the "mix" column gets 0 when the call forced it to 1 -
 *                      AdvancedMixConsole.forward, reference mst/modules.py:420 - and the true gradient of
 *                      (1 - mix) fx_in + mix wet under MST_NO_RANGE_CHECK, i.e. forward_mix_console)
 *   grad_tracks        (bs, n_tracks, n_samples) dense out, or NULL when tracks need no grad */
int mst_console_backward(const mst_console_desc* d, const float* tracks, const float* track_params,
                         const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                         const float* grad_mix, const float* grad_mixed_tracks, float* grad_track_params,
                         float* grad_fx_params, float* grad_master_params, float* grad_tracks, int32_t* status,
                         void* workspace, size_t workspace_bytes, void* stream);
/* ... over a workspace that mst_console_forward_overlapped filled with the same d->flags (MST_SPLIT_BATCH included) */
int mst_console_backward_overlapped(const mst_console_desc* d, const float* tracks, const float* track_params,
                                    const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                    const float* grad_mix, const float* grad_mixed_tracks, float* grad_track_params,
                                    float* grad_fx_params, float* grad_master_params, float* grad_tracks, int32_t* status,
                                    void* workspace, size_t workspace_bytes, void* stream, const mst_console_overlap* ov);
/* status (ABI v7; may be null): the forward's status word, raised to MST_STATUS_EXCHANGE_TIMEOUT by a backward launch whose
 * in-launch exchange gave up - the gradients of that call are poisoned. */

/* The part of the backward that depends only on what forward saved (the all-pole carry scan of the coefficient-gradient
 * pass, 15 us at 64 x 262144): a caller may enqueue it on ANOTHER stream once forward has finished there, where it overlaps
 * with whatever runs between the two calls (the loss), make `stream` of mst_console_backward wait for it and pass
 * MST_BWD_PREPARED in d->flags.  Without the flag mst_console_backward runs it itself.  (diffmst_hip/modules.py does this.) */
int mst_console_backward_prepare(const mst_console_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-resolution STFT loss: auraloss.freq.MultiResolutionSTFTLoss as configured by the
 * reference (configs/models/naive.yaml:54-68; called as loss(pred, target), mst/system.py:331).
 * pred / target are (rows, n_samples) dense fp32 with rows = bs * channels. */
#define MST_MAX_RESOLUTIONS 8
typedef struct mst_mrstft_desc {
    int32_t rows;
    int64_t n_samples;
    int32_t n_res;
    int32_t fft_size[MST_MAX_RESOLUTIONS];   /* powers of two, 128..8192 */
    int32_t hop_size[MST_MAX_RESOLUTIONS];
    int32_t win_length[MST_MAX_RESOLUTIONS]; /* periodic Hann, centre-padded to fft_size */
    float w_sc, w_log_mag, w_lin_mag;        /* auraloss defaults 1, 1, 0 */
    int32_t sc_per_example;                  /* 1: mean over rows of per-row norm ratios (0.4.0); 0: global ratio */
    float eps;                               /* clamp of |X|^2, 1e-8 */
} mst_mrstft_desc;

/* Twiddle + window tables: built once per descriptor into caller memory, read-only afterwards. */
size_t mst_mrstft_tables_bytes(const mst_mrstft_desc* d);
int mst_mrstft_init_tables(const mst_mrstft_desc* d, void* tables, void* stream);
/* Per-call scratch; forward leaves what backward needs in it: the partial sums and coefficients, and - for the reference's
 * resolutions (512 / 2048 / 8192 points, hop = n_fft / 2) - the prediction's spectrum and the target's clamped magnitudes of every
 * bin and frame (12 bytes each; ~9.6 MB per row of 262144 samples): the backward reads them instead of transforming again, and
 * touches `pred` / `target` only on the generic path (other resolutions). */
size_t mst_mrstft_workspace_bytes(const mst_mrstft_desc* d);
/* loss: one fp32 on the device. */
int mst_mrstft_forward(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                       float* loss, void* workspace, size_t workspace_bytes, void* stream);
/* The same loss with NOTHING kept for a backward (ABI v8): the spectra / magnitude planes are not written (153 MB per call at BASELINE
 * cfg #2), the workspace afterwards holds no valid backward state - for callers that only want the value (validation loops,
 * torch.no_grad()).  Same workspace size. */
int mst_mrstft_forward_eval(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                            float* loss, void* workspace, size_t workspace_bytes, void* stream);
/* Sharded evaluation (the batch rows are split over ranks, one process per GPU; reference: DDP over the batch axis,
 * configs/config.yaml:34-42).  Every term of the loss is a mean over rows except the batch-global spectral-convergence
 * ratio (sc_per_example = 0), whose two squared norms must be summed over ALL ranks before the division:
 *   mst_mrstft_forward_partial   transforms + reductions of this rank's rows; totals (n_res, 4) float64 on the device =
 *                                {sum (|Y|-|X|)^2, sum |Y|^2, sum |log|X| - log|Y||, sum ||X| - |Y||} per resolution
 *   (the caller all-reduces `totals` over the ranks - SUM - into global_totals)
 *   mst_mrstft_forward_finish    loss of this rank = global ratio + this rank's mean terms, so that the MEAN of the rank
 *                                losses is the single-process loss over the global batch; the backward coefficients carry
 *                                `world` on the ratio term (the adjoint of the all-reduce), so that gradients averaged over
 *                                ranks (DDP) equal the single-process gradients.  global_totals = NULL, world = 1: plain
 *                                local evaluation, identical to mst_mrstft_forward. */
int mst_mrstft_forward_partial(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                               double* totals, void* workspace, size_t workspace_bytes, void* stream);
int mst_mrstft_forward_finish(const mst_mrstft_desc* d, const double* global_totals, int32_t world, float* loss,
                              void* workspace, size_t workspace_bytes, void* stream);
/* grad_loss: one fp32 on the device (dL/dloss); grad_pred (rows, n_samples) is overwritten. */
int mst_mrstft_backward(const mst_mrstft_desc* d, const float* pred, const float* target, const void* tables,
                        const float* grad_loss, float* grad_pred, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * batch_stereo_peak_normalize (reference mst/utils.py:14-29): y = x / clamp(max_{ch,t}|x|, 1e-8)
 * per batch item; x, y, grads are dense (bs, 2, n_samples).  The workspace written by forward is
 * what backward reads (peak and arg-max per batch item). */
size_t mst_peak_normalize_workspace_bytes(int32_t bs, int64_t n_samples);
int mst_peak_normalize_forward(const float* x, float* y, int32_t bs, int64_t n_samples, void* workspace,
                               size_t workspace_bytes, void* stream);
int mst_peak_normalize_backward(const float* x, const float* grad_y, float* grad_x, int32_t bs, int64_t n_samples,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * AudioFeatureLoss (reference mst/loss.py:198-260): five weighted MSE terms between features of
 * pred and target, both dense (bs, 2, n_samples): rms, crest factor, stereo width, stereo imbalance
 * (:127-195) and the 24-band Bark spectrum of mid/side (:62-124; STFT 32768 / hop 8192 / Hann).
 *   weights5     host array of the five weights, in that order
 *   filterbank   device (16385, 24) fp32 = barkscale_fbanks(16385, 20, 20000, 24, sample_rate)
 *                (reference mst/filter.py:107-161; a constant table built by the host wrapper)
 *   losses5      device, the five weighted losses in the reference's key order */
size_t mst_afloss_tables_bytes(void);
int mst_afloss_init_tables(void* tables, void* stream);
size_t mst_afloss_workspace_bytes(int32_t bs, int64_t n_samples);
int mst_afloss_forward(const float* pred, const float* target, int32_t bs, int64_t n_samples, const float* weights5,
                       const void* tables, const float* filterbank, float* losses5, void* workspace,
                       size_t workspace_bytes, void* stream);
/* grad_losses5: device, dL/d(loss_k); grad_pred (bs, 2, n_samples) is overwritten. */
int mst_afloss_backward(const float* pred, const float* target, int32_t bs, int64_t n_samples, const float* weights5,
                        const void* tables, const float* filterbank, const float* grad_losses5, float* grad_pred,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Spectrogram encoder: SpectrogramEncoder (reference mst/modules.py:740-806) = STFT front end +
 * Cnn14 (mst/panns.py:126-209, ConvBlock :27-85).  The 3x3 convolutions run on the matrix cores (MFMA); activations are
 * NHWC in the workspace; master weights, BatchNorm statistics and every gradient are fp32 in torch's own layouts.
 *
 * mst_spectrogram_forward: x (rows, n_samples) -> spec (rows, frames, bins) fp32 = (|STFT| + 1e-8)^0.3 with
 *   frames = 1 + n_samples / hop, bins = n_fft / 2 + 1 (torch.stft: periodic Hann, centre frames, reflect padding;
 *   modules.py:789-800).  n_fft = 2048 (the reference's the yaml files under configs/models).  Note the (frames, bins) order: the network
 *   below runs on the transposed image (its weights are transposed on the fly), nothing of it is user-visible. */
size_t mst_spectrogram_tables_bytes(void);
int mst_spectrogram_init_tables(void* tables, void* stream);
int mst_spectrogram_forward(const float* x, int32_t rows, int64_t n_samples, int32_t n_fft, int32_t hop, const void* tables,
                            float* spec, void* stream);

#define MST_CNN14_CONVS 12 /* six ConvBlocks x two convolutions, in forward order */
typedef struct mst_cnn14_desc {
    int32_t n;          /* signals = batch x channels of the encoder call */
    int32_t frames, bins;
    int32_t embed_dim;  /* num_classes of the final Linear(2048, embed_dim) */
    int32_t precision;  /* 0: bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16); 1: fp32 operands (v_mfma_f32_16x16x4_f32);
                         * 2: "bf16x3" - fp32 tensors everywhere (same buffers and sizes as 1), the convolutions' operands split into
                         *    bf16 (hi, lo) pairs on the way into the matrix pipe, x y ~ hi hi + hi lo + lo hi with fp32 accumulation
                         *    (~2^-17 per product); 3: "bf16x6" - the same with (hi, mid, lo) triples and the six product terms down to
                         *    2^-24: fp32-grade results (passes the parity bounds of precision 1) */
    int32_t training;   /* 1: BatchNorm2d with batch statistics (returned in batch_stats); 0: running statistics */
    float bn_eps;       /* 1e-5 */
    int32_t world;      /* 0 or 1: statistics over this call's signals.  > 1 (with a sync hook, mst_cnn14_forward_sync): BatchNorm statistics
                         * over `world` calls of equal n - torch.nn.SyncBatchNorm, reference configs/config.yaml:41 sync_batchnorm */
} mst_cnn14_desc;
typedef struct mst_cnn14_params { /* device pointers, fp32, torch layouts */
    const float* conv_w[MST_CNN14_CONVS];   /* (co, ci, 3, 3); channels 1-64-64-128-128-...-2048-2048 */
    const float* bn_gamma[MST_CNN14_CONVS]; /* BatchNorm2d weight */
    const float* bn_beta[MST_CNN14_CONVS];  /* BatchNorm2d bias */
    const float* bn_mean[MST_CNN14_CONVS];  /* running_mean (read when training = 0) */
    const float* bn_var[MST_CNN14_CONVS];   /* running_var */
    const float* fc_w;                      /* (embed_dim, 2048) */
    const float* fc_b;                      /* (embed_dim) */
} mst_cnn14_params;
typedef struct mst_cnn14_grads { /* device pointers, fp32, same layouts; every array is overwritten */
    float* conv_w[MST_CNN14_CONVS];
    float* bn_gamma[MST_CNN14_CONVS];
    float* bn_beta[MST_CNN14_CONVS];
    float* fc_w;
    float* fc_b;
} mst_cnn14_grads;
size_t mst_cnn14_workspace_bytes(const mst_cnn14_desc* d);
/* spec (n, frames, bins) fp32 -> embed (n, embed_dim).  batch_stats (12, 2, 2048) fp32 or NULL: per convolution the batch
 * mean and the biased batch variance of its output (training = 1) - what nn.BatchNorm2d folds into its running statistics.
 * The workspace keeps what mst_cnn14_backward needs. */
int mst_cnn14_forward(const mst_cnn14_desc* d, const float* spec, const mst_cnn14_params* params, float* embed,
                      float* batch_stats, void* workspace, size_t workspace_bytes, void* stream);
/* grad_embed (n, embed_dim) -> gradients of every parameter.  The spectrogram gets no gradient (the encoder's inputs are
 * audio, which the reference never differentiates: mst/system.py:284-300). */
int mst_cnn14_backward(const mst_cnn14_desc* d, const float* spec, const mst_cnn14_params* params, const float* grad_embed,
                       const mst_cnn14_grads* grads, void* workspace, size_t workspace_bytes, void* stream);

/* Cross-rank BatchNorm statistics (reference: Lightning's sync_batchnorm = torch.nn.SyncBatchNorm over the process group,
 * configs/config.yaml:41, mst/panns.py:27-85).  The library has no communication of its own: between the per-channel fp64 sums of a
 * layer and their use it calls `sync(user, sums, n_doubles, stream)`, which must leave the ELEMENT-WISE SUM over all ranks in `sums`,
 * ordered after everything already queued on `stream` and before everything queued later (torch.distributed.all_reduce on a tensor that
 * aliases the workspace does exactly that).  Twelve calls per forward (sum, sum of squares of every convolution's output) and twelve
 * per backward (sum g, sum g xhat of every BatchNorm adjoint).  d->world = number of ranks (equal n on every rank).  The parameter
 * gradients of BatchNorm come out as (global sum) / world on every rank - what an averaging DistributedDataParallel leaves of the
 * per-rank sums.  sync == NULL or world <= 1 or training == 0: identical to mst_cnn14_forward / _backward. */
typedef void (*mst_sync_fn)(void* user, double* sums, size_t n_doubles, void* stream);
int mst_cnn14_forward_sync(const mst_cnn14_desc* d, const float* spec, const mst_cnn14_params* params, float* embed,
                           float* batch_stats, void* workspace, size_t workspace_bytes, void* stream, mst_sync_fn sync, void* user);
int mst_cnn14_backward_sync(const mst_cnn14_desc* d, const float* spec, const mst_cnn14_params* params, const float* grad_embed,
                            const mst_cnn14_grads* grads, void* workspace, size_t workspace_bytes, void* stream, mst_sync_fn sync,
                            void* user);

/* ---- TransformerController encoder stack (reference mst/modules.py:848-854, :893-895: torch.nn.TransformerEncoder of
 * post-norm TransformerEncoderLayer(d_model, nhead, dim_feedforward, relu, dropout 0, batch_first) over the (bs, seq, d_model)
 * token sequence, with the key-padding mask of :880-890).  fp32 operands on v_mfma_f32_16x16x4_f32.  Limits: seq <= 128,
 * d_model % 128 == 0, d_model <= 1024, d_model / nhead <= 64, d_ff % 128 == 0, n_layers <= 16 (workspace_bytes returns 0 otherwise).
 * Stateless and re-entrant like the rest of the library; run-to-run deterministic (no atomics). */
typedef struct mst_ctrl_desc {
    int32_t bs, seq, d_model, nhead, d_ff, n_layers;
    float ln_eps; /* 1e-5 */
} mst_ctrl_desc;
typedef struct mst_ctrl_layer { /* device pointers, fp32, torch layouts (the state_dict entries of one layer) */
    const float* in_proj_weight;  /* self_attn.in_proj_weight (3 d_model, d_model) */
    const float* in_proj_bias;    /* (3 d_model) */
    const float* out_proj_weight; /* self_attn.out_proj.weight (d_model, d_model) */
    const float* out_proj_bias;
    const float* linear1_weight;  /* (d_ff, d_model) */
    const float* linear1_bias;
    const float* linear2_weight;  /* (d_model, d_ff) */
    const float* linear2_bias;
    const float* norm1_weight;
    const float* norm1_bias;
    const float* norm2_weight;
    const float* norm2_bias;
} mst_ctrl_layer;
typedef struct mst_ctrl_layer_grads { /* same entries; every array is overwritten */
    float* in_proj_weight;
    float* in_proj_bias;
    float* out_proj_weight;
    float* out_proj_bias;
    float* linear1_weight;
    float* linear1_bias;
    float* linear2_weight;
    float* linear2_bias;
    float* norm1_weight;
    float* norm1_bias;
    float* norm2_weight;
    float* norm2_bias;
} mst_ctrl_layer_grads;
size_t mst_ctrl_workspace_bytes(const mst_ctrl_desc* d);
/* tokens (bs, seq, d_model) -> out (bs, seq, d_model).  key_padding_mask (bs, seq) bytes, non-zero = ignore that key, or NULL.
 * layers: HOST array of n_layers entries.  The workspace keeps what mst_ctrl_backward needs. */
int mst_ctrl_forward(const mst_ctrl_desc* d, const float* tokens, const uint8_t* key_padding_mask, const mst_ctrl_layer* layers,
                     float* out, void* workspace, size_t workspace_bytes, void* stream);
/* grad_out (bs, seq, d_model) -> gradients of every layer parameter (grads: HOST array of n_layers entries) and of the tokens. */
int mst_ctrl_backward(const mst_ctrl_desc* d, const float* tokens, const mst_ctrl_layer* layers, const float* grad_out,
                      const mst_ctrl_layer_grads* grads, float* grad_tokens, void* workspace, size_t workspace_bytes, void* stream);

/* The controller's own ends (reference mst/modules.py:841-859, :866-914): the token sequence
 *   cat(track_embeds + track_embedding, mix_embeds + mix_embedding, fx_bus_embedding, master_bus_embedding)  (seq = n_tracks + 4)
 * with the key-padding mask extended by four always-attended tokens, and the three heads sigmoid(Linear) of the track tokens, the fx
 * token (seq - 2) and the master token (seq - 1).  n_t / n_f / n_m <= 32 outputs, d_model <= 1024.  A NULL head cotangent (g_f, g_m) means
 * that head's output was not used: its weight gradients are NOT written (the caller reports None, like autograd). */
typedef struct mst_ctrl_io { /* device pointers, fp32 */
    const float* track_embedding;      /* (d_model) */
    const float* mix_embedding;        /* (2, d_model) */
    const float* fx_bus_embedding;     /* (d_model) */
    const float* master_bus_embedding; /* (d_model) */
    const float* track_w;  const float* track_b;   /* track_projection (n_t, d_model), (n_t) */
    const float* fx_w;     const float* fx_b;
    const float* master_w; const float* master_b;
} mst_ctrl_io;
typedef struct mst_ctrl_io_grads { /* same entries; written by the two backward calls */
    float* track_embedding;
    float* mix_embedding;
    float* fx_bus_embedding;
    float* master_bus_embedding;
    float* track_w;  float* track_b;
    float* fx_w;     float* fx_b;
    float* master_w; float* master_b;
} mst_ctrl_io_grads;
int mst_ctrl_tokens_forward(const mst_ctrl_desc* d, int32_t n_tracks, const float* track_embeds, const float* mix_embeds,
                            const uint8_t* track_padding_mask, const mst_ctrl_io* io, float* tokens, uint8_t* key_padding_mask_out,
                            void* stream);
int mst_ctrl_heads_forward(const mst_ctrl_desc* d, int32_t n_tracks, const float* z, const mst_ctrl_io* io, int32_t n_t, int32_t n_f,
                           int32_t n_m, float* out_t, float* out_f, float* out_m, void* stream);
size_t mst_ctrl_heads_scratch_bytes(const mst_ctrl_desc* d, int32_t n_tracks);
/* grad_z (bs, seq, d_model): every row is written (the mix tokens get zeros); grads: the six head entries */
int mst_ctrl_heads_backward(const mst_ctrl_desc* d, int32_t n_tracks, const float* z, const mst_ctrl_io* io, int32_t n_t, int32_t n_f,
                            int32_t n_m, const float* out_t, const float* out_f, const float* out_m, const float* g_t, const float* g_f,
                            const float* g_m, const mst_ctrl_io_grads* grads, float* grad_z, void* scratch, void* stream);
/* grads: the four embedding entries; the cotangents of track_embeds / mix_embeds are the rows [0, T) / [T, T + 2) of grad_tokens */
int mst_ctrl_tokens_backward(const mst_ctrl_desc* d, int32_t n_tracks, const float* grad_tokens, const mst_ctrl_io_grads* grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFMST_HIP_H */
