// mst_console.hip - C-ABI entry points of the mix console (include/diffmst_hip.h) and the
// launch sequences behind them.  No allocation, no host sync: everything is enqueued on the
// caller's stream over the caller's workspace.
#include "mst_kernels.h"

namespace mst {
static int check_desc(const mst_console_desc* d) {
    if (!d || d->bs <= 0 || d->n_tracks <= 0 || d->n_samples <= 0) return hipErrorInvalidValue;
    if (d->flags & MST_USE_FX_BUS) {
        if (d->fx_ir_samples < 4096 || d->fx_ir_samples % 4096 || d->fx_ir_samples > (1 << 20)) return hipErrorInvalidValue;
        if (d->fx_bandpass_taps < 1 || d->fx_bandpass_taps > 1023 || !(d->fx_bandpass_taps & 1)) return hipErrorInvalidValue;
    }
    if (!(d->flags & MST_USE_TRACK_PANNER)) return hipErrorInvalidValue;  // reference branch is broken (mst/modules.py:269)
    if (d->track_row_stride < d->n_samples) return hipErrorInvalidValue;
    if ((d->track_lookahead & 3) || (d->master_lookahead & 3) || d->track_lookahead < 0 || d->master_lookahead < 0)
        return hipErrorInvalidValue;
    return hipSuccess;
}
}  // namespace mst

using namespace mst;

// build-time A/B switch (-DMST_ALLPOLE_SEPARATE): keep the all-pole zero-state pass as its own backward kernel
static constexpr bool fuse_allpole() {
#ifdef MST_ALLPOLE_SEPARATE
    return false;
#else
    return true;
#endif
}

// build-time A/B switch (-DMST_COMP_ZS_SEPARATE): the smoother's zero-state passes as launches of their own (k_comp_zs<2>, k_comp_bwd_zs)
// instead of block aggregates exchanged inside the run launches (mst_common.h: granules)
static constexpr bool fuse_comp_zs() {
#ifdef MST_COMP_ZS_SEPARATE
    return false;
#else
    return true;
#endif
}

// build-time A/B switch (-DMST_EQ_ZS_VALU): zero-state EQ passes on the vector ALU (round-2 kernels) instead of the matrix pipe
static constexpr bool mfma_zs() {
#ifdef MST_EQ_ZS_VALU
    return false;
#else
    return true;
#endif
}

// build-time A/B switch (-DMST_EQ_ZS_SEPARATE): the EQ's zero-state passes as launches of their own (k_eq_zs_mfma) instead of inside the
// run launches (ZsIn, mst_kernels.h)
static constexpr bool fuse_eq_zs() {
#ifdef MST_EQ_ZS_SEPARATE
    return false;
#else
    return true;
#endif
}

extern "C" int mst_abi_version(void) { return 9; }
#ifdef MST_DEV_PROBE  // developer probe (tools/sidestream_probe.py): an event recorded in the middle of the forward's launch sequence
static hipEvent_t g_probe_ev = nullptr;
static int g_probe_where = 0;
extern "C" void mst_debug_set_mid_event(void* ev, int where) { g_probe_ev = (hipEvent_t)ev; g_probe_where = where; }
#endif

extern "C" size_t mst_console_fx_tables_bytes(void) { return (size_t)8192 * 2 * sizeof(float); }
extern "C" int mst_console_fx_init_tables(void* tables, void* stream) {
    if (!tables) return hipErrorInvalidValue;
    launch_fx_tables((float*)tables, (hipStream_t)stream);
    return (int)hipGetLastError();
}

static FxPlan fx_plan(const Layout& L) {
    FxPlan p{};
    p.bs = L.bs;
    p.S = L.fxS;
    p.taps = L.fxTaps;
    p.K = L.fxK;
    p.nblk = L.fxBlk;
    p.nblk_ir = L.fxBlkIr;
    p.n = L.N;
    p.Ns = row_stride(L.N);
    p.rcfx = L.fx_rc; p.fx_in = L.fx_in; p.wnf = L.fx_wnf; p.ir = L.fx_ir; p.Xs = L.fx_Xs; p.Hs = L.fx_Hs; p.Ys = L.fx_Ys;
    p.dXs = L.fx_dXs; p.dHs = L.fx_dHs; p.dir = L.fx_dir; p.dfx_in = L.fx_din; p.fxpart = L.fx_part; p.Hf = L.fx_Hf; p.mixv = L.fx_mix; p.dry = L.fx_dry;
    return p;
}

// MST_SPLIT_BATCH (ABI v9): the mixes of a call are dealt to two halves that run as two independent console calls - the first on the
// caller's stream, the second on a side stream the caller lends (mst_console_overlap) - each over its own part of the workspace.  The
// two halves' launches interleave on the device: the first half's latency-bound stretches (k_prep's fp64 chains, the lone-wave master
// chain, the single-workgroup tails) run beside the second half's occupancy-full track kernels and vice versa.  Same kernels, same
// arithmetic per mix: results are bit-identical to the unsplit call.
static bool split_on(const mst_console_desc* d) {
    return (d->flags & MST_SPLIT_BATCH) && d->bs >= 2 && !(d->flags & MST_USE_FX_BUS) && !basic_path(d);
}
struct Halves {
    mst_console_desc da, db;
    int64_t ws_b;  // float offset of the second half's workspace
    int64_t total;
};
static Halves make_halves(const mst_console_desc* d) {
    Halves h{*d, *d, 0, 0};
    h.da.bs = d->bs / 2;
    h.db.bs = d->bs - h.da.bs;
    h.da.flags &= ~MST_SPLIT_BATCH;
    h.db.flags &= ~MST_SPLIT_BATCH;
    h.ws_b = make_layout(&h.da).total;
    h.total = h.ws_b + make_layout(&h.db).total;
    return h;
}
static bool overlap_ok(const mst_console_overlap* ov) { return ov && ov->side_stream && ov->fork_event && ov->join_event; }

extern "C" size_t mst_console_workspace_bytes(const mst_console_desc* d) {
    if (check_desc(d) != hipSuccess) return 0;
    if (split_on(d)) return (size_t)make_halves(d).total * sizeof(float);
    return (size_t)make_layout(d).total * sizeof(float);
}

static int console_forward_impl(const mst_console_desc* d, const float* tracks, const float* track_params,
                                const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                float* mix, float* mixed_tracks, int32_t* status, void* workspace, size_t workspace_bytes,
                                void* stream_, int32_t* status_host, void* status_event) {
    if (int e = check_desc(d)) return e;
    const Layout L = make_layout(d);
    if (!workspace || workspace_bytes < (size_t)L.total * sizeof(float) || ((uintptr_t)workspace & 255)) return hipErrorInvalidValue;
    if (!tracks || !track_params || !fx_bus_params || !master_bus_params || !mix || !status) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    float* ws = (float*)workspace;
    const int64_t n = L.N, Ns = row_stride(L.N);
    const bool save = d->flags & MST_SAVE_FOR_BACKWARD;
    const int aligned = (n % 4 == 0) && !((uintptr_t)mix & 15) && !((uintptr_t)mixed_tracks & 15);
    const bool t_comp = d->flags & MST_USE_TRACK_COMPRESSOR;
    const bool m_on = d->flags & MST_USE_MASTER_BUS;
    const bool o_on = d->flags & MST_USE_OUTPUT_FADER;
    const bool fx_on = d->flags & MST_USE_FX_BUS;
    if (fx_on && (!fx || !fx->noise || !fx->filters || !fx->tables)) return hipErrorInvalidValue;
    if (basic_path(d)) {  // BASELINE cfg #1: gain + pan + bus sum in one launch
        BasicArgs ba{tracks, track_params, fx_bus_params, master_bus_params, mix, mixed_tracks, status, nullptr, nullptr, nullptr, nullptr, nullptr,
                     ws + L.cp_t, *d};
        launch_basic_forward(ba, stream);
        if (status_host) {  // (see below: the verdict of the range check, mirrored to the host)
            (void)hipMemcpyAsync(status_host, status, sizeof(int32_t), hipMemcpyDeviceToHost, stream);
            if (status_event) (void)hipEventRecord((hipEvent_t)status_event, stream);
        }
        return (int)hipGetLastError();
    }

    PrepArgs pa{track_params, fx_bus_params, master_bus_params, ws + L.rc_t, ws + L.rc_m,
                ws + L.powF_t, ws + L.powF_m, ws + L.powA_t, ws + L.powA_m, ws + L.powP_t, ws + L.powP_m,
                ws + L.pow1F_t, ws + L.pow1F_m, ws + L.pow1A_t, ws + L.pow1A_m, ws + L.wzF_t, ws + L.wzF_m, ws + L.wzA_t, ws + L.wzA_m, fx_on ? ws + L.fx_rc : nullptr, fx_on ? ws + L.fx_mix : nullptr, status, L.R, L.bs, L.KE,
                L.eq1, *d, (gran_t*)(ws + L.gran_f), L.gran_nf + L.gran_nb, nullptr, 0, 0};
#ifndef MST_PREP_RIDER_MAX_BYTES
#define MST_PREP_RIDER_MAX_BYTES (128ll << 20)  // the riders pull the track rows through the 256 MB Infinity Cache for the launch that follows:
                                                // only while the rows fit it (cfg #2: 67 MB).  At cfg #3 (537 MB) they were 120 us of k_prep
                                                // for rows that were gone again before the EQ pass read them
#endif
    if (n % 4 == 0 && d->track_row_stride % 4 == 0 && !((uintptr_t)tracks & 15) && (int64_t)L.R * n * 4 <= MST_PREP_RIDER_MAX_BYTES) {
        pa.pf_src = tracks;
        pa.pf_stride = d->track_row_stride;
        pa.pf_n = n;
    }
    launch_prep(pa, stream);
    // the range check is complete when k_prep is: its verdict travels to the host NOW, behind k_prep and ahead of the rest of the forward
    // (mst_console_forward_mirrored) - a caller that wants the reference's immediate ValueError waits for this copy, not for the mix
    if (status_host && status) {
        (void)hipMemcpyAsync(status_host, status, sizeof(int32_t), hipMemcpyDeviceToHost, stream);
        if (status_event) (void)hipEventRecord((hipEvent_t)status_event, stream);
    }
#ifdef MST_DEV_PROBE
    if (g_probe_ev && g_probe_where == 0) (void)hipEventRecord(g_probe_ev, stream);
#endif

    // ---- tracks: EQ (zs -> carry scan -> run), compressor smoother (zs -> scan), apply + pan + bus sum
    // L.eq1: the carries are scanned inside the zs / run kernels (the run kernel reads the zs kernel's states directly)
    const float* p1F_t = L.eq1 ? ws + L.pow1F_t : nullptr;
    const float* p1F_m = L.eq1 ? ws + L.pow1F_m : nullptr;
    const float* sE_t = L.eq1 ? ws + L.zE_t : ws + L.sE_t;
    const float* sE_m = L.eq1 ? ws + L.zE_m : ws + L.sE_m;
    // a call that saves for backward also leaves the all-pole zero-state ends of the coefficient-gradient pass
    float* zP_t = (save && fuse_allpole()) ? ws + L.zP_t : nullptr;
    float* zP_m = (save && fuse_allpole()) ? ws + L.zP_m : nullptr;
    // round 5: the zero-state pass rides inside the run launch (tile aggregates exchanged as granules) wherever the run is a SCAN1 kernel
    // with that variant: the tracks' compressor-fused run and both master-bus runs
    const bool zsin = L.eq1 && mfma_zs() && fuse_eq_zs();
    const ZsIn zi_t{ws + L.wzF_t, (gran_t*)(ws + L.eqg_f), L.eqg_nf, status};
    const ZsIn zi_m{ws + L.wzF_m, (gran_t*)(ws + L.eqg_f) + (int64_t)L.R * kMaxTiles1 * kStates, L.eqg_nf, status};
    // tracks: only while (almost) every tile of the launch is resident at once - with many rounds of workgroups (cfg #3: 32768 tiles, 8
    // rounds) the stand-alone zero-state launch streams the rows at the HBM rate and the merged form measured 0.6 % slower
#ifndef MST_ZSIN_MAX_TILES
#define MST_ZSIN_MAX_TILES 8192
#endif
    const bool zsin_t = zsin && t_comp && (int64_t)L.R * L.ntE <= MST_ZSIN_MAX_TILES;
    if (zsin_t) {}
    else if (L.eq1 && mfma_zs()) launch_eq_zs_mfma(EQ_FWD, tracks, d->track_row_stride, ws + L.wzF_t, L.R, ws + L.zE_t, L.ncE_pad, n, L.R, stream, p1F_t, L.ntE, ws + L.aggF_t);
    else launch_cascade(EQ_FWD, false, tracks, d->track_row_stride, nullptr, 0, ws + L.rc_t, L.R, nullptr, ws + L.zE_t, L.ncE_pad, n, L.R, stream, p1F_t, L.ntE, ws + L.aggF_t);
    if (!L.eq1) launch_scan12(false, ws + L.zE_t, ws + L.sE_t, ws + L.powF_t, L.R, L.ncE, L.ncE_pad, L.KE, L.R, stream);
    if (t_comp)  // EQ run fused with the gain computer + per-block envelope aggregates
        launch_cascade_run_gc(tracks, d->track_row_stride, ws + L.u_t, Ns, ws + L.rc_t, L.R, sE_t, L.ncE_pad, n, L.R,
                              ws + L.zS_t, L.nblkC, stream, p1F_t, L.ntE, ws + L.aggF_t, zP_t, zsin_t ? &zi_t : nullptr);
    else
        launch_cascade(EQ_FWD, true, tracks, d->track_row_stride, ws + L.u_t, Ns, ws + L.rc_t, L.R, sE_t, nullptr, L.ncE_pad, n, L.R, stream, p1F_t, L.ntE, ws + L.aggF_t, zP_t);
    const bool bus_is_mix = !m_on && !o_on;
    float* busp = bus_is_mix ? mix : ws + L.bus;
    const int64_t bus_stride = bus_is_mix ? n : Ns;
    TrackApplyArgs ta{ws + L.u_t, Ns, ws + L.rc_t, ws + L.zS_t, (save && t_comp) ? ws + L.gs_t : nullptr,
                      busp, bus_stride, mixed_tracks, fx_on ? ws + L.fx_in : nullptr,
                      L.T, L.ncC_pad, d->track_lookahead, t_comp ? 1 : 0, n, aligned};
    launch_apply_tracks(ta, L.bs, stream);
#ifdef MST_DEV_PROBE
    if (g_probe_ev && g_probe_where == 1) (void)hipEventRecord(g_probe_ev, stream);
#endif
    // ---- fx bus: reverberate the send bus and add it to the stereo bus (reference mst/modules.py:275-284)
    if (fx_on) launch_fx_forward(fx_plan(L), fx->noise, fx->filters, (const float*)fx->tables, ws, busp, bus_stride, stream);

    // ---- master bus
    if (m_on) {
        const bool apscan_m = L.apscan_fwd && zP_m && zP_t && p1F_m;
        const bool zsin_m = zsin && apscan_m;
        if (zsin_m) {}
        else if (L.eq1 && mfma_zs()) launch_eq_zs_mfma(EQ_FWD, ws + L.bus, Ns, ws + L.wzF_m, 0, ws + L.zE_m, L.ncE_pad, n, 2 * L.bs, stream, p1F_m, L.ntE, ws + L.aggF_m);
        else launch_cascade(EQ_FWD, false, ws + L.bus, Ns, nullptr, 0, ws + L.rc_m, 0, nullptr, ws + L.zE_m, L.ncE_pad, n, 2 * L.bs, stream, p1F_m, L.ntE, ws + L.aggF_m);
        if (!L.eq1) launch_scan12(false, ws + L.zE_m, ws + L.sE_m, ws + L.powF_m, 0, L.ncE, L.ncE_pad, L.KE, 2 * L.bs, stream);
        if (apscan_m)  // + the track rows' all-pole carry scan as extra workgroups of this (one wave per SIMD) launch
            launch_master_run_apscan(ws + L.bus, Ns, ws + L.v_m, Ns, ws + L.rc_m, sE_m, L.ncE_pad, n, 2 * L.bs, stream, p1F_m, L.ntE, ws + L.aggF_m, zP_m,
                                     ws + L.zP_t, ws + L.sP_t, ws + L.powP_t, L.R * 12, L.ncE, L.apscan_sh, EQ_FWD, zsin_m ? &zi_m : nullptr);
        else
            launch_cascade(EQ_FWD, true, ws + L.bus, Ns, ws + L.v_m, Ns, ws + L.rc_m, 0, sE_m, nullptr, L.ncE_pad, n, 2 * L.bs, stream, p1F_m, L.ntE, ws + L.aggF_m, zP_m);
        if (!fuse_comp_zs()) launch_comp_zs(2, ws + L.v_m, Ns, ws + L.rc_m, ws + L.zS_m, L.ncC_pad, n, L.bs, stream);
        MasterApplyArgs ma{ws + L.v_m, Ns, ws + L.rc_m, ws + L.zS_m, save ? ws + L.gs_m : nullptr, mix, n,
                           L.ncC_pad, d->master_lookahead, 1, n, aligned, fuse_comp_zs() ? (gran_t*)(ws + L.gran_f) : nullptr, (int64_t)L.bs * L.nblkC, status};
        launch_apply_master(ma, L.bs, stream);
    } else if (o_on) {
        MasterApplyArgs ma{ws + L.bus, Ns, ws + L.rc_m, nullptr, nullptr, mix, n, L.ncC_pad, 0, 0, n, aligned};
        launch_apply_master(ma, L.bs, stream);
    }
    return (int)hipGetLastError();
}

// carry scan of the all-pole bank's chunk states: every row, or - when the forward's master-bus run already carried the track rows'
// (Layout::apscan_fwd) - the master rows only
extern "C" int mst_console_forward(const mst_console_desc* d, const float* tracks, const float* track_params,
                                   const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                   float* mix, float* mixed_tracks, int32_t* status, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (d && (d->flags & MST_SPLIT_BATCH)) return hipErrorInvalidValue;  // the split form needs the side stream: mst_console_forward_overlapped
    return console_forward_impl(d, tracks, track_params, fx_bus_params, master_bus_params, fx, mix, mixed_tracks, status, workspace,
                                workspace_bytes, stream, nullptr, nullptr);
}
extern "C" int mst_console_forward_overlapped(const mst_console_desc* d, const float* tracks, const float* track_params,
                                              const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                              float* mix, float* mixed_tracks, int32_t* status, void* workspace, size_t workspace_bytes,
                                              void* stream, const mst_console_overlap* ov) {
    if (int e = check_desc(d)) return e;
    if (!split_on(d))
        return console_forward_impl(d, tracks, track_params, fx_bus_params, master_bus_params, fx, mix, mixed_tracks, status, workspace,
                                    workspace_bytes, stream, nullptr, nullptr);
    if (!overlap_ok(ov) || ov->side_stream == stream) return hipErrorInvalidValue;
    const Halves h = make_halves(d);
    if (!workspace || workspace_bytes < (size_t)h.total * sizeof(float) || ((uintptr_t)workspace & 255)) return hipErrorInvalidValue;
    if (!tracks || !track_params || !fx_bus_params || !master_bus_params || !mix || !status) return hipErrorInvalidValue;
    hipStream_t main_s = (hipStream_t)stream, side = (hipStream_t)ov->side_stream;
    const int64_t ba = h.da.bs, T = d->n_tracks, n = d->n_samples;
    if (int e = (int)hipEventRecord((hipEvent_t)ov->fork_event, main_s)) return e;
    if (int e = (int)hipStreamWaitEvent(side, (hipEvent_t)ov->fork_event, 0)) return e;
    float* ws = (float*)workspace;
    // the second half first: its stream has just been released and its k_prep is the first thing the device can start beside the main stream's
    int e = console_forward_impl(&h.db, tracks + ba * T * d->track_row_stride, track_params + ba * T * MST_NUM_TRACK_PARAMS,
                                 fx_bus_params + ba * MST_NUM_FX_PARAMS, master_bus_params + ba * MST_NUM_MASTER_PARAMS, nullptr, mix + ba * 2 * n,
                                 mixed_tracks ? mixed_tracks + ba * 2 * T * n : nullptr, status, ws + h.ws_b,
                                 (size_t)(h.total - h.ws_b) * sizeof(float), side, nullptr, nullptr);
    if (!e)
        e = console_forward_impl(&h.da, tracks, track_params, fx_bus_params, master_bus_params, nullptr, mix, mixed_tracks, status, ws,
                                 (size_t)h.ws_b * sizeof(float), main_s, nullptr, nullptr);
    // always rejoin (also after an error: a captured graph must not end with the side stream forked)
    (void)hipEventRecord((hipEvent_t)ov->join_event, side);
    (void)hipStreamWaitEvent(main_s, (hipEvent_t)ov->join_event, 0);
    return e ? e : (int)hipGetLastError();
}
extern "C" int mst_console_forward_mirrored(const mst_console_desc* d, const float* tracks, const float* track_params,
                                            const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                            float* mix, float* mixed_tracks, int32_t* status, void* workspace, size_t workspace_bytes,
                                            void* stream, int32_t* status_host, void* status_event) {
    if (!status || !status_host) return hipErrorInvalidValue;
    if (d && (d->flags & MST_SPLIT_BATCH)) return hipErrorInvalidValue;  // the mirrored verdict is formed by ONE k_prep: no split form
    return console_forward_impl(d, tracks, track_params, fx_bus_params, master_bus_params, fx, mix, mixed_tracks, status, workspace,
                                workspace_bytes, stream, status_host, status_event);
}

static void allpole_scan(const Layout& L, float* ws, int nsig_all, hipStream_t stream) {
    if (L.apscan_fwd && fuse_allpole()) {
        // nothing up front: the track rows' scans rode on the forward's master-bus run, the master rows' ride on the backward's adjoint run
        (void)nsig_all;
    } else {
        launch_scan2(ws + L.zP_t, ws + L.sP_t, ws + L.powP_t, L.R, L.ncE, L.ncE_pad, L.KE, nsig_all, stream);
    }
}

extern "C" int mst_console_backward_prepare(const mst_console_desc* d, void* workspace, size_t workspace_bytes, void* stream_) {
    if (int e = check_desc(d)) return e;
    const Layout L = make_layout(d);
    if (!workspace || workspace_bytes < (size_t)L.total * sizeof(float) || ((uintptr_t)workspace & 255)) return hipErrorInvalidValue;
    if (!(d->flags & MST_SAVE_FOR_BACKWARD) || (d->flags & MST_SPLIT_BATCH)) return hipErrorInvalidValue;
    float* ws = (float*)workspace;
    const int64_t Ns = row_stride(L.N);
    const int nsig_all = L.R + ((d->flags & MST_USE_MASTER_BUS) ? 2 * L.bs : 0);
    if (!fuse_allpole()) launch_allpole_zs(ws + L.u_t, Ns, ws + L.rc_t, L.R, ws + L.zP_t, L.ncE_pad, L.N, nsig_all, (hipStream_t)stream_);
    allpole_scan(L, ws, nsig_all, (hipStream_t)stream_);
    return (int)hipGetLastError();
}

static int console_backward_impl(const mst_console_desc* d, const float* tracks, const float* track_params,
                                 const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                 const float* grad_mix, const float* grad_mixed_tracks, float* grad_track_params,
                                 float* grad_fx_params, float* grad_master_params, float* grad_tracks, int32_t* status,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
    if (int e = check_desc(d)) return e;
    const Layout L = make_layout(d);
    if (!workspace || workspace_bytes < (size_t)L.total * sizeof(float) || ((uintptr_t)workspace & 255)) return hipErrorInvalidValue;
    if (!(d->flags & MST_SAVE_FOR_BACKWARD)) return hipErrorInvalidValue;
    if (!track_params || !master_bus_params || !grad_mix || !grad_track_params || !grad_master_params) return hipErrorInvalidValue;
    hipStream_t stream = (hipStream_t)stream_;
    float* ws = (float*)workspace;
    const int64_t n = L.N, Ns = row_stride(L.N);
    const bool t_comp = d->flags & MST_USE_TRACK_COMPRESSOR;
    const bool m_on = d->flags & MST_USE_MASTER_BUS;
    const bool o_on = d->flags & MST_USE_OUTPUT_FADER;
    const bool fx_on = d->flags & MST_USE_FX_BUS;
    if (fx_on && (!fx || !fx->tables || !fx_bus_params)) return hipErrorInvalidValue;
    const int aligned = (n % 4 == 0) && !((uintptr_t)grad_mix & 15) && !((uintptr_t)grad_mixed_tracks & 15);
    if (basic_path(d)) {
        if (!tracks) return hipErrorInvalidValue;
        BasicArgs ba{tracks, track_params, fx_bus_params, master_bus_params, nullptr, nullptr, nullptr, grad_mix, grad_mixed_tracks,
                     grad_track_params, grad_master_params, grad_tracks, ws + L.cp_t, *d};
        launch_basic_backward(ba, stream);
        return (int)hipGetLastError();
    }

    // ---- all-pole states of the coefficient-gradient pass depend only on what forward saved: one
    // launch covers the track rows and the master rows (signal rows [0,R) and [R,R+2bs) of the same arrays)
    const int nsig_all = L.R + (m_on ? 2 * L.bs : 0);
    if (!(d->flags & MST_BWD_PREPARED)) {  // else: mst_console_backward_prepare ran (on a side stream the caller has joined)
        if (!fuse_allpole()) launch_allpole_zs(ws + L.u_t, Ns, ws + L.rc_t, L.R, ws + L.zP_t, L.ncE_pad, n, nsig_all, stream);
        allpole_scan(L, ws, nsig_all, stream);
    }

    // ---- master bus: compressor adjoint, EQ adjoint (-> grad of the stereo bus)
    const float* gbus = grad_mix;  // cotangent of the stereo bus as seen by the track stage
    int64_t gbus_stride = n;
    if (m_on) {
        CompBwdArgs ca{ws + L.v_m, Ns, ws + L.gs_m, ws + L.rc_m, nullptr, ws + L.zQ_m, ws + L.du_m, ws + L.cp_m,
                       grad_mix, n, nullptr, nullptr, 0, 1, L.ncC_pad, d->master_lookahead, 1, n, aligned};
        if (fuse_comp_zs()) {
            ca.gran = (gran_t*)(ws + L.gran_b) + 2 * (int64_t)L.R * L.nblkC;
            ca.gran_near = (int64_t)L.bs * L.nblkC;
            ca.status = status;
        }
        else launch_comp_bwd(true, false, ca, L.bs, stream);
        ca.s0 = ws + L.zQ_m;
        // coefficient-gradient sums of the two bus channels: in this launch (round 3), or - when the all-pole scans ride on other launches
        // (Layout::apscan_fwd) - as extra rows of the TRACKS' run launch below: the master launch is one lockstep round of lone
        // workgroups, where two more 64-sample walks per workgroup are pure latency (31 -> 18 us), the track launch absorbs them
        const bool master_cg_later = MST_FUSE_COEFGRAD && L.apscan_fwd && fuse_allpole();
        if (MST_FUSE_COEFGRAD && !master_cg_later) {
            ca.ap_s0 = ws + L.sP_m;
            ca.ap_nc_pad = L.ncE_pad;
            ca.ep = ws + L.ep_m;
        }
        launch_comp_bwd(true, true, ca, L.bs, stream);
        const float* p1A_m = L.eq1 ? ws + L.pow1A_m : nullptr;
        // round 5: the adjoint run carries its own zero-state pass when it is the SCAN1 kernel with the riders (ZsIn, mst_kernels.h)
        const bool zsin_a = L.eq1 && mfma_zs() && fuse_eq_zs() && master_cg_later;
        const ZsIn zi_a{ws + L.wzA_m, (gran_t*)(ws + L.eqg_b), L.eqg_nb, status};
        if (zsin_a) {}
        else if (L.eq1 && mfma_zs()) launch_eq_zs_mfma(EQ_ADJ, ws + L.du_m, Ns, ws + L.wzA_m, 0, ws + L.zA_m, L.ncE_pad, n, 2 * L.bs, stream, p1A_m, L.ntE, ws + L.aggA_m);
        else launch_cascade(EQ_ADJ, false, ws + L.du_m, Ns, nullptr, 0, ws + L.rc_m, 0, nullptr, ws + L.zA_m, L.ncE_pad, n, 2 * L.bs, stream, p1A_m, L.ntE, ws + L.aggA_m);
        if (!L.eq1) launch_scan12(true, ws + L.zA_m, ws + L.sA_m, ws + L.powA_m, 0, L.ncE, L.ncE_pad, L.KE, 2 * L.bs, stream);
        if (master_cg_later)  // + the master rows' own all-pole carry scans as extra workgroups of this (one wave per SIMD) launch
            launch_master_run_apscan(ws + L.du_m, Ns, ws + L.dbus, Ns, ws + L.rc_m, ws + L.zA_m, L.ncE_pad, n, 2 * L.bs, stream, p1A_m, L.ntE, ws + L.aggA_m,
                                     nullptr, ws + L.zP_m, ws + L.sP_m, ws + L.powP_m, 2 * L.bs * 12, L.ncE, L.apscan_sh, EQ_ADJ, zsin_a ? &zi_a : nullptr);
        else
            launch_cascade(EQ_ADJ, true, ws + L.du_m, Ns, ws + L.dbus, Ns, ws + L.rc_m, 0, L.eq1 ? ws + L.zA_m : ws + L.sA_m, nullptr, L.ncE_pad, n,
                           2 * L.bs, stream, p1A_m, L.ntE, ws + L.aggA_m);
        gbus = ws + L.dbus;
        gbus_stride = Ns;
    } else if (o_on) {
        CompBwdArgs ca{ws + L.bus, Ns, nullptr, ws + L.rc_m, nullptr, nullptr, ws + L.dbus, ws + L.cp_m,
                       grad_mix, n, nullptr, nullptr, 0, 1, L.ncC_pad, 0, 0, n, aligned};
        launch_comp_bwd(true, true, ca, L.bs, stream);
        gbus = ws + L.dbus;
        gbus_stride = Ns;
    } else {
        (void)hipMemsetAsync(ws + L.cp_m, 0, (size_t)L.bs * L.nblkC * CP_COUNT * sizeof(float), stream);
    }

    // ---- fx bus: the wet signal was added to the stereo bus, so its cotangent is the bus cotangent
    if (fx_on) launch_fx_backward(fx_plan(L), gbus, gbus_stride, (const float*)fx->tables, ws, stream);

    // ---- tracks
    {
        CompBwdArgs ca{ws + L.u_t, Ns, ws + L.gs_t, ws + L.rc_t, nullptr, ws + L.zQ_t, ws + L.du_t, ws + L.cp_t,
                       gbus, gbus_stride, grad_mixed_tracks, fx_on ? ws + L.fx_din : nullptr, Ns, L.T, L.ncC_pad, d->track_lookahead,
                       t_comp ? 1 : 0, n, aligned};
        if (t_comp) {
            if (fuse_comp_zs()) {
                ca.gran = (gran_t*)(ws + L.gran_b);
                ca.gran_near = (int64_t)L.R * L.nblkC;
                ca.status = status;
            }
            else launch_comp_bwd(false, false, ca, L.R, stream);
            ca.s0 = ws + L.zQ_t;
        }
        if (MST_FUSE_COEFGRAD) {
            // the run pass forms the tracks' coefficient-gradient sums from the du and u it holds in registers: du crosses HBM only
            // when the EQ adjoint below needs it (grad_tracks), u is not read a second time
            ca.ap_s0 = ws + L.sP_t;
            ca.ap_nc_pad = L.ncE_pad;
            ca.ep = ws + L.ep_t;
            if (!grad_tracks) ca.du = nullptr;
            if (m_on && L.apscan_fwd && fuse_allpole()) {  // the master channels' coefficient-gradient walks ride here (see above)
                ca.cg2_u = ws + L.v_m;
                ca.cg2_du = ws + L.du_m;
                ca.cg2_rc = ws + L.rc_m;
                ca.cg2_rows = 2 * L.bs;
            }
        }
        launch_comp_bwd(false, true, ca, L.R, stream);
        // without the fusion: one k_coefgrad launch for the track rows and the master rows (which follow the tracks in the same arrays)
        if (!MST_FUSE_COEFGRAD) launch_coefgrad(ws + L.u_t, Ns, ws + L.du_t, Ns, ws + L.rc_t, L.R, ws + L.sP_t, L.ncE_pad, ws + L.ep_t, n, nsig_all, stream);
        if (grad_tracks) {
            const float* p1A_t = L.eq1 ? ws + L.pow1A_t : nullptr;
            if (L.eq1 && mfma_zs()) launch_eq_zs_mfma(EQ_ADJ, ws + L.du_t, Ns, ws + L.wzA_t, L.R, ws + L.zA_t, L.ncE_pad, n, L.R, stream, p1A_t, L.ntE, ws + L.aggA_t);
            else launch_cascade(EQ_ADJ, false, ws + L.du_t, Ns, nullptr, 0, ws + L.rc_t, L.R, nullptr, ws + L.zA_t, L.ncE_pad, n, L.R, stream, p1A_t, L.ntE, ws + L.aggA_t);
            if (!L.eq1) launch_scan12(true, ws + L.zA_t, ws + L.sA_t, ws + L.powA_t, L.R, L.ncE, L.ncE_pad, L.KE, L.R, stream);
            launch_cascade(EQ_ADJ, true, ws + L.du_t, Ns, grad_tracks, n, ws + L.rc_t, L.R, L.eq1 ? ws + L.zA_t : ws + L.sA_t, nullptr, L.ncE_pad,
                           n, L.R, stream, p1A_t, L.ntE, ws + L.aggA_t);
        }
    }
    PrepBwdArgs pb{track_params, master_bus_params, ws + L.rc_t, ws + L.rc_m, ws + L.cp_t, ws + L.cp_m, ws + L.ep_t, ws + L.ep_m,
                   grad_track_params, grad_master_params, fx_bus_params, fx_on ? ws + L.fx_part : nullptr,
                   fx_on ? grad_fx_params : nullptr, L.fxBlkIr, fx_on ? ws + L.fx_mix : nullptr, fx_on ? ws + L.fx_dry : nullptr, L.fxBlk, L.R, L.bs, L.nblkC, L.nblkE, L.nblkEt, *d,
                   (gran_t*)(ws + L.gran_b), L.gran_nb};
    launch_prep_bwd(pb, stream);
    return (int)hipGetLastError();
}

extern "C" int mst_console_backward(const mst_console_desc* d, const float* tracks, const float* track_params,
                                    const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                    const float* grad_mix, const float* grad_mixed_tracks, float* grad_track_params,
                                    float* grad_fx_params, float* grad_master_params, float* grad_tracks, int32_t* status,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (d && (d->flags & MST_SPLIT_BATCH)) return hipErrorInvalidValue;  // mst_console_backward_overlapped
    return console_backward_impl(d, tracks, track_params, fx_bus_params, master_bus_params, fx, grad_mix, grad_mixed_tracks, grad_track_params,
                                 grad_fx_params, grad_master_params, grad_tracks, status, workspace, workspace_bytes, stream);
}
extern "C" int mst_console_backward_overlapped(const mst_console_desc* d, const float* tracks, const float* track_params,
                                               const float* fx_bus_params, const float* master_bus_params, const mst_console_fx* fx,
                                               const float* grad_mix, const float* grad_mixed_tracks, float* grad_track_params,
                                               float* grad_fx_params, float* grad_master_params, float* grad_tracks, int32_t* status,
                                               void* workspace, size_t workspace_bytes, void* stream, const mst_console_overlap* ov) {
    if (int e = check_desc(d)) return e;
    if (!split_on(d))
        return console_backward_impl(d, tracks, track_params, fx_bus_params, master_bus_params, fx, grad_mix, grad_mixed_tracks, grad_track_params,
                                     grad_fx_params, grad_master_params, grad_tracks, status, workspace, workspace_bytes, stream);
    if (!overlap_ok(ov) || ov->side_stream == stream || (d->flags & MST_BWD_PREPARED)) return hipErrorInvalidValue;
    const Halves h = make_halves(d);
    if (!workspace || workspace_bytes < (size_t)h.total * sizeof(float) || ((uintptr_t)workspace & 255)) return hipErrorInvalidValue;
    if (!track_params || !master_bus_params || !grad_mix || !grad_track_params || !grad_master_params) return hipErrorInvalidValue;
    hipStream_t main_s = (hipStream_t)stream, side = (hipStream_t)ov->side_stream;
    const int64_t ba = h.da.bs, T = d->n_tracks, n = d->n_samples;
    if (int e = (int)hipEventRecord((hipEvent_t)ov->fork_event, main_s)) return e;
    if (int e = (int)hipStreamWaitEvent(side, (hipEvent_t)ov->fork_event, 0)) return e;
    float* ws = (float*)workspace;
    int e = console_backward_impl(&h.db, tracks ? tracks + ba * T * d->track_row_stride : nullptr, track_params + ba * T * MST_NUM_TRACK_PARAMS,
                                  fx_bus_params ? fx_bus_params + ba * MST_NUM_FX_PARAMS : nullptr, master_bus_params + ba * MST_NUM_MASTER_PARAMS, nullptr,
                                  grad_mix + ba * 2 * n, grad_mixed_tracks ? grad_mixed_tracks + ba * 2 * T * n : nullptr,
                                  grad_track_params + ba * T * MST_NUM_TRACK_PARAMS, nullptr, grad_master_params + ba * MST_NUM_MASTER_PARAMS,
                                  grad_tracks ? grad_tracks + ba * T * n : nullptr, status, ws + h.ws_b, (size_t)(h.total - h.ws_b) * sizeof(float), side);
    if (!e)
        e = console_backward_impl(&h.da, tracks, track_params, fx_bus_params, master_bus_params, nullptr, grad_mix, grad_mixed_tracks, grad_track_params,
                                  nullptr, grad_master_params, grad_tracks, status, ws, (size_t)h.ws_b * sizeof(float), main_s);
    (void)hipEventRecord((hipEvent_t)ov->join_event, side);
    (void)hipStreamWaitEvent(main_s, (hipEvent_t)ov->join_event, 0);
    return e ? e : (int)hipGetLastError();
}
