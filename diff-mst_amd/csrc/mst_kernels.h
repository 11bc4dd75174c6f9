// mst_kernels.h - argument blocks and host-side launch helpers shared by the console translation
// units (each kernel is launched only from the file that defines it: no relocatable device code).
#pragma once
#include "mst_common.h"

namespace mst {

enum { EQ_FWD = 0, EQ_ADJ = 1 };
// `split` arguments: signal rows below it are mono (tracks, one filter row each), rows from it on are stereo
// pairs sharing a filter row (master buses).  Tracks-only launch: split = nsig; master-only: split = 0.

// ---- mst_params.hip
struct PrepArgs {
    const float* track_params;   // (R, 27)
    const float* fx_params;      // (bs, 25)
    const float* master_params;  // (bs, 26)
    float* rc_t;
    float* rc_m;
    float* powF_t; float* powF_m;  // forward-cascade scan tables
    float* powA_t; float* powA_m;  // adjoint-cascade scan tables
    float* powP_t; float* powP_m;  // all-pole scan tables
    float* pow1F_t; float* pow1F_m;  // in-wave scan tables (forward / adjoint cascade)
    float* pow1A_t; float* pow1A_m;
    float* wzF_t; float* wzF_m;      // zero-state maps of the forward / adjoint cascade (eq1 only)
    float* wzA_t; float* wzA_m;
    float* rc_fx;   // (bs, 24): reverberation band gains (times the wet/dry mix) and decay rates 10 d + 1 (fx bus), or nullptr
    float* fx_mix;  // (bs): wet/dry mix - 1 (reference mst/modules.py:420) unless MST_NO_RANGE_CHECK hands a value over
    int32_t* status;
    int R, bs;
    int KE;  // chunks per scan lane (EQ scans)
    int eq1; // which family of cascade tables to build (Layout::eq1)
    mst_console_desc d;
    gran_t* gran;       // every granule array of the call (mst_common.h), zeroed here
    int64_t gran_n;
    // prefetch riders (mst_params.hip: k_prep): R rows of pf_n floats, pf_stride apart, pulled through the Infinity Cache while the
    // design chains run; null = none.  Rows must be 16-byte aligned with pf_n % 4 == 0.
    const float* pf_src;
    int64_t pf_stride, pf_n;
};
struct PrepBwdArgs {
    const float* track_params;
    const float* master_params;
    const float* rc_t;
    const float* rc_m;
    const float* cp_t; const float* cp_m;  // compressor partial sums  rows x nblkC x CP_COUNT
    const float* ep_t; const float* ep_m;  // coefficient partial sums sigrows x nblkE x EP_COUNT
    float* grad_track_params;              // (R,27)
    float* grad_master_params;             // (bs,26)
    const float* fx_params;                // (bs,25) fx bus only
    const float* fx_part;                  // (bs, nblkF, 24) partial sums of the reverberation parameters, or nullptr
    float* grad_fx_params;                 // (bs,25) or nullptr
    int nblkF;
    const float* fx_mix;                   // (bs) wet/dry mix used by forward
    const float* fx_dry;                   // (bs, nblkX) partial sums <dbus, fx_in>
    int nblkX;
    int R, bs, nblkC, nblkE;
    int nblkEt;                            // partial rows per track row (Layout::nblkEt)
    mst_console_desc d;
    gran_t* gran;                          // the backward's granule arrays, re-armed (zeroed) for a second backward over the same forward
    int64_t gran_n;
};
// BASELINE cfg #1 (gain + pan + bus sum only): one forward launch, two backward launches, no k_prep chain (mst_params.hip)
constexpr int kBasicMaxTracks = 256;
struct BasicArgs {
    const float* tracks;         // (bs, T, n), row stride d.track_row_stride
    const float* track_params;   // (bs, T, 27)
    const float* fx_params;      // (bs, 25)   range check only
    const float* master_params;  // (bs, 26)   range check only
    float* mix;                  // (bs, 2, n) forward out
    float* mixed;                // (bs, 2, T, n) forward out or null
    int32_t* status;
    const float* grad_mix;       // backward in
    const float* grad_mixed;     // (bs, 2, T, n) or null
    float* grad_track_params;    // (bs, T, 27) out
    float* grad_master_params;   // (bs, 26) out (zeros) or null
    float* grad_tracks;          // (bs, T, n) out or null
    float* part;                 // (bs, nblk, T, 2) scratch
    mst_console_desc d;
};
inline bool basic_path(const mst_console_desc* d) {  // every stage but input fader / panner off, and few enough tracks for the LDS table
    const uint32_t stages = MST_USE_TRACK_EQ | MST_USE_TRACK_COMPRESSOR | MST_USE_FX_BUS | MST_USE_MASTER_BUS | MST_USE_OUTPUT_FADER;
#ifdef MST_NO_BASIC_PATH
    return false;
#else
    return !(d->flags & stages) && (d->flags & MST_USE_TRACK_PANNER) && d->n_tracks <= kBasicMaxTracks && !(d->flags & MST_DEV_MULTIPASS_EQ);
#endif
}
void launch_basic_forward(const BasicArgs& a, hipStream_t stream);
void launch_basic_backward(const BasicArgs& a, hipStream_t stream);
void launch_prep(const PrepArgs& a, hipStream_t stream);
void launch_prep_bwd(const PrepBwdArgs& a, hipStream_t stream);

// ---- mst_eq.hip
// The zero-state pass of a SCAN1 run INSIDE the run launch (round 5; wz = nullptr: off - a zs launch went before).  Every tile forms
// its zero-state chunk end states on the matrix pipe itself, publishes its 12-state aggregate as granules and picks up the aggregates
// of the tiles before it (mst_common.h: gran_publish_vec / gran_read_vec).
struct ZsIn {
    const float* wz;    // zero-state maps, filter rows x 64 x 16 (k_prep)
    gran_t* gran;       // (signal rows, kMaxTiles1, 12) zeroed granules, indexed by the tile's position in recurrence order
    int64_t gran_near;  // the near copies follow this many granules later
    int32_t* status;    // raised to kStatusExchangeTimeout when a wait gives up (may be null)
};
// zp != nullptr (forward run only): the all-pole bank of the coefficient-gradient pass rides along, its zero-state chunk
// end states are written to zp (nsig x 24 x nc_pad) and the backward needs no k_allpole_zs launch.
// pw1 (in-wave scan tables of these rows) != nullptr: SCAN1 kernels, no carry-scan launch between zs and run (mst_eq.hip);
// agg (nsig x 12 x kMaxTiles1) carries the tile aggregates from the zs launch to the run launch
void launch_cascade(int dir, bool run, const float* in, int64_t in_stride, float* out, int64_t out_stride, const float* rc,
                    int split, const float* s0, float* z, int nc_pad, int64_t n, int nsig, hipStream_t stream,
                    const float* pw1 = nullptr, int ntiles = 0, float* agg = nullptr, float* zp = nullptr);
// forward run of mono rows fused with the compressor's zero-state block aggregates (replaces k_comp_zs<1>)
void launch_cascade_run_gc(const float* in, int64_t in_stride, float* out, int64_t out_stride, const float* rc, int split,
                           const float* s0, int nc_pad, int64_t n, int nsig, float* zs_comp, int nblk_comp, hipStream_t stream,
                           const float* pw1 = nullptr, int ntiles = 0, float* agg = nullptr, float* zp = nullptr, const ZsIn* zi = nullptr);
// zero-state pass of the SCAN1 path on the matrix pipe: chunk end states = W^T chunk (wz: filter rows x 64 x 16, made by k_prep)
void launch_eq_zs_mfma(int dir, const float* in, int64_t in_stride, const float* wz, int split, float* z, int nc_pad, int64_t n, int nsig,
                       hipStream_t stream, const float* pw1, int ntiles, float* agg);
// the master-bus forward run (SCAN1, all-pole bank riding along) with the TRACK rows' all-pole carry scan as extra one-wave
// workgroups of the same launch (mst_eq.hip: k_master_run_apscan); sc_sh: 64 = KE 2^sc_sh
void launch_master_run_apscan(const float* in, int64_t in_stride, float* out, int64_t out_stride, const float* rc, const float* s0, int nc_pad,
                              int64_t n, int nsig, hipStream_t stream, const float* pw1, int ntiles, float* agg, float* zp,
                              const float* sc_z, float* sc_s0, const float* sc_tab, int sc_jobs, int sc_nc, int sc_sh, int dir = EQ_FWD,
                              const ZsIn* zi = nullptr);  // zi (wz != null): the master rows' zero-state pass runs inside the launch too (no k_eq_zs_mfma before it)
void launch_allpole_zs(const float* u, int64_t u_stride, const float* rc, int split, float* z, int nc_pad, int64_t n, int nsig,
                       hipStream_t stream);
void launch_coefgrad(const float* u, int64_t u_stride, const float* g, int64_t g_stride, const float* rc, int split,
                     const float* s0, int nc_pad, float* part, int64_t n, int nsig, hipStream_t stream);

// ---- mst_scan.hip
void launch_scan12(bool reverse, const float* z, float* s0, const float* tab, int split, int nc, int nc_pad, int K, int nsig,
                   hipStream_t stream);
void launch_scan2(const float* z, float* s0, const float* tab, int split, int nc, int nc_pad, int K, int nsig, hipStream_t stream);

// ---- mst_comp.hip
struct TrackApplyArgs {
    const float* u;      // (R, stride) EQ output
    int64_t stride;
    const float* rc;     // (R, RC_STRIDE)
    const float* s0;     // (R, nc_pad) smoother start states
    float* gs;           // (R, stride) out: smoothed gain reduction in dB (saved for backward), may be null
    float* bus;          // (bs, 2, bus_stride) out
    int64_t bus_stride;
    float* mixed;        // (bs, 2, T, n) out or null
    float* fx;           // (bs, 2, bus_stride) out or null: the fx send bus sum_t send_t * mixed_t (reference stereo_bus)
    int T, nc_pad, lookahead, comp_on;
    int64_t n;
    int aligned;         // every row base / stride is 16-byte aligned: interior blocks skip all guards
};
struct MasterApplyArgs {
    const float* v;      // (bs*2, stride) master EQ output (or the raw bus when the master bus is off)
    int64_t stride;
    const float* rc;     // (bs, RC_STRIDE)
    const float* s0;     // (bs, nc_pad)
    float* gs;           // (bs, stride) or null
    float* out;          // (bs, 2, out_stride)
    int64_t out_stride;
    int nc_pad, lookahead, comp_on;
    int64_t n;
    int aligned;
    gran_t* gran;        // (bs, nblk) zeroed granules (+ their near copies gran_near granules later): the smoother's block aggregates are exchanged inside this launch (no k_comp_zs); null = read s0
    int64_t gran_near;
    int32_t* status;     // raised to kStatusExchangeTimeout when an exchange wait gives up (may be null)
};
// One argument block for tracks (NCH = 1) and master (NCH = 2).
struct CompBwdArgs {
    const float* u;       // (rows*NCH, stride)  compressor input (EQ output)
    int64_t stride;
    const float* gs;      // (rows, stride) saved smoothed gain (dB)
    const float* rc;      // (rows, RC_STRIDE)
    const float* s0;      // run pass: adjoint smoother state entering each chunk from the right
    float* zq;            // (rows, nc_pad) out of the zs pass
    float* du;            // (rows*NCH, stride) out of the run pass: cotangent of the compressor input
    float* part;          // (rows, nblk, CP_COUNT) out of the run pass
    const float* gup;     // tracks: grad wrt stereo bus ; master: grad wrt mix ; (bs,2,gup_stride)
    int64_t gup_stride;
    const float* gmixed;  // tracks only: grad wrt mixed_tracks (bs,2,T,n) or null
    const float* gfx;     // tracks only: grad wrt the fx send bus (bs,2,gfx_stride) or null
    int64_t gfx_stride;
    int T, nc_pad, lookahead, comp_on;
    int64_t n;
    int aligned;
    // tracks, MST_FUSE_COEFGRAD: the run pass also forms the coefficient-gradient sums of its 2048 samples (else ep = null)
    const float* ap_s0;   // all-pole states entering every 64-sample chunk (rows, 24, ap_nc_pad)
    int ap_nc_pad;
    float* ep;            // (rows, nblkC, EP_COUNT)
    // tracks launch only: extra signal rows whose coefficient-gradient sums ride along (the two channels of every master bus: their
    // compressor adjoint ran in the master launch, all that is left is the all-pole walk over u2 = master EQ output and du2 = its cotangent)
    const float* cg2_u;   // (cg2_rows, stride)
    const float* cg2_du;  // (cg2_rows, stride)
    const float* cg2_rc;  // (cg2_rows / 2, RC_STRIDE) filter rows
    int cg2_rows;         // 0: none; signal row of extra row j = main rows + j (all-pole states, partial sums)
    gran_t* gran;         // (rows, nblk) zeroed granules (+ near copies gran_near granules later): the run pass publishes / awaits the block aggregates itself (no zs launch); null = read s0
    int64_t gran_near;
    int32_t* status;      // raised to kStatusExchangeTimeout when an exchange wait gives up (may be null)
    int mw_split;         // k_comp_bwd_mix only (set by launch_comp_bwd): workgroups the tracks of one mix are dealt to (1 or 2)
};
void launch_comp_zs(int nch, const float* u, int64_t stride, const float* rc, float* zs, int nc_pad, int64_t n, int rows,
                    hipStream_t stream);
void launch_apply_tracks(const TrackApplyArgs& a, int bs, hipStream_t stream);
void launch_apply_master(const MasterApplyArgs& a, int bs, hipStream_t stream);
void launch_comp_bwd(bool master, bool run, const CompBwdArgs& a, int rows, hipStream_t stream);

// ---- mst_fx.hip: the fx bus (noise-shaped reverberation on a send bus); offsets are float offsets into the workspace
struct FxPlan {
    int bs, S, taps, K, nblk, nblk_ir;
    int64_t n, Ns;
    int64_t rcfx, fx_in, wnf, ir, Xs, Hs, Ys, dXs, dHs, dir, dfx_in, fxpart, Hf, mixv, dry;
};
void launch_fx_forward(const FxPlan& p, const float* noise, const float* filters, const float* tables, float* ws, float* bus,
                       int64_t bus_stride, hipStream_t stream);
void launch_fx_backward(const FxPlan& p, const float* dbus, int64_t dbus_stride, const float* tables, float* ws, hipStream_t stream);
void launch_fx_tables(float* tables, hipStream_t stream);

}  // namespace mst
