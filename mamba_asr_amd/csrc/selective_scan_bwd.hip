// selective_scan_bwd.hip — C-ABI entry of the selective-scan backward.  Kernel: scan_bwd_impl.h.
#include "cm_common.h"

int cm_scan_bwd_f32(const cm_scan_bwd_args &a, int S, bool vecok);
int cm_scan_bwd_bf16(const cm_scan_bwd_args &a, int S, bool vecok);
int cm_scan_bwd_bf16_f32(const cm_scan_bwd_args &a, int S, bool vecok);
int cm_scan_pick_split(int batch, int dim, int dstate, int want);

// lane split of the backward kernels (shared by the launch and the workspace query)
static int bwd_split(const cm_scan_fwd_args &f) {
    // backward kernels exist for lane splits 4, 8, 16.  8 lanes per channel (2 states per lane) is the widest split that
    // still fits two waves per SIMD and beats 4 lanes (one wave per SIMD, 390 VGPRs) at every batch measured: 526 vs
    // 572 us at 32 x 512 channels, 379 vs 547 us at 16; an explicit lanes_per_channel = 4 still selects 4.
    int S = cm_scan_pick_split(f.batch, f.dim, f.dstate, f.lanes_per_channel);
    if (S < 4) S = 4;
    if (S < 8 && f.lanes_per_channel == 0) S = 8;
    if (S > f.dstate) S = f.dstate;
    return S;
}

extern "C" int64_t cm_selective_scan_bwd_workspace_bytes(const cm_scan_bwd_args *args) {
    if (!args) return 0;
    const cm_scan_fwd_args &f = args->fwd;
    if (f.batch <= 0 || f.dim <= 0 || f.seqlen <= 0 || f.dstate <= 0) return 0;
    const int S = bwd_split(f), chans = 4 * (64 / S);            // channels per workgroup (4 waves)
    const int64_t gx = (f.dim + chans - 1) / chans;
    return 4 * (2 * gx * f.batch * f.dstate * f.seqlen + (int64_t)f.batch * f.dim * f.dstate + 2 * (int64_t)f.batch * f.dim);
}

extern "C" int cm_selective_scan_bwd(const cm_scan_bwd_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "scan_bwd: args is NULL");
    const cm_scan_bwd_args &a = *args;
    const cm_scan_fwd_args &f = a.fwd;
    CM_REQUIRE(f.batch > 0 && f.dim > 0 && f.seqlen > 0 && f.dstate > 0, CM_EINVAL,
               "scan_bwd: bad sizes batch=%d dim=%d seqlen=%d dstate=%d", f.batch, f.dim, f.seqlen, f.dstate);
    CM_REQUIRE(f.batch <= 65535, CM_EINVAL, "scan_bwd: batch %d exceeds the grid limit 65535", f.batch);
    CM_REQUIRE(f.u && f.delta && f.A && f.B && f.C && f.x && a.dout, CM_EINVAL,
               "scan_bwd: u/delta/A/B/C/x/dout must be non-NULL");
    CM_REQUIRE(a.du && a.ddelta && a.dA && a.dB && a.dC, CM_EINVAL, "scan_bwd: du/ddelta/dA/dB/dC must be non-NULL");
    CM_REQUIRE(!f.z || a.dz, CM_EINVAL, "scan_bwd: dz is NULL although z is given");
    CM_REQUIRE(f.lanes_per_channel >= 0 && f.lanes_per_channel <= 16 && (f.lanes_per_channel & (f.lanes_per_channel - 1)) == 0, CM_EINVAL,
               "scan_bwd: lanes_per_channel %d (0, 4, 8 or 16)", f.lanes_per_channel);
    CM_REQUIRE(f.h0 == nullptr, CM_EUNSUPPORTED, "scan_bwd: an initial state (h0) is a forward-only feature");
    const int vec = f.io_dtype == CM_F32 ? 4 : 8;
    auto rows_ok = [&](const void *p, int64_t s0, int64_t s1) {
        return !p || (cm_aligned(p, 16) && s0 % vec == 0 && s1 % vec == 0);
    };
    const bool vecok = f.seqlen % vec == 0 && rows_ok(f.u, f.u_bs, f.u_ds) && rows_ok(f.delta, f.delta_bs, f.delta_ds) &&
                       rows_ok(f.z, f.z_bs, f.z_ds) && rows_ok(a.dout, a.dout_bs, a.dout_ds) &&
                       rows_ok(a.du, a.du_bs, a.du_ds) && rows_ok(a.ddelta, a.ddelta_bs, a.ddelta_ds) &&
                       rows_ok(a.dz, a.dz_bs, a.dz_ds) && rows_ok(f.z ? f.out_z : nullptr, f.out_bs, f.out_ds);
    CM_REQUIRE((a.workspace == nullptr) == (a.workspace_bytes == 0) && (!a.workspace || cm_aligned(a.workspace, 16)), CM_EINVAL,
               "scan_bwd: workspace must be 16-byte aligned with workspace_bytes > 0, or NULL with 0");
    const int S = bwd_split(f);
    switch (f.io_dtype * 4 + f.bc_dtype) {
        case CM_F32 * 4 + CM_F32: return cm_scan_bwd_f32(a, S, vecok);
        case CM_BF16 * 4 + CM_BF16: return cm_scan_bwd_bf16(a, S, vecok);
        case CM_BF16 * 4 + CM_F32: return cm_scan_bwd_bf16_f32(a, S, vecok);
        default:
            cm_set_error("scan_bwd: unsupported dtype pair io=%d bc=%d (built: f32/f32, bf16/bf16, bf16/f32)",
                         f.io_dtype, f.bc_dtype);
            return CM_EUNSUPPORTED;
    }
}
