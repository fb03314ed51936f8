// selective_scan_fwd.hip — C-ABI entry of the selective-scan forward (argument checks, lane-split
// choice, dtype dispatch).  Kernel: scan_fwd_impl.h.
#include "cm_common.h"

int cm_scan_fwd_f32(const cm_scan_fwd_args &a, int S, bool vecok);
int cm_scan_fwd_bf16(const cm_scan_fwd_args &a, int S, bool vecok);
int cm_scan_fwd_bf16_f32(const cm_scan_fwd_args &a, int S, bool vecok);

// Lane split: smallest S (fewest redundant per-channel ops) whose grid still gives about one
// wave per SIMD (1024 waves on MI355X); callers run the two BiMamba directions concurrently.
// `want` = the caller's cm_scan_fwd_args.lanes_per_channel (a power of two), 0 = choose here.
int cm_scan_pick_split(int batch, int dim, int dstate, int want) {
    const int smax = dstate < 16 ? dstate : 16;
    if (want >= 1 && (want & (want - 1)) == 0) return want <= smax ? want : smax;
    int S = 1;
    while (S < smax) {
        const long waves = (long)batch * ((dim + 64 / S - 1) / (64 / S));
        if (waves >= 1024) break;
        S *= 2;
    }
    return S;
}

extern "C" int cm_selective_scan_fwd(const cm_scan_fwd_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "scan_fwd: args is NULL");
    const cm_scan_fwd_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.dim > 0 && a.seqlen > 0 && a.dstate > 0, CM_EINVAL,
               "scan_fwd: bad sizes batch=%d dim=%d seqlen=%d dstate=%d", a.batch, a.dim, a.seqlen, a.dstate);
    CM_REQUIRE(a.batch <= 65535, CM_EINVAL, "scan_fwd: batch %d exceeds the grid limit 65535", a.batch);
    CM_REQUIRE(a.u && a.delta && a.A && a.B && a.C, CM_EINVAL, "scan_fwd: u/delta/A/B/C must be non-NULL");
    CM_REQUIRE(a.lanes_per_channel >= 0 && a.lanes_per_channel <= 16 && (a.lanes_per_channel & (a.lanes_per_channel - 1)) == 0, CM_EINVAL,
               "scan_fwd: lanes_per_channel %d (0, 1, 2, 4, 8 or 16)", a.lanes_per_channel);
    CM_REQUIRE(a.z ? a.out_z != nullptr : a.out != nullptr, CM_EINVAL,
               "scan_fwd: %s output pointer is NULL", a.z ? "out_z" : "out");
    const int vec = a.io_dtype == CM_F32 ? 4 : 8;
    const int bvec = a.bc_dtype == CM_F32 ? 4 : 8;
    auto rows_ok = [](const void *p, int64_t s0, int64_t s1, int v) {
        return !p || (cm_aligned(p, 16) && s0 % v == 0 && s1 % v == 0);
    };
    // fast path: every row starts 16-byte aligned and holds whole vectors
    const bool vecok = a.seqlen % vec == 0 && a.seqlen % bvec == 0 &&
                       rows_ok(a.u, a.u_bs, a.u_ds, vec) && rows_ok(a.delta, a.delta_bs, a.delta_ds, vec) &&
                       rows_ok(a.z, a.z_bs, a.z_ds, vec) && rows_ok(a.out, a.out_bs, a.out_ds, vec) &&
                       rows_ok(a.out_z, a.out_bs, a.out_ds, vec) && rows_ok(a.B, a.B_bs, a.B_ns, bvec) &&
                       rows_ok(a.C, a.C_bs, a.C_ns, bvec);
    int S = vecok ? cm_scan_pick_split(a.batch, a.dim, a.dstate, a.lanes_per_channel) : 4;
    switch (a.io_dtype * 4 + a.bc_dtype) {
        case CM_F32 * 4 + CM_F32: return cm_scan_fwd_f32(a, S, vecok);
        case CM_BF16 * 4 + CM_BF16: return cm_scan_fwd_bf16(a, S, vecok);
        case CM_BF16 * 4 + CM_F32: return cm_scan_fwd_bf16_f32(a, S, vecok);
        default:
            cm_set_error("scan_fwd: unsupported dtype pair io=%d bc=%d (built: f32/f32, bf16/bf16, bf16/f32)",
                         a.io_dtype, a.bc_dtype);
            return CM_EUNSUPPORTED;
    }
}
