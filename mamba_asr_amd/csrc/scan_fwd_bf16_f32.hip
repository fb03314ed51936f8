// scan_fwd_bf16_f32.hip — instantiates the selective-scan forward kernels for io=cm_bf16, B/C=float.
#include "scan_fwd_impl.h"
int cm_scan_fwd_bf16_f32(const cm_scan_fwd_args &a, int S, bool vecok) { return cm_scan_fwd_dispatch<cm_bf16, float>(a, S, vecok); }
