// scan_bwd_bf16_f32.hip — instantiates the selective-scan backward kernels for io=cm_bf16, B/C=float.
#include "scan_bwd_impl.h"
int cm_scan_bwd_bf16_f32(const cm_scan_bwd_args &a, int S, bool vecok) { return cm_scan_bwd_dispatch<cm_bf16, float>(a, S, vecok); }
