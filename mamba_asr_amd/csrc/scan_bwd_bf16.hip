// scan_bwd_bf16.hip — instantiates the selective-scan backward kernels for io=cm_bf16, B/C=cm_bf16.
#include "scan_bwd_impl.h"
int cm_scan_bwd_bf16(const cm_scan_bwd_args &a, int S, bool vecok) { return cm_scan_bwd_dispatch<cm_bf16, cm_bf16>(a, S, vecok); }
