// scan_fwd_bf16.hip — instantiates the selective-scan forward kernels for io=cm_bf16, B/C=cm_bf16.
#include "scan_fwd_impl.h"
int cm_scan_fwd_bf16(const cm_scan_fwd_args &a, int S, bool vecok) { return cm_scan_fwd_dispatch<cm_bf16, cm_bf16>(a, S, vecok); }
