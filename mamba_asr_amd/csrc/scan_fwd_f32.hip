// scan_fwd_f32.hip — instantiates the selective-scan forward kernels for io=float, B/C=float.
#include "scan_fwd_impl.h"
int cm_scan_fwd_f32(const cm_scan_fwd_args &a, int S, bool vecok) { return cm_scan_fwd_dispatch<float, float>(a, S, vecok); }
