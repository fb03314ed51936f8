// scan_bwd_f32.hip — instantiates the selective-scan backward kernels for io=float, B/C=float.
#include "scan_bwd_impl.h"
int cm_scan_bwd_f32(const cm_scan_bwd_args &a, int S, bool vecok) { return cm_scan_bwd_dispatch<float, float>(a, S, vecok); }
