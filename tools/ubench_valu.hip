// ubench_valu.hip — VALU issue-rate microbenchmark for gfx950 (design input for the scan kernel):
// cycles per wave-instruction for v_fma_f32, v_pk_fma_f32, v_exp_f32 and v_add_f32 dpp, at 1/2/4
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 0.999f, c = 1e-3f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent v_fma_f32
            a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
            a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
        } else if (MODE == 1) {   // 4 independent v_pk_fma_f32 (same flops as 8 fma)
            const float2_ mm = {m, m}, cc = {c, c};
            p0 = __builtin_elementwise_fma(p0, mm, cc); p1 = __builtin_elementwise_fma(p1, mm, cc);
            p2 = __builtin_elementwise_fma(p2, mm, cc); p3 = __builtin_elementwise_fma(p3, mm, cc);
        } else if (MODE == 2) {   // 8 independent v_exp_f32
            a0 = __builtin_amdgcn_exp2f(a0 * m); a1 = __builtin_amdgcn_exp2f(a1 * m); a2 = __builtin_amdgcn_exp2f(a2 * m); a3 = __builtin_amdgcn_exp2f(a3 * m);
            a4 = __builtin_amdgcn_exp2f(a4 * m); a5 = __builtin_amdgcn_exp2f(a5 * m); a6 = __builtin_amdgcn_exp2f(a6 * m); a7 = __builtin_amdgcn_exp2f(a7 * m);
        } else {   // 8 dpp adds
#define DPPADD(x) x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true))
            DPPADD(a0); DPPADD(a1); DPPADD(a2); DPPADD(a3); DPPADD(a4); DPPADD(a5); DPPADD(a6); DPPADD(a7);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}

int main() {
    float *d;
    hipMalloc(&d, 64);
    const int iters = 20000;
    const char *names[] = {"8x v_fma_f32", "4x v_pk_fma_f32", "8x (v_mul+v_exp_f32)", "8x v_add_f32 dpp"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: block = wps*4 waves, one block per CU
            dim3 grid(256), block(64 * 4 * wps > 1024 ? 1024 : 64 * 4 * wps);
            int g = 64 * 4 * wps > 1024 ? 256 * (64 * 4 * wps / 1024) : 256;
            grid = dim3(g);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters, 1.0f); break;
                    case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters, 1.0f); break;
                    case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters, 1.0f); break;
                    default: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, iters, 1.0f); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
            double cyc_per_iter = h[1] / iters;     // shader cycles (s_memtime-based counter runs at 100 MHz? report both)
            printf("%-22s waves/SIMD=%d  wall=%.3f ms  counter/iter=%.2f  ns/iter/wave=%.2f\n", names[mode], wps, ms,
                   cyc_per_iter, ms * 1e6 / iters);
        }
    }
    return 0;
}
