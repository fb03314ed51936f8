// pad_rows.hip — reflect padding of the time and frequency axes of a channels-last (batch, time, freq, channel) tensor, forward and
// backward (contract: cm_reflect_pad_tf in include/conmamba_hip.h).  The reference's front end is speechbrain's
// ConvolutionFrontEnd: every Conv2d pads 'same' in reflect mode before a stride-2 3x3 convolution (hparams/CTC/conmamba_large.yaml:
// 187-194); in training the padded copy of block 2's input is 344 MB per 32 x 40 s micro-batch, and torch's 3-d reflection pad moved it
// at 1.5 TB/s forward and 0.8 TB/s backward (448 / 851 us).
//   forward : y[b, to, fo, :] = x[b, r(to - p, T), r(fo - p, F), :],  r(i, n) = -i for i < 0, 2 (n - 1) - i for i >= n, else i
//   backward: dx[b, t, f, :]  = sum of dy over the (at most 2 x 2) padded positions that read (t, f)
// One thread = one 16-byte (or, for channel rows that are not a multiple of 16 bytes, one element-sized) piece of a channel row.
#include "cm_common.h"

namespace {

__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// positions of the padded axis (length n + 2 p) that read source index i: i + p, and the mirror images near either end
__device__ __forceinline__ int sources(int i, int n, int p, int (&out)[3]) {
    int k = 0;
    out[k++] = i + p;
    if (i >= 1 && i <= p) out[k++] = p - i;
    if (i <= n - 2 && i >= n - 1 - p) out[k++] = 2 * (n - 1) - i + p;
    return k;
}

template <typename V> __device__ __forceinline__ V vzero();
template <> __device__ __forceinline__ uint4 vzero<uint4>() { return make_uint4(0, 0, 0, 0); }
template <> __device__ __forceinline__ uint32_t vzero<uint32_t>() { return 0u; }
template <> __device__ __forceinline__ uint16_t vzero<uint16_t>() { return 0; }

// V: the piece type; add(a, b) adds two pieces element-wise in the tensor's dtype (fp32 accumulation per pair)
template <int DT> struct piece_ops;
template <> struct piece_ops<CM_F32> {
    static __device__ __forceinline__ uint4 add(uint4 a, uint4 b) {
        return make_uint4(__float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x)), __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y)),
                          __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z)), __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w)));
    }
    static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return __float_as_uint(__uint_as_float(a) + __uint_as_float(b)); }
    static __device__ __forceinline__ uint16_t add(uint16_t a, uint16_t) { return a; }
};
template <> struct piece_ops<CM_BF16> {
    static __device__ __forceinline__ uint32_t add2(uint32_t a, uint32_t b) {
        return cm_pack_bf16(cm_bf16_lo(a) + cm_bf16_lo(b), cm_bf16_hi(a) + cm_bf16_hi(b));
    }
    static __device__ __forceinline__ uint4 add(uint4 a, uint4 b) { return make_uint4(add2(a.x, b.x), add2(a.y, b.y), add2(a.z, b.z), add2(a.w, b.w)); }
    static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return add2(a, b); }
    static __device__ __forceinline__ uint16_t add(uint16_t a, uint16_t b) { return (uint16_t)(add2(a, b) & 0xffffu); }
};

// pieces per channel row: cv; src / dst are dense (batch, T or T + 2p, F or F + 2p, cv pieces)
template <typename V, int DT, bool BWD>
__global__ __launch_bounds__(256) void reflect_pad_kernel(const V *__restrict__ src, V *__restrict__ dst, const int batch, const int T, const int F,
                                                          const int cv, const int p) {
    const int To = T + 2 * p, Fo = F + 2 * p;
    const int64_t n = (int64_t)batch * (BWD ? T : To) * (BWD ? F : Fo) * cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cv);
        int64_t r = i / cv;
        const int f = (int)(r % (BWD ? F : Fo));
        r /= (BWD ? F : Fo);
        const int t = (int)(r % (BWD ? T : To)), b = (int)(r / (BWD ? T : To));
        if constexpr (!BWD) {
            dst[i] = src[(((int64_t)b * T + refl(t - p, T)) * F + refl(f - p, F)) * cv + c];
        } else {
            int ts[3], fs[3];
            const int nt = sources(t, T, p, ts), nf = sources(f, F, p, fs);
            V acc = vzero<V>();
            bool first = true;
            for (int a = 0; a < nt; ++a)
                for (int e = 0; e < nf; ++e) {
                    const V v = src[(((int64_t)b * To + ts[a]) * Fo + fs[e]) * cv + c];
                    acc = first ? v : piece_ops<DT>::add(acc, v);
                    first = false;
                }
            dst[i] = acc;
        }
    }
}

template <typename V, int DT>
int launch(const void *src, void *dst, int batch, int T, int F, int cv, int p, bool bwd, hipStream_t st) {
    const int64_t n = (int64_t)batch * (bwd ? T : T + 2 * p) * (bwd ? F : F + 2 * p) * cv;
    const int64_t blocks = (n + 255) / 256;
    const dim3 grid((unsigned)(blocks > (1 << 20) ? (1 << 20) : blocks));
    if (bwd) hipLaunchKernelGGL((reflect_pad_kernel<V, DT, true>), grid, dim3(256), 0, st, (const V *)src, (V *)dst, batch, T, F, cv, p);
    else hipLaunchKernelGGL((reflect_pad_kernel<V, DT, false>), grid, dim3(256), 0, st, (const V *)src, (V *)dst, batch, T, F, cv, p);
    return cm_launch_status("cm_reflect_pad_tf");
}

}  // namespace

extern "C" int cm_reflect_pad_tf(const void *src, void *dst, int32_t batch, int32_t time, int32_t freq, int32_t channels, int32_t pad,
                                 int32_t dtype, int32_t backward, void *stream) {
    CM_REQUIRE(src && dst && src != dst && batch > 0 && time > 0 && freq > 0 && channels > 0, CM_EINVAL, "reflect_pad_tf: bad sizes or NULL tensor");
    CM_REQUIRE(pad >= 1 && 2 * pad < time && 2 * pad < freq, CM_EUNSUPPORTED, "reflect_pad_tf: needs 1 <= pad and 2 pad < time, freq (got pad %d, %d x %d)", pad, time, freq);
    CM_REQUIRE(dtype == CM_F32 || dtype == CM_BF16, CM_EUNSUPPORTED, "reflect_pad_tf: dtype %d unsupported", dtype);
    const int es = dtype == CM_F32 ? 4 : 2;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int row = channels * es;
    if (row % 16 == 0 && cm_aligned(src, 16) && cm_aligned(dst, 16))
        return dtype == CM_F32 ? launch<uint4, CM_F32>(src, dst, batch, time, freq, row / 16, pad, backward != 0, st)
                               : launch<uint4, CM_BF16>(src, dst, batch, time, freq, row / 16, pad, backward != 0, st);
    if (row % 4 == 0 && cm_aligned(src, 4) && cm_aligned(dst, 4))
        return dtype == CM_F32 ? launch<uint32_t, CM_F32>(src, dst, batch, time, freq, row / 4, pad, backward != 0, st)
                               : launch<uint32_t, CM_BF16>(src, dst, batch, time, freq, row / 4, pad, backward != 0, st);
    CM_REQUIRE(dtype == CM_BF16 && cm_aligned(src, 2) && cm_aligned(dst, 2), CM_EALIGN, "reflect_pad_tf: misaligned tensors");
    return launch<uint16_t, CM_BF16>(src, dst, batch, time, freq, channels, pad, backward != 0, st);
}
