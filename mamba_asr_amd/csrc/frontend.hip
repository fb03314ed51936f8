// frontend.hip — Fbank back end and SpecAugment masking for gfx950 (contracts in include/conmamba_hip.h).
// The STFT itself (framing, Hamming window, real FFT) stays on the vendor FFT; everything after it — power,
// mel projection, dB, per-utterance top_db clamp, global normalisation, masking — is fused here.
#include "cm_common.h"

namespace {

// one workgroup = one (batch, 16-frame tile); thread = (frame within tile, mel group).
// power spectrum of the tile is staged in LDS [n_freq][16]; the mel filterbank is sparse (triangles), but a dense
// 257 x 80 product per frame is only 41 kFLOP: HBM traffic (reading the complex spectrum once) dominates.
constexpr int FT = 16;
constexpr int PWS = FT + 1;   // LDS row stride of the power tile: odd, so different bins of one frame sit in different banks
                              // (stride 16 put all bins of a frame in 4 banks: the mel loop ran 16-way conflicted)

__global__ __launch_bounds__(256) void fbank_mel_db_kernel(const cm_fbank_args p) {
    extern __shared__ float pw[];                                 // [n_freq][FT] then [2][n_mels] band limits
    const int b = blockIdx.y, t0 = blockIdx.x * FT;
    const int F = p.n_freq, T = p.frames, M = p.n_mels;
    int *band = reinterpret_cast<int *>(pw + F * PWS);             // triangular filters are contiguous bands: [lo, hi)
    float *bw = reinterpret_cast<float *>(band + 3 * M + 1);      // packed band weights (when provided)
    const bool packed = p.band_w != nullptr && p.band_off != nullptr && p.band_lo != nullptr;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {           // triangular filters: contiguous bands (host-provided)
        band[m] = p.band_lo ? p.band_lo[m] : 0;
        band[M + m] = p.band_hi ? p.band_hi[m] : F;
    }
    if (packed) {
        for (int m = threadIdx.x; m <= M; m += blockDim.x) band[2 * M + m] = p.band_off[m];
        const int total = p.band_off[M];
        for (int i = threadIdx.x; i < total; i += blockDim.x) bw[i] = p.band_w[i];
    }
    const bool dflt = p.spec_bs == 0 && p.spec_fs == 0 && p.spec_ts == 0;
    const int64_t sb = dflt ? (int64_t)F * T : p.spec_bs, sf = dflt ? T : p.spec_fs, st = dflt ? 1 : p.spec_ts;
    const float2 *spec = reinterpret_cast<const float2 *>(p.spec) + (int64_t)b * sb;
    const bool f_fast = sf < st;                                  // which axis is contiguous in memory
    for (int i = threadIdx.x; i < F * FT; i += blockDim.x) {
        const int f = f_fast ? i % F : i / FT, j = f_fast ? i / F : i % FT;   // consecutive threads -> contiguous bytes
        float v = 0.f;
        if (t0 + j < T) {
            const float2 c = spec[(int64_t)f * sf + (int64_t)(t0 + j) * st];
            v = c.x * c.x + c.y * c.y;
        }
        pw[f * PWS + j] = v;
    }
    __syncthreads();
    float local_max = -INFINITY;
    for (int o = threadIdx.x; o < FT * M; o += blockDim.x) {
        const int j = o / M, m = o % M;                           // consecutive threads -> consecutive mels (coalesced store)
        if (t0 + j >= T) continue;
        float acc = 0.f;
        const int lo = band[m], hi = band[M + m];
        if (packed) {                                             // LDS-only inner loop (no dependent global loads)
            const float *wm = bw + band[2 * M + m] - lo;
            for (int f = lo; f < hi; ++f) acc = fmaf(pw[f * PWS + j], wm[f], acc);
        } else {
            for (int f = lo; f < hi; ++f) acc = fmaf(pw[f * PWS + j], p.fbank[f * M + m], acc);
        }
        const float db = 10.f * log10f(fmaxf(acc, p.amin));
        p.db[((int64_t)b * T + t0 + j) * M + m] = db;
        local_max = fmaxf(local_max, db);
    }
    // per-tile maximum: wave reduce, then the four waves through LDS.  With umax_part the tile maximum is stored (no
    // atomics: 32k same-address float atomics cost more than the rest of this kernel, 208 of 379 us at 32 x 4000 frames)
    // and cm_fbank_finish reduces the tiles of an utterance; without it, one atomic per workgroup.
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, off, 64));
    __syncthreads();                                              // pw is dead: reuse its first floats
    if ((threadIdx.x & 63) == 0) pw[threadIdx.x >> 6] = local_max;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mx = fmaxf(fmaxf(pw[0], pw[1]), fmaxf(pw[2], pw[3]));
        if (p.umax_part) {
            p.umax_part[(int64_t)b * gridDim.x + blockIdx.x] = mx;
        } else if (mx > -INFINITY) {
            int *addr = reinterpret_cast<int *>(p.umax + b);
            if (mx >= 0.f) atomicMax(addr, __float_as_int(mx));
            else atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(mx));
        }
    }
}

// grid (chunks, batch): a workgroup first reduces its utterance's tile maxima (when umax_part is given), then clamps and
// normalises its slice of the utterance.
__global__ __launch_bounds__(256) void fbank_finish_kernel(const cm_fbank_args p, int ntiles) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    float um;
    if (p.umax_part) {
        float mx = -INFINITY;
        for (int i = threadIdx.x; i < ntiles; i += blockDim.x) mx = fmaxf(mx, p.umax_part[(int64_t)b * ntiles + i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        um = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (blockIdx.x == 0 && threadIdx.x == 0) p.umax[b] = um;
    } else {
        um = p.umax[b];
    }
    const float floor_db = um - p.top_db;
    const int64_t per_utt = (int64_t)p.frames * p.n_mels;
    float *db = p.db + (int64_t)b * per_utt;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_utt; i += (int64_t)gridDim.x * blockDim.x) {
        float v = fmaxf(db[i], floor_db);
        if (p.mean) { const int m = (int)(i % p.n_mels); v = (v - p.mean[m]) / p.std[m]; }
        db[i] = v;
    }
}

__global__ __launch_bounds__(256) void spec_drop_kernel(const cm_spec_drop_args p) {
    const int64_t n = (int64_t)p.batch * p.frames * p.n_mels;
    const float fill = *p.fill;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % p.n_mels);
        const int t = (int)((i / p.n_mels) % p.frames);
        const int b = (int)(i / ((int64_t)p.frames * p.n_mels));
        const int pos = p.dim == 1 ? t : f;
        bool hit = false;
        for (int k = 0; k < p.n_masks; ++k) {
            const int s = p.start[b * p.n_masks + k], l = p.length[b * p.n_masks + k];
            hit = hit || (pos >= s && pos < s + l);
        }
        if (hit) p.feats[i] = fill;
    }
}

}  // namespace

extern "C" int cm_fbank_mel_db(const cm_fbank_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "fbank_mel_db: args is NULL");
    const cm_fbank_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.n_freq > 0 && a.frames > 0 && a.n_mels > 0 && a.spec && a.fbank && a.db && a.umax, CM_EINVAL,
               "fbank_mel_db: bad sizes or NULL tensor");
    const size_t smem = (size_t)a.n_freq * PWS * 4 + (size_t)(3 * a.n_mels + 1) * 4 + (a.band_w ? (size_t)4096 * 4 : 0);
    CM_REQUIRE(a.batch <= 65535 && smem <= 64 * 1024, CM_EUNSUPPORTED, "fbank_mel_db: n_freq %d too large", a.n_freq);
    dim3 grid((a.frames + FT - 1) / FT, a.batch);
    hipLaunchKernelGGL(fbank_mel_db_kernel, grid, dim3(256), smem, reinterpret_cast<hipStream_t>(a.stream), a);
    return cm_launch_status("cm_fbank_mel_db");
}

extern "C" int cm_fbank_finish(const cm_fbank_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "fbank_finish: args is NULL");
    const cm_fbank_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.frames > 0 && a.n_mels > 0 && a.db && a.umax, CM_EINVAL, "fbank_finish: bad sizes or NULL tensor");
    CM_REQUIRE(!a.mean || a.std, CM_EINVAL, "fbank_finish: mean without std");
    CM_REQUIRE(a.batch <= 65535, CM_EINVAL, "fbank_finish: batch %d exceeds the grid limit", a.batch);
    const int64_t per_utt = (int64_t)a.frames * a.n_mels;
    int64_t chunks = (per_utt + 256 * 8 - 1) / (256 * 8);
    if (chunks > 256) chunks = 256;
    hipLaunchKernelGGL(fbank_finish_kernel, dim3((unsigned)chunks, a.batch), dim3(256), 0, reinterpret_cast<hipStream_t>(a.stream), a,
                       (a.frames + FT - 1) / FT);
    return cm_launch_status("cm_fbank_finish");
}

extern "C" int cm_spec_drop(const cm_spec_drop_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "spec_drop: args is NULL");
    const cm_spec_drop_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.frames > 0 && a.n_mels > 0 && a.n_masks >= 0 && a.feats && a.fill, CM_EINVAL,
               "spec_drop: bad sizes or NULL tensor");
    CM_REQUIRE(a.n_masks == 0 || (a.start && a.length), CM_EINVAL, "spec_drop: masks requested without start/length");
    CM_REQUIRE(a.dim == 1 || a.dim == 2, CM_EINVAL, "spec_drop: dim must be 1 (time) or 2 (frequency)");
    if (a.n_masks == 0) return CM_OK;
    const int64_t n = (int64_t)a.batch * a.frames * a.n_mels;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(spec_drop_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(a.stream), a);
    return cm_launch_status("cm_spec_drop");
}
