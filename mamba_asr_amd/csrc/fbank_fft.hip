// fbank_fft.hip — waveform -> log-mel features in ONE kernel (contract: cm_fbank_wav in include/conmamba_hip.h;
// speechbrain Fbank semantics as restated in SURVEY Appendix A: STFT(n_fft 512, Hamming window centred in the frame,
// hop, center=True with zero padding) -> power -> triangular mel -> 10 log10 (floor amin); cm_fbank_finish then
// applies top_db and the global normalisation).
//
// Through the vendor FFT this stage was: pad kernel, frame+window kernel (524 MB written at 64 x 40 s), rocFFT (two
// kernels + per-chunk copies), then cm_fbank_mel_db reading the 526 MB complex spectrum: ~1.6 ms per 64-utterance step
// for 6 GFLOP.  Here a wave owns a frame end to end and nothing but the waveform (read ~3.2x through L1/L2: frames
// overlap) and the 80 log-mel values per frame touch memory:
//   * real FFT of 512 points = complex FFT of 256 points on z[m] = x[2m] + i x[2m+1] + one split step;
//   * complex FFT-256 = four radix-4 decimation-in-frequency stages, one butterfly per lane per stage, in place in a
//     2 KB per-wave LDS buffer (a wave's LDS operations execute in order: no barriers); stage 0 takes its inputs
//     straight from global memory (8-byte loads, window applied in registers) and the lane's twiddles for the three
//     twiddled stages stay in registers across frames;
//   * outputs land in base-4 digit-reversed order, so lane l finds Z[l + 64 i], i = 0..3, in 32 contiguous bytes;
//   * |X[k]|^2 goes to the [bin][frame] power tile in LDS and the mel / dB stage of cm_fbank_mel_db runs unchanged.
#include "cm_common.h"

namespace {

constexpr int FT = 16;            // frames per workgroup (4 per wave)
constexpr int PWS = FT + 1;       // power-tile row stride (odd: conflict-free mel loop)
constexpr int NF = 512;           // n_fft
constexpr int NH = NF / 2;        // complex FFT length
constexpr int BWCAP = 1024;       // band weights staged in LDS (triangular filters: <= 2 per bin = 514 for 257 bins); a larger
                                  // table is read from global memory instead.  4 KB instead of the table's 16 KB bound: a
                                  // workgroup holds 32.5 KB of LDS and four of them (16 waves) share a CU

// 8-byte aligned: LDS accesses are ds_read_b64 / ds_write_b64 (a 4-byte aligned pair compiles to ds_read2_b32, whose two
// dword accesses are banked separately modulo 32 dwords: every element access of the padded layouts below was 2-way)
struct __attribute__((aligned(8))) cf { float re, im; };
__device__ __forceinline__ cf operator+(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf operator-(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {fmaf(a.re, b.re, -a.im * b.im), fmaf(a.re, b.im, a.im * b.re)}; }
__device__ __forceinline__ cf mul_mi(cf a) { return {a.im, -a.re}; }      // a * (-i)

// radix-4 DIF butterfly: y_q = sum_r a_r (-i)^{q r}
__device__ __forceinline__ void bfly4(cf &a0, cf &a1, cf &a2, cf &a3) {
    const cf b0 = a0 + a2, b1 = a0 - a2, b2 = a1 + a3, b3 = mul_mi(a1 - a3);
    a0 = b0 + b2; a2 = b0 - b2; a1 = b1 + b3; a3 = b1 - b3;
}

// LDS position of FFT element p: 4 pad elements per 16 spread the strided butterflies of stages 1 and 2 over the banks
__device__ __forceinline__ int zp(int p) { return p + 4 * (p >> 4); }
// ... and from the output of stage 2 on, one pad element per 4: stage 3 and the split step have every lane walk FOUR
// CONSECUTIVE elements (lane stride 4 elements = 32 bytes: 4-way conflicts on the 8-byte accesses under zp -- rocprofv3
// counted 70 % of this kernel's LDS cycles as bank conflicts and the LDS array busy 85 % of its duration); with a lane
// stride of 5 elements the 32 lanes of a ds_read_b64 group hit 32 different 8-byte slots.  Stage 2 reads under zp and
// writes under zq: a wave's LDS instructions execute in order and its four reads cover the whole buffer before the first write.
__device__ __forceinline__ int zq(int p) { return p + (p >> 2); }
constexpr int ZN = NH + NH / 4;       // padded work-buffer length (both paddings)

__device__ __forceinline__ int rev4_8bit(int k) {                     // reverse the four base-4 digits of k < 256
    return ((k & 3) << 6) | ((k & 12) << 2) | ((k >> 2) & 12) | (k >> 6);
}

__global__ __launch_bounds__(256) void fbank_wav_kernel(const cm_fbank_args p) {
    extern __shared__ float sm[];
    const int b = blockIdx.y, t0 = blockIdx.x * FT;
    const int F = NH + 1, T = p.frames, M = p.n_mels;
    float *pw = sm;                                               // [F][PWS] power tile
    int *band = reinterpret_cast<int *>(pw + F * PWS);            // [3 M + 1]
    float *bw = reinterpret_cast<float *>(band + 3 * M + 1);      // [BWCAP] packed band weights
    cf *zb = reinterpret_cast<cf *>(bw + BWCAP + ((3 * M + 1 + F * PWS) & 1));  // [4 waves][ZN] FFT work buffers (padded), 8-byte aligned
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int m = tid; m < M; m += 256) { band[m] = p.band_lo[m]; band[M + m] = p.band_hi[m]; }
    for (int m = tid; m <= M; m += 256) band[2 * M + m] = p.band_off[m];
    const int bw_total = p.band_off[M];
    const bool bw_lds = bw_total <= BWCAP;
    if (bw_lds)
        for (int i = tid; i < bw_total; i += 256) bw[i] = p.band_w[i];

    // ---- per-lane constants
    const cf *tw = reinterpret_cast<const cf *>(p.twiddle);       // tw[k] = exp(-2 pi i k / 512)
    cf w0[3], w1[3], w2[3];                                       // twiddles of stages 0..2: W_L^{q j}, q = 1..3
    {
        const int j0 = lane, j1 = lane & 15, j2 = lane & 3;
#pragma unroll
        for (int q = 1; q <= 3; ++q) {
            w0[q - 1] = tw[2 * ((q * j0) & 255)];                 // L = 256: W_256^{q j} = tw[2 q j]
            w1[q - 1] = tw[2 * ((4 * q * j1) & 255)];             // L = 64
            w2[q - 1] = tw[2 * ((16 * q * j2) & 255)];            // L = 16
        }
    }
    float2 win[4];                                                // window at samples 2m, 2m+1 for m = lane + 64 q
#pragma unroll
    for (int q = 0; q < 4; ++q) win[q] = *reinterpret_cast<const float2 *>(p.window + 2 * (lane + 64 * q));
    // split-step constants for k = lane + 64 i: position of Z[(256 - k) & 255] and exp(-2 pi i k / 512)
    int ppos[4];
    cf wk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        ppos[i] = zq(rev4_8bit((NH - k) & (NH - 1)));
        wk[i] = tw[k];
    }
    const int r3 = ((lane & 3) << 4) | (lane & 12) | (lane >> 4);  // Z[lane + 64 i] sits at position 4 r3 + i
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.wav + (int64_t)b * p.wav_bs), 0, p.samples * 4, 0x00020000);
    cf *z = zb + wave * ZN;

    // positions (padded) of this lane's butterfly operands in stages 0..3 and of its split-step reads
    int p0[4], p1[4], p2[4], p2w[4], p3[4], ps[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p0[q] = zp(lane + 64 * q);
        p1[q] = zp(64 * (lane >> 4) + (lane & 15) + 16 * q);
        p2[q] = zp(16 * (lane >> 2) + (lane & 3) + 4 * q);
        p2w[q] = zq(16 * (lane >> 2) + (lane & 3) + 4 * q);
        p3[q] = zq(4 * lane + q);
        ps[q] = zq(4 * r3 + q);
    }
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    // frame t covers samples t*hop - 256 .. +255 (center=True); outside [0, samples) the buffer load returns 0.
    // The 8-byte load never straddles the start: t*hop - 256 and 2m are even.
    auto fetch = [&](int t, u32x2(&raw)[4]) {
        const int s0 = t * p.hop - NH;
#pragma unroll
        for (int q = 0; q < 4; ++q) raw[q] = __builtin_amdgcn_raw_buffer_load_b64(wr, (s0 + 2 * (lane + 64 * q)) * 4, 0, 0);
    };
    u32x2 raw[4], nxt[4];
    fetch(t0 + wave * (FT / 4), raw);
    for (int fi = 0; fi < FT / 4; ++fi) {
        const int j = wave * (FT / 4) + fi, t = t0 + j;           // frame (wave-uniform)
        if (fi + 1 < FT / 4) fetch(t + 1, nxt);                   // next frame's samples arrive under this frame's FFT
        if (t < T) {
            cf a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = cf{__uint_as_float(raw[q][0]) * win[q].x, __uint_as_float(raw[q][1]) * win[q].y};
            // stage 0 (L = 256): positions lane + 64 q
            bfly4(a[0], a[1], a[2], a[3]);
            z[p0[0]] = a[0];
            z[p0[1]] = cmul(a[1], w0[0]);
            z[p0[2]] = cmul(a[2], w0[1]);
            z[p0[3]] = cmul(a[3], w0[2]);
            // stage 1 (L = 64): block = lane / 16, j = lane % 16
            a[0] = z[p1[0]]; a[1] = z[p1[1]]; a[2] = z[p1[2]]; a[3] = z[p1[3]];
            bfly4(a[0], a[1], a[2], a[3]);
            z[p1[0]] = a[0]; z[p1[1]] = cmul(a[1], w1[0]); z[p1[2]] = cmul(a[2], w1[1]); z[p1[3]] = cmul(a[3], w1[2]);
            // stage 2 (L = 16): block = lane / 4, j = lane % 4
            a[0] = z[p2[0]]; a[1] = z[p2[1]]; a[2] = z[p2[2]]; a[3] = z[p2[3]];
            bfly4(a[0], a[1], a[2], a[3]);
            z[p2w[0]] = a[0]; z[p2w[1]] = cmul(a[1], w2[0]); z[p2w[2]] = cmul(a[2], w2[1]); z[p2w[3]] = cmul(a[3], w2[2]);
            // stage 3 (L = 4): positions 4 lane .. 4 lane + 3, no twiddles
            a[0] = z[p3[0]]; a[1] = z[p3[1]]; a[2] = z[p3[2]]; a[3] = z[p3[3]];
            bfly4(a[0], a[1], a[2], a[3]);
            z[p3[0]] = a[0]; z[p3[1]] = a[1]; z[p3[2]] = a[2]; z[p3[3]] = a[3];
            // split step: X[k] = E + (-i) W_512^k O,  E = (Z[k] + conj Z[N-k]) / 2,  O = (Z[k] - conj Z[N-k]) / 2
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const cf A = z[ps[i]], Bc = z[ppos[i]];
                const cf E = {0.5f * (A.re + Bc.re), 0.5f * (A.im - Bc.im)};
                const cf O = {0.5f * (A.re - Bc.re), 0.5f * (A.im + Bc.im)};
                const cf X = E + mul_mi(cmul(O, wk[i]));
                pw[(lane + 64 * i) * PWS + j] = fmaf(X.re, X.re, X.im * X.im);
            }
            if (lane == 0) {                                      // Nyquist bin: X[256] = Re Z[0] - Im Z[0]
                const cf A = z[0];
                const float x = A.re - A.im;
                pw[NH * PWS + j] = x * x;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) raw[q] = nxt[q];
    }
    __syncthreads();

    // ---- mel projection + dB.  thread = (mel band, frame) with the FRAME fastest: a wave's 64 lanes are 4 bands x 16
    // frames, so a power-tile read is 16 consecutive floats per band (at most 2-way bank conflicts between the two bands of a
    // 32-lane group; with the band fastest, bands whose first bins are 8 apart met on 4 banks: 8-way), the weight read is
    // a broadcast per band, and the loop length is that of 4 neighbouring bands, not of the 64 widest.
    // dB values pass through LDS so that the global stores are whole 320-byte frame rows.
    float local_max = -INFINITY;
    static_assert(FT == 16, "the mel stage maps 16 frames to 16 lanes");
    float *dbt = reinterpret_cast<float *>(zb);                  // [FT][M] dB tile; the FFT buffers are dead (barrier above)
    for (int o = tid; o < FT * M; o += 256) {
        const int m = o >> 4, j = o & 15;
        float acc = 0.f;
        const int lo = band[m], hi = band[M + m];
        if (bw_lds) {                                             // (two loops: one address space per pointer)
            const float *wm = bw + band[2 * M + m] - lo;
            for (int f = lo; f < hi; ++f) acc = fmaf(pw[f * PWS + j], wm[f], acc);
        } else {
            const float *wm = p.band_w + band[2 * M + m] - lo;
            for (int f = lo; f < hi; ++f) acc = fmaf(pw[f * PWS + j], wm[f], acc);
        }
        const float db = 10.f * log10f(fmaxf(acc, p.amin));
        dbt[j * M + m] = db;
        if (t0 + j < T) local_max = fmaxf(local_max, db);
    }
    __syncthreads();
    {
        const int nrow = min(FT, T - t0);                         // rows t0 .. t0 + nrow - 1 are contiguous in db: one linear copy
        float *dst = p.db + ((int64_t)b * T + t0) * M;
        for (int i = tid; i < nrow * M; i += 256) dst[i] = dbt[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local_max = fmaxf(local_max, __shfl_xor(local_max, off, 64));
    __syncthreads();
    if (lane == 0) pw[wave] = local_max;
    __syncthreads();
    if (tid == 0) p.umax_part[(int64_t)b * gridDim.x + blockIdx.x] = fmaxf(fmaxf(pw[0], pw[1]), fmaxf(pw[2], pw[3]));
}

}  // namespace

extern "C" int cm_fbank_wav(const cm_fbank_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "fbank_wav: args is NULL");
    const cm_fbank_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.frames > 0 && a.n_mels > 0 && a.wav && a.window && a.twiddle && a.db && a.umax_part && a.band_lo &&
                   a.band_hi && a.band_off && a.band_w, CM_EINVAL, "fbank_wav: bad sizes or NULL tensor");
    CM_REQUIRE(a.n_fft == NF && a.n_freq == NH + 1, CM_EUNSUPPORTED, "fbank_wav: n_fft %d unsupported (512 only)", a.n_fft);
    CM_REQUIRE(a.hop > 0 && a.hop % 2 == 0 && a.samples > 0 && a.samples < (1 << 29) && a.wav_bs % 2 == 0 && cm_aligned(a.wav, 8) &&
                   cm_aligned(a.window, 8) && cm_aligned(a.twiddle, 8), CM_EALIGN,
               "fbank_wav: hop and the batch stride must be even, wav / window / twiddle 8-byte aligned");
    CM_REQUIRE(a.frames == 1 + a.samples / a.hop, CM_EINVAL, "fbank_wav: frames must be 1 + samples / hop (center=True)");
    CM_REQUIRE(a.batch <= 65535 && a.n_mels <= 128, CM_EUNSUPPORTED, "fbank_wav: batch / n_mels too large");
    const size_t smem = (size_t)(NH + 1) * PWS * 4 + (size_t)(3 * a.n_mels + 1) * 4 + (size_t)BWCAP * 4 + (size_t)4 * ZN * 8 + 4;
    dim3 grid((a.frames + FT - 1) / FT, a.batch);
    hipLaunchKernelGGL(fbank_wav_kernel, grid, dim3(256), smem, reinterpret_cast<hipStream_t>(a.stream), a);
    return cm_launch_status("cm_fbank_wav");
}
