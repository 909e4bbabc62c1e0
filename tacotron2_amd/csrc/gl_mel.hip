// Mel -> linear-frequency inversion on the matrix cores (gfx950 fp32-input MFMA).
//
// Replaces griffin_lim.py:18-19:   np.dot(filterbank.T, mel).T
//     mag[b][t][f] = sum_m mel[b][m][t] * filterbank[m][f]
// i.e. C[T x F] = A[T x K] * Bm[K x F] with A[t][m] = mel[b][m][t] (contiguous in t)
// and Bm = the filterbank as stored (contiguous in f); K = n_mels (80).
//
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; the accumulator holds col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  With i = t and j = f both operand loads
// and the C stores are contiguous across lanes, so no LDS staging is needed; the
// output is written directly in the (B, T, F) row-major layout the Griffin-Lim
// kernels read.  The result is bit-for-bit an fp32 fma chain over m = 0..K-1.
#include "gl_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileT = 32;      // rows (t) per wave
constexpr int kTileF = 128;     // cols (f) per wave: 4 accumulators of 32
constexpr int kWaves = 4;       // waves per workgroup, stacked along t
constexpr int kKC = 80;         // mel bands per LDS stage of the filterbank tile (40 KB)
typedef float v4f __attribute__((ext_vector_type(4)));

// LOGMEL = false: mel is [B][K][T] linear amplitude (what save_mel_to_wav receives).
// LOGMEL = true : mel is [B][T][K] natural-log amplitude, the decoder's output layout; the kernel
//                 applies dynamic_range_decompression (data.py:84-85: exp(mel) / compression_factor)
//                 and the transpose of infer.py:27-28 on the fly, so no (80, T) array is materialised.
template <bool LOGMEL>
__global__ void __launch_bounds__(64 * kWaves) k_mel_to_mag(const float* __restrict__ fb,
                                                            const float* __restrict__ mel,
                                                            float* __restrict__ mag, int K, int F, int T,
                                                            float inv_cf, float floor_val, int n_ftiles,
                                                            int n_tblocks, int n_groups) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by linear id, and a 128-byte
    // store into the 4100-byte rows rarely covers a whole cache line: the F-tiles that share lines
    // (same utterance, same frames) are given ids with equal id % 8, so they run on one XCD and
    // their partial lines merge in that XCD's L2 instead of leaving as partial writes from several.
    const int id = blockIdx.x;
    const int q = id >> 3;
    const int ft = q % n_ftiles;
    const int g = (q / n_ftiles) * 8 + (id & 7);        // (utterance, frame block)
    if (g >= n_groups) return;                          // padding of the last group of eight
    const int b = g / n_tblocks;
    const int t0 = ((g - b * n_tblocks) * kWaves + wave) * kTileT;
    const int f0 = ft * kTileF;
    const bool active = t0 < T;   // wave-uniform; idle waves still take part in the barriers below

    const int li = lane & 31;
    const int lk = lane >> 5;
    const int t = t0 + li;
    const bool t_ok = active && t < T;
    const float* a_ptr = LOGMEL ? mel + ((long long)b * T + (t_ok ? t : 0)) * K
                                : mel + (long long)b * K * T + (t_ok ? t : 0);

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // The filterbank tile of the workgroup (kKC bands x 128 bins) is staged in LDS once per K-chunk, laid
    // out [k][j][4] so that the four B values a lane feeds to its four MFMAs are one ds_read_b128; the
    // K loop then issues one coalesced global load (operand A) and one LDS read per step instead of five
    // dependent global loads (the kernel spent 58 % of its wave time in s_waitcnt on those).
    __shared__ __attribute__((aligned(16))) v4f Bs[kKC][32];
    for (int kc = 0; kc < K; kc += kKC) {
        // Operand A of the whole chunk goes to registers first (kKC/2 independent loads in flight while
        // the workgroup stages B), so the MFMA loop below touches no global memory.
        float av[kKC / 2];
#pragma unroll
        for (int s = 0; s < kKC / 2; ++s) {
            const int k = kc + 2 * s + lk;
            const bool ok = t_ok && k < K;
            const int kk = ok ? k : 0;
            const float a = LOGMEL ? a_ptr[kk] : a_ptr[(long long)kk * T];
            av[s] = ok ? (LOGMEL ? expf(a) * inv_cf : a) : 0.0f;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < kKC * 32; e += 64 * kWaves) {
            const int k = e >> 5, jj = e & 31;
            const bool k_ok = kc + k < K;
            const float* row = fb + (long long)(k_ok ? kc + k : 0) * F;
            v4f v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int f = f0 + 32 * c + jj;
                v[c] = (k_ok && f < F) ? row[f] : 0.0f;
            }
            Bs[k][jj] = v;
        }
        __syncthreads();
        const int kend = min(kKC, K - kc);
        if (active) {
#pragma unroll
            for (int s = 0; s < kKC / 2; ++s) {
                if (2 * s < kend) {                          // wave-uniform; rows past K are zero in Bs and av
                    const v4f bv = Bs[2 * s + lk][li];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[j], acc[j], 0, 0, 0);
                }
            }
        }
    }
    if (!active) return;

    float* out = mag + (long long)b * T * F;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = f0 + 32 * j + li;
        if (f >= F) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = t0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < T) out[(long long)row * F + f] = fmaxf(acc[j][r], floor_val);   // -inf: no clamp
        }
    }
}

// N3  data.py:112-114: mel = filterbank @ |stft|, log(clip(mel, 0.01) * compression_factor).
//     logmel[b][m][t] = log(max(sum_f fb[m][f] * mag[b][t][f], clip_min) * cf)
// C[m x t] = A[m x f] * B[f x t] with i = m, j = t: the accumulator's 32 lanes of one register are 32
// consecutive t of one mel band, so the (B, n_mels, T) output is stored coalesced.  One wave owns
// 32 frames and all mel bands (kMelTiles accumulators of 32 bands); K = F runs two bins per MFMA.
constexpr int kMelTiles = 4;     // up to 128 mel bands
__global__ void __launch_bounds__(64 * kWaves) k_mag_to_logmel(const float* __restrict__ fb,
                                                               const float* __restrict__ mag,
                                                               float* __restrict__ logmel, int n_mels, int F, int T,
                                                               float clip_min, float cf) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * kWaves + wave) * 32;
    if (t0 >= T) return;  // wave-uniform
    const int li = lane & 31;
    const int lk = lane >> 5;
    const int t = t0 + li;
    const bool t_ok = t < T;
    const float* b_ptr = mag + ((long long)b * T + (t_ok ? t : 0)) * F;
    const int tiles = (n_mels + 31) / 32;
    f32x16 acc[kMelTiles];
#pragma unroll
    for (int j = 0; j < kMelTiles; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    for (int k0 = 0; k0 < F; k0 += 2) {
        const int k = k0 + lk;
        const bool k_ok = k < F;
        const int kk = k_ok ? k : 0;
        const float bv = (k_ok && t_ok) ? b_ptr[kk] : 0.0f;
#pragma unroll
        for (int j = 0; j < kMelTiles; ++j) {
            if (j < tiles) {
                const int m = 32 * j + li;
                const float av = (k_ok && m < n_mels) ? fb[(long long)m * F + kk] : 0.0f;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
            }
        }
    }
    float* out = logmel + (long long)b * n_mels * T;
    if (!t_ok) return;
#pragma unroll
    for (int j = 0; j < kMelTiles; ++j) {
        if (j >= tiles) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < n_mels) out[(long long)m * T + t] = logf(fmaxf(acc[j][r], clip_min) * cf);
        }
    }
}

}  // namespace

int gl_mag_to_logmel_launch(const float* fb, const float* mag, float* logmel, int n_mels, int F, int B, int T,
                            float clip_min, float compression_factor, hipStream_t st) {
    dim3 grid((T + 32 * kWaves - 1) / (32 * kWaves), B);
    hipLaunchKernelGGL(k_mag_to_logmel, grid, dim3(64 * kWaves), 0, st, fb, mag, logmel, n_mels, F, T, clip_min,
                       compression_factor);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_mel_launch(const float* fb, const float* mel, float* mag, int n_mels, int F, int B, int T, bool logmel,
                  float compression_factor, bool clamp, hipStream_t st) {
    const float floor_val = clamp ? 0.0f : -INFINITY;     // GL_OPT_CLAMP_MAG (pseudo-inverse bases go negative)
    const int n_ftiles = (F + kTileF - 1) / kTileF;
    const int n_tblocks = (T + kTileT * kWaves - 1) / (kTileT * kWaves);
    const int n_groups = n_tblocks * B;
    dim3 grid((unsigned)(((n_groups + 7) / 8) * 8 * n_ftiles));
    if (logmel)
        hipLaunchKernelGGL(k_mel_to_mag<true>, grid, dim3(64 * kWaves), 0, st, fb, mel, mag, n_mels, F, T,
                           1.0f / compression_factor, floor_val, n_ftiles, n_tblocks, n_groups);
    else
        hipLaunchKernelGGL(k_mel_to_mag<false>, grid, dim3(64 * kWaves), 0, st, fb, mel, mag, n_mels, F, T, 1.0f,
                           floor_val, n_ftiles, n_tblocks, n_groups);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}
