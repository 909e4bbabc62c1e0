// Small streaming kernels around the Griffin-Lim loop: device RNG for the start
// signal, per-utterance peak normalisation, row copy.
#include <mutex>

#include "gl_internal.h"

namespace {

constexpr int kThreads = 256;

// ---- Philox4x32-10 -----------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ float u01(uint32_t r) {  // (0, 1]
    return ((float)(r >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// Stands in for np.random.randn(len_samples) (audio_utilities.py:176) in
// throughput runs; four normals per counter value.  Element i of the call is element
// `offset + i` of the seed's stream, so a rank can fill just its slice of a long signal.
__global__ void __launch_bounds__(kThreads) k_randn(float* __restrict__ out, long long n, uint64_t seed,
                                                    long long offset) {
    const long long q0 = offset >> 2;
    const long long nq = ((offset + n + 3) >> 2) - q0;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; t < nq; t += (long long)gridDim.x * blockDim.x) {
        const long long q = q0 + t;
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float r0 = sqrtf(-2.0f * __logf(u01(c[0])));
        float r1 = sqrtf(-2.0f * __logf(u01(c[2])));
        float s0, c0, s1, c1;
        __sincosf(6.283185307179586f * u01(c[1]), &s0, &c0);
        __sincosf(6.283185307179586f * u01(c[3]), &s1, &c1);
        float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
        const long long base = (q << 2) - offset;      // index into `out` of this quad's first normal
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (base + j >= 0 && base + j < n) out[base + j] = v[j];
    }
}

// ---- chunk stitching (long-form reconstruction) -------------------------------------------
// out[out_offset + i] += w(i) * chunk[i]: w ramps 0 -> 1 over the first fade_in samples and
// 1 -> 0 over the last fade_out samples (half-sample centred, so two chunks whose fades overlap
// by the same length sum to exactly one), 1 in between.
__global__ void __launch_bounds__(kThreads) k_stitch_add(float* __restrict__ out, long long out_len,
                                                         const float* __restrict__ chunk, long long chunk_len,
                                                         long long out_offset, long long fade_in,
                                                         long long fade_out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunk_len;
         i += (long long)gridDim.x * blockDim.x) {
        const long long o = out_offset + i;
        if (o < 0 || o >= out_len) continue;
        float w = 1.0f;
        if (i < fade_in) w = ((float)i + 0.5f) / (float)fade_in;
        if (chunk_len - 1 - i < fade_out) w *= ((float)(chunk_len - 1 - i) + 0.5f) / (float)fade_out;
        out[o] += w * chunk[i];
    }
}

// ---- N3: sample-rate conversion of librosa.load (data.py:101) --------------------------------------
// Band-limited sinc interpolation as resampy's 'kaiser_best' does it (Smith's algorithm): output sample t sits at
// time t / ratio of the input; the filter's right wing is tabulated at num_table points per zero crossing and read
// with linear interpolation (win + eta * delta), stepped by index_step = scale * num_table so that its cutoff
// follows the lower of the two rates; left wing over x[n], x[n-1], ..., right wing over x[n+1], ...  One thread
// per output sample, taps in resampy's order, weights in double, the running sum rounded to float after every
// tap like a float32 output array.  (~2 x 64 / scale taps per sample: not a hot path, it runs once per file.)
__global__ void __launch_bounds__(kThreads) k_resample(const float* __restrict__ x, long long x_stride, long long n_in,
                                                       float* __restrict__ y, long long y_stride, long long n_out,
                                                       long long n_calc, double ratio, const double2* __restrict__ win,
                                                       int nwin, int num_table) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_out) return;
    const float* xr = x + (long long)blockIdx.y * x_stride;
    float* yr = y + (long long)blockIdx.y * y_stride;
    if (t >= n_calc) {                 // librosa.util.fix_length: the one sample ceil() adds over int() is zero
        yr[t] = 0.0f;
        return;
    }
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const int index_step = (int)(scale * num_table);
    const double time_register = (double)t * (1.0 / ratio);
    const long long n = (long long)time_register;
    float acc = 0.0f;
    double frac = scale * (time_register - (double)n);
    double index_frac = frac * num_table;
    int offset = (int)index_frac;
    double eta = index_frac - offset;
    long long i_max = (nwin - offset) / index_step;
    if (i_max > n + 1) i_max = n + 1;
    for (long long i = 0; i < i_max; ++i) {
        const double2 w = win[offset + i * index_step];
        acc = (float)((double)acc + (w.x + eta * w.y) * scale * (double)xr[n - i]);
    }
    frac = scale - frac;
    index_frac = frac * num_table;
    offset = (int)index_frac;
    eta = index_frac - offset;
    long long k_max = (nwin - offset) / index_step;
    if (k_max > n_in - n - 1) k_max = n_in - n - 1;
    for (long long k = 0; k < k_max; ++k) {
        const double2 w = win[offset + k * index_step];
        acc = (float)((double)acc + (w.x + eta * w.y) * scale * (double)xr[n + k + 1]);
    }
    yr[t] = acc;
}

// ---- peak normalisation (griffin_lim.py:27-29) ---------------------------------------
__global__ void __launch_bounds__(kThreads) k_absmax(const float* __restrict__ x, long long row_stride,
                                                     long long len, unsigned* __restrict__ peaks) {
    const int b = blockIdx.y;
    const float* row = x + (long long)b * row_stride;
    float m = 0.0f;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthr = (long long)gridDim.x * blockDim.x;
    // 16-byte loads over the aligned body of the row (a streaming pass: 1.8 TB/s with dword loads), scalars for the rest
    const long long n4 = ((reinterpret_cast<uintptr_t>(row) & 15) == 0) ? (len >> 2) : 0;
    const float4* row4 = reinterpret_cast<const float4*>(row);
    long long i4 = tid;
    for (; i4 + 3 * nthr < n4; i4 += 4 * nthr) {                 // four independent loads in flight per thread
        const float4 a = row4[i4], c = row4[i4 + nthr], d = row4[i4 + 2 * nthr], e = row4[i4 + 3 * nthr];
        const float ma = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
        const float mc = fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w)));
        const float md = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
        const float me = fmaxf(fmaxf(fabsf(e.x), fabsf(e.y)), fmaxf(fabsf(e.z), fabsf(e.w)));
        m = fmaxf(m, fmaxf(fmaxf(ma, mc), fmaxf(md, me)));
    }
    for (; i4 < n4; i4 += nthr) {
        const float4 v = row4[i4];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (long long i = 4 * n4 + tid; i < len; i += nthr) m = fmaxf(m, fabsf(row[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    __shared__ float wmax[kThreads / 64];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = wmax[0];
        for (int w = 1; w < kThreads / 64; ++w) r = fmaxf(r, wmax[w]);
        // non-negative floats order like their bit patterns
        atomicMax(&peaks[b], __float_as_uint(r));
    }
}

__global__ void __launch_bounds__(kThreads) k_peak_scale(float* __restrict__ x, long long row_stride, long long len,
                                                         const unsigned* __restrict__ peaks,
                                                         float* __restrict__ peaks_out, int mode) {
    const int b = blockIdx.y;
    const float peak = __uint_as_float(peaks[b]);
    if (peaks_out && blockIdx.x == 0 && threadIdx.x == 0) peaks_out[b] = peak;
    const bool scale = (mode == 0) ? (peak > 1.0f) : (peak > 0.0f);
    if (!scale) return;
    float* row = x + (long long)b * row_stride;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len;
         i += (long long)gridDim.x * blockDim.x)
        row[i] = row[i] / peak;
}

// ---- 16-bit PCM encode (audio_utilities.py:202-210) ------------------------------------
// x /= max|x|; x *= 32767; int(round(x[i])) -- the reference works in float64 and Python's round
// is half-to-even, so the same three float64 operations (IEEE divide, multiply, v_rndne_f64) on the
// fp32 samples give the reference's bytes for the same input.
__global__ void __launch_bounds__(kThreads) k_pcm16(const float* __restrict__ x, long long row_stride, long long len,
                                                    const unsigned* __restrict__ peaks, int16_t* __restrict__ pcm,
                                                    long long pcm_stride, float* __restrict__ peaks_out) {
    const int b = blockIdx.y;
    const float peakf = __uint_as_float(peaks[b]);
    if (peaks_out && blockIdx.x == 0 && threadIdx.x == 0) peaks_out[b] = peakf;
    const double peak = (double)peakf;
    const float* row = x + (long long)b * row_stride;
    int16_t* orow = pcm + (long long)b * pcm_stride;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len;
         i += (long long)gridDim.x * blockDim.x) {
        double v = (double)row[i];
        v = v / peak;          // peak == 0 -> NaN like the reference; the cast below then stores 0
        v = v * 32767.0;
        v = rint(v);
        orow[i] = (peakf > 0.0f) ? (int16_t)(int)v : (int16_t)0;
    }
}

// Ragged batches: rows t >= frames[b] of mag[b] are zeroed.  A zero magnitude row projects to an exactly
// zero frame (X * (0 * rsq) = 0 and the zero-bin rule gives (mag, 0) = 0), which adds nothing to the
// overlap-add, so utterance b comes out as if it had been run alone with frames[b] frames.
__global__ void __launch_bounds__(kThreads) k_mask_frames(float* __restrict__ mag, const int* __restrict__ frames,
                                                          int T, int F) {
    const int b = blockIdx.y;
    const int tb = min(max(frames[b], 0), T);
    const long long n = (long long)(T - tb) * F;
    float* base = mag + ((long long)b * T + tb) * F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        base[i] = 0.0f;
}

}  // namespace

int gl_mask_launch(float* mag, const int* frames, int B, int T, int F, hipStream_t st) {
    hipLaunchKernelGGL(k_mask_frames, dim3(64, B), dim3(kThreads), 0, st, mag, frames, T, F);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_pcm16_launch(gl_plan* p, const float* x, long long row_stride, long long len, int16_t* pcm,
                    long long pcm_stride, float* peaks, int B, hipStream_t st) {
    GL_HIP_CHECK(hipMemsetAsync(p->d_peaks, 0, sizeof(unsigned) * (size_t)B, st));
    // about four workgroups per CU over the whole batch, each thread with several 16-byte loads in flight
    long long blocks = (len + kThreads * 16 - 1) / (kThreads * 16);
    const long long cap = (4LL * p->n_cu + B - 1) / B;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks, B);
    hipLaunchKernelGGL(k_absmax, grid, dim3(kThreads), 0, st, x, row_stride, len, p->d_peaks);
    GL_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_pcm16, grid, dim3(kThreads), 0, st, x, row_stride, len, p->d_peaks, pcm, pcm_stride, peaks);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_stitch_launch(float* out, long long out_len, const float* chunk, long long chunk_len, long long out_offset,
                     long long fade_in, long long fade_out, hipStream_t st) {
    long long blocks = (chunk_len + kThreads * 4 - 1) / (kThreads * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_stitch_add, dim3((unsigned)blocks), dim3(kThreads), 0, st, out, out_len, chunk, chunk_len,
                       out_offset, fade_in, fade_out);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_resample_launch(const gl_plan* p, const float* x, long long x_stride, long long n_in, float* y,
                       long long y_stride, long long n_out, long long n_calc, double ratio, int B, hipStream_t st) {
    dim3 grid((unsigned)((n_out + kThreads - 1) / kThreads), B);
    hipLaunchKernelGGL(k_resample, grid, dim3(kThreads), 0, st, x, x_stride, n_in, y, y_stride, n_out, n_calc, ratio,
                       reinterpret_cast<const double2*>(p->d_rs_win), GL_RS_ZEROS * GL_RS_TABLE + 1, GL_RS_TABLE);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_randn_launch(float* out, long long n, uint64_t seed, long long offset, hipStream_t st) {
    long long nq = ((offset + n + 3) >> 2) - (offset >> 2);
    long long blocks = (nq + kThreads - 1) / kThreads;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_randn, dim3((unsigned)blocks), dim3(kThreads), 0, st, out, n, seed, offset);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- peak normalisation without any allocation (gl_peak_normalise_static) --------------------
// The per-row maxima k_absmax / k_peak_scale exchange are the only device memory gl_peak_normalise
// needs beyond its operands; here they live in a file-scope device array instead of the plan, so a
// caller that may not allocate (stream capture, serving loops) needs no plan at all.  `scratch` is
// the host-side handle of the array only: host code uses the per-device address hipGetSymbolAddress
// returned (every GPU holds its own copy at its own address).
static __device__ __align__(256) unsigned scratch[GL_STATIC_MAX_BATCH];     // 16 KB per GPU, held until exit

namespace {
constexpr int kMaxDevices = 64;
struct StaticScratch {
    unsigned* ptr;
    int n_cu;
};
StaticScratch g_static[kMaxDevices];
std::mutex g_static_mutex;
}  // namespace

// Once per device, with that device current: the array's address on it and its CU count.
int gl_static_scratch_setup(int device, unsigned** ptr, int* n_cu) {
    if (device < 0 || device >= kMaxDevices)
        return gl_set_error(GL_ERR_INVALID, "device %d outside [0, %d)", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_static_mutex);
    StaticScratch& s = g_static[device];
    if (!s.ptr) {
        unsigned* q = nullptr;
        GL_HIP_CHECK(hipGetSymbolAddress((void**)&q, HIP_SYMBOL(scratch)));
        if (!q) return gl_set_error(GL_ERR_HIP, "hipGetSymbolAddress returned a null address on device %d", device);
        int cus = 0;
        GL_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
        s.n_cu = cus > 0 ? cus : 256;
        s.ptr = q;
    }
    *ptr = s.ptr;
    *n_cu = s.n_cu;
    return 0;
}

int gl_peak_launch_on(unsigned* d_peaks, int n_cu, float* x, long long row_stride, long long len, float* peaks,
                          int mode, int B, hipStream_t st) {
    // the maxima buffer keeps the last call's values: the B entries this call uses start at zero
    GL_HIP_CHECK(hipMemsetAsync(d_peaks, 0, sizeof(unsigned) * (size_t)B, st));
    long long blocks = (len + kThreads * 16 - 1) / (kThreads * 16);
    const long long cap = (4LL * n_cu + B - 1) / B;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks, B);
    hipLaunchKernelGGL(k_absmax, grid, dim3(kThreads), 0, st, x, row_stride, len, d_peaks);
    GL_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_peak_scale, grid, dim3(kThreads), 0, st, x, row_stride, len, d_peaks, peaks, mode);
    GL_HIP_CHECK(hipGetLastError());
    return 0;
}

int gl_peak_launch(gl_plan* p, float* x, long long row_stride, long long len, float* peaks, int mode, int B,
                   hipStream_t st) {
    return gl_peak_launch_on(p->d_peaks, p->n_cu, x, row_stride, len, peaks, mode, B, st);
}

int gl_copy_rows(float* dst, const float* src, long long n, hipStream_t st) {
    if (dst == src || n <= 0) return 0;
    GL_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

// ---- the no-allocation entry point (include/gl_hip.h) ------------------------------------------
extern "C" int gl_peak_normalise_static(float* x, long long row_stride, long long len, float* peaks, int mode, int B,
                                        int device, void* stream) {
    if (!x) return gl_set_error(GL_ERR_INVALID, "gl_peak_normalise_static: x is NULL");
    if (B < 1 || B > GL_STATIC_MAX_BATCH)
        return gl_set_error(GL_ERR_INVALID, "gl_peak_normalise_static: B=%d outside [1, %d] (the static scratch "
                            "array holds GL_STATIC_MAX_BATCH row maxima)", B, GL_STATIC_MAX_BATCH);
    if (len < 1 || row_stride < len || (mode != 0 && mode != 1))
        return gl_set_error(GL_ERR_INVALID, "gl_peak_normalise_static: bad len/stride/mode");
    int ndev = 0, prev = -1;
    GL_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return gl_set_error(GL_ERR_INVALID, "gl_peak_normalise_static: device %d out of range (%d visible)", device, ndev);
    GL_HIP_CHECK(hipGetDevice(&prev));
    if (prev != device) GL_HIP_CHECK(hipSetDevice(device));
    unsigned* d_peaks = nullptr;
    int n_cu = 0;
    int rc = gl_static_scratch_setup(device, &d_peaks, &n_cu);      // first call on a device: symbol lookup
    if (!rc) rc = gl_peak_launch_on(d_peaks, n_cu, x, row_stride, len, peaks, mode, B, (hipStream_t)stream);
    if (prev != device) (void)hipSetDevice(prev);
    return rc;
}
