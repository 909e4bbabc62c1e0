// C-ABI of libgl_hip.so (see include/gl_hip.h): argument validation, plan
// lifetime, and dispatch to the fused (gl_fast.hip) or generic (gl_generic.hip)
// kernels.  Host code only; no kernels live here.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <new>
#include <vector>

#include "gl_internal.h"

namespace {
thread_local char g_err[512] = "";

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) return;
        if (prev != dev && hipSetDevice(dev) != hipSuccess) return;
        ok = true;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// modified Bessel function of the first kind, order 0 (power series; the Kaiser window needs x <= beta ~ 15)
double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    const double q = 0.25 * x * x;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

// Right wing of resampy's 'kaiser_best' interpolation filter (resampy.filters.sinc_window): n + 1 points over
// [0, num_zeros], rolloff * sinc(rolloff * t) tapered by the right half of a Kaiser window of 2 n + 1 points; stored
// with the difference to the next point (interp_delta), 0 after the last.
void resample_filter(std::vector<double>& tab) {
    const int n = GL_RS_ZEROS * GL_RS_TABLE;
    const double pi = 3.14159265358979323846;
    std::vector<double> w(n + 1);
    const double i0b = bessel_i0(GL_RS_BETA);
    for (int j = 0; j <= n; ++j) {
        const double t = (double)GL_RS_ZEROS * j / n;
        const double a = pi * GL_RS_ROLLOFF * t;
        const double sinc = j == 0 ? 1.0 : sin(a) / a;
        const double r = (double)j / n;
        const double kaiser = bessel_i0(GL_RS_BETA * sqrt(fmax(0.0, 1.0 - r * r))) / i0b;
        w[j] = GL_RS_ROLLOFF * sinc * kaiser;
    }
    tab.resize(2 * (size_t)(n + 1));
    for (int j = 0; j <= n; ++j) {
        tab[2 * j] = w[j];
        tab[2 * j + 1] = j < n ? w[j + 1] - w[j] : 0.0;
    }
}
}  // namespace

int gl_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int gl_plan_alloc(gl_plan* p, void** ptr, size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) {
        *ptr = nullptr;
        (void)hipGetLastError();
        return gl_set_error(GL_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    p->bytes += (long long)bytes;
    return 0;
}

extern "C" {

int gl_version(void) { return 300; }

const char* gl_last_error(void) { return g_err; }

long long gl_signal_length(int n_fft, int hop, int frames) { return (long long)frames * hop + n_fft; }

int gl_num_frames(int n_fft, int hop, long long len) {
    if (hop < 1 || len <= n_fft) return 0;
    return (int)((len - n_fft + hop - 1) / hop);  // len(range(0, len - N, H))
}

int gl_plan_create(int n_fft, int hop, int max_frames, int max_batch, int device, gl_plan** out) {
    return gl_plan_create_capacity(n_fft, hop, max_frames, max_batch, 0, device, out);
}

int gl_plan_create_capacity(int n_fft, int hop, int max_frames, int max_batch, long long max_total_frames, int device,
                            gl_plan** out) {
    if (!out) return gl_set_error(GL_ERR_INVALID, "gl_plan_create: out is NULL");
    *out = nullptr;
    if (!is_pow2(n_fft) || n_fft < 64 || n_fft > 4096)
        return gl_set_error(GL_ERR_INVALID, "n_fft must be a power of two in [64, 4096], got %d", n_fft);
    if (hop < 1) return gl_set_error(GL_ERR_INVALID, "hop must be >= 1, got %d", hop);
    if (max_frames < 1 || max_batch < 1 || max_batch > 65535)
        return gl_set_error(GL_ERR_INVALID, "max_frames must be >= 1 and max_batch in [1, 65535] (got %d, %d)",
                            max_frames, max_batch);
    const long long full = (long long)max_frames * max_batch;
    if (max_total_frames < 0 || max_total_frames > full)
        return gl_set_error(GL_ERR_INVALID, "max_total_frames must be in [0, max_frames * max_batch = %lld], got %lld", full,
                            max_total_frames);
    int ndev = 0;
    GL_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return gl_set_error(GL_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard guard(device);
    if (!guard.ok) return gl_set_error(GL_ERR_HIP, "cannot select device %d", device);

    gl_plan* p = new (std::nothrow) gl_plan();
    if (!p) return gl_set_error(GL_ERR_NOMEM, "host allocation failed");
    memset(p, 0, sizeof(*p));
    p->n_fft = n_fft;
    p->hop = hop;
    p->max_T = max_frames;
    p->max_B = max_batch;
    p->cap_frames = max_total_frames > 0 ? max_total_frames : full;
    p->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) p->n_cu = prop.multiProcessorCount;
    if (p->n_cu <= 0) p->n_cu = 256;

    // np.hanning(N) = 0.5 - 0.5*cos(2*pi*n/(N-1))  (audio_utilities.py:88,108), in double
    // periodic Hann of the analysis path: scipy.signal.get_window('hann', N, fftbins=True) = 0.5 - 0.5 cos(2 pi n / N)
    std::vector<float> win(n_fft), win_p(n_fft);
    std::vector<float2> tw(n_fft);
    const double pi = 3.14159265358979323846;
    for (int n = 0; n < n_fft; ++n) {
        win[n] = (float)(0.5 - 0.5 * cos(2.0 * pi * n / (double)(n_fft - 1)));
        win_p[n] = (float)(0.5 - 0.5 * cos(2.0 * pi * n / (double)n_fft));
        double a = -2.0 * pi * n / (double)n_fft;
        tw[n] = make_float2((float)cos(a), (float)sin(a));
    }
    p->wsum_T = -1;
    int rc = gl_plan_alloc(p, (void**)&p->d_window, sizeof(float) * n_fft);
    if (!rc) rc = gl_plan_alloc(p, (void**)&p->d_window_p, sizeof(float) * n_fft);
    if (!rc) rc = gl_plan_alloc(p, (void**)&p->d_tw, sizeof(float2) * n_fft);
    if (!rc) rc = gl_plan_alloc(p, (void**)&p->d_peaks, sizeof(unsigned) * (size_t)max_batch);
    std::vector<double> rs_tab;
    resample_filter(rs_tab);
    if (!rc) rc = gl_plan_alloc(p, (void**)&p->d_rs_win, sizeof(double) * rs_tab.size());
    if (!rc) {
        hipError_t e = hipMemcpy(p->d_window, win.data(), sizeof(float) * n_fft, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_window_p, win_p.data(), sizeof(float) * n_fft, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * n_fft, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_rs_win, rs_tab.data(), sizeof(double) * rs_tab.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = gl_set_error(GL_ERR_HIP, "table upload failed: %s", hipGetErrorString(e));
    }
    // Scratch and tables of every entry point of the plan's default path, now: later calls only enqueue kernels.
    // A geometry the fused kernels cover completely (n_fft 2048) gets the any-N kernels' scratch only when
    // gl_plan_set_path(plan, 1) asks for them.
    const bool fused_run = gl_fast_supported(p, GL_FAST_OP_RUN);
    const bool fused_all = fused_run && gl_fast_supported(p, GL_FAST_OP_STFT) && gl_fast_supported(p, GL_FAST_OP_ISTFT);
    if (!rc && fused_run) rc = gl_fast_alloc(p);
    if (!rc && !fused_all) rc = gl_generic_alloc(p);
    if (rc) {
        gl_plan_destroy(p);
        return rc;
    }
    *out = p;
    return 0;
}

int gl_plan_destroy(gl_plan* p) {
    if (!p) return 0;
    DeviceGuard guard(p->device);
    void* ptrs[] = {p->d_window, p->d_tw,     p->d_frames, p->d_xa,      p->d_xb,
                    p->d_tail_a, p->d_tail_b, p->d_fast_tab, p->d_peaks, p->d_fast_tab2, p->d_magp,
                    p->d_inv_wsum, p->d_prev, p->d_window_p, p->d_yframes, p->d_xc, p->d_w2, p->d_rs_win};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    delete p;
    return 0;
}

long long gl_plan_device_bytes(const gl_plan* p) { return p ? p->bytes : 0; }

int gl_plan_set_path(gl_plan* p, int path) {
    if (!p || path < 0 || path > 1) return gl_set_error(GL_ERR_INVALID, "gl_plan_set_path: bad argument");
    if (path == 1 && !p->d_frames) {         // first switch to the any-N kernels: their scratch (may allocate)
        DeviceGuard guard(p->device);
        int rc = gl_generic_alloc(p);
        if (rc) return rc;
    }
    p->path = path;
    return 0;
}

int gl_plan_set_run_frames(gl_plan* p, int frames) {
    if (!p || frames < 0) return gl_set_error(GL_ERR_INVALID, "gl_plan_set_run_frames: bad argument");
    p->run_frames = frames;
    return 0;
}

int gl_plan_set_options(gl_plan* p, int flags, float momentum) {
    if (!p || (flags & ~(GL_OPT_WSUM_NORMALISE | GL_OPT_CLAMP_MAG)) != 0)
        return gl_set_error(GL_ERR_INVALID, "gl_plan_set_options: bad argument");
    if (!(momentum >= 0.0f && momentum < 1.0f))
        return gl_set_error(GL_ERR_INVALID, "gl_plan_set_options: momentum must be in [0, 1), got %g", (double)momentum);
    // buffers of the variants, the first time one is switched on (a configuration call: may allocate and copy)
    DeviceGuard guard(p->device);
    const int N = p->n_fft;
    if (momentum != 0.0f && !p->d_prev) {
        int rc = gl_plan_alloc(p, (void**)&p->d_prev, gl_signal_floats(p) * sizeof(float));
        if (rc) return rc;
    }
    if ((flags & GL_OPT_WSUM_NORMALISE) && !(p->d_w2 && p->d_inv_wsum && p->w2_uploaded)) {
        // each step is guarded on its own, so a call that failed half way (an allocation, the upload) is completed
        // by the next one instead of leaking a buffer or leaving d_w2 unwritten
        std::vector<double> w2(N);
        const double pi = 3.14159265358979323846;
        for (int n = 0; n < N; ++n) {
            const double w = 0.5 - 0.5 * cos(2.0 * pi * n / (double)(N - 1));
            w2[n] = w * w;
        }
        int rc = 0;
        if (!p->d_w2) rc = gl_plan_alloc(p, (void**)&p->d_w2, sizeof(double) * N);
        if (!rc && !p->d_inv_wsum)
            rc = gl_plan_alloc(p, (void**)&p->d_inv_wsum, (size_t)gl_len(N, p->hop, p->max_T) * sizeof(float));
        if (rc) return rc;
        GL_HIP_CHECK(hipMemcpy(p->d_w2, w2.data(), sizeof(double) * N, hipMemcpyHostToDevice));
        p->w2_uploaded = 1;
        p->wsum_T = -1;
    }
    p->opt_flags = flags;
    p->momentum = momentum;
    return 0;
}

static int check_call(const gl_plan* p, int B, int T, const char* who) {
    if (!p) return gl_set_error(GL_ERR_INVALID, "%s: plan is NULL", who);
    if (B < 1 || B > p->max_B) return gl_set_error(GL_ERR_INVALID, "%s: B=%d outside [1, %d]", who, B, p->max_B);
    if (T < 1 || T > p->max_T) return gl_set_error(GL_ERR_INVALID, "%s: T=%d outside [1, %d]", who, T, p->max_T);
    if ((long long)B * T > p->cap_frames)
        return gl_set_error(GL_ERR_INVALID, "%s: B*T = %lld frames beyond the plan's capacity of %lld", who,
                            (long long)B * T, p->cap_frames);
    return 0;
}

static bool use_fast(const gl_plan* p, int op) { return p->path == 0 && gl_fast_supported(p, op); }

int gl_stft(gl_plan* p, const float* x, long long x_stride, float* S, int B, int T, void* stream) {
    int rc = check_call(p, B, T, "gl_stft");
    if (rc) return rc;
    if (!x || !S) return gl_set_error(GL_ERR_INVALID, "gl_stft: NULL buffer");
    if (x_stride < (long long)(T - 1) * p->hop + p->n_fft)
        return gl_set_error(GL_ERR_INVALID, "gl_stft: x_stride %lld shorter than (T-1)*H+N", x_stride);
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    return use_fast(p, GL_FAST_OP_STFT) ? gl_fast_stft(p, x, x_stride, S, B, T, st) : gl_generic_stft(p, x, x_stride, S, B, T, st);
}

int gl_istft(gl_plan* p, const float* S, float* y, int B, int T, void* stream) {
    int rc = check_call(p, B, T, "gl_istft");
    if (rc) return rc;
    if (!S || !y) return gl_set_error(GL_ERR_INVALID, "gl_istft: NULL buffer");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    rc = use_fast(p, GL_FAST_OP_ISTFT) ? gl_fast_istft(p, S, y, B, T, st) : gl_generic_istft(p, S, y, B, T, st);
    if (rc || !(p->opt_flags & GL_OPT_WSUM_NORMALISE)) return rc;
    rc = gl_wsum_enqueue(p, T, st);
    // momentum belongs to the loop: the single inverse transform is only normalised
    const float momentum = p->momentum;
    p->momentum = 0.0f;
    if (!rc) rc = gl_post_iter(p, y, nullptr, y, B, T, 1, 1, true, true, st);
    p->momentum = momentum;
    return rc;
}

int gl_mel_to_mag(gl_plan* p, const float* filterbank, const float* mel, float* mag, int n_mels, int B, int T,
                  void* stream) {
    int rc = check_call(p, B, T, "gl_mel_to_mag");
    if (rc) return rc;
    if (!filterbank || !mel || !mag) return gl_set_error(GL_ERR_INVALID, "gl_mel_to_mag: NULL buffer");
    if (n_mels < 1 || n_mels > 4096) return gl_set_error(GL_ERR_INVALID, "gl_mel_to_mag: n_mels=%d", n_mels);
    DeviceGuard guard(p->device);
    return gl_mel_launch(filterbank, mel, mag, n_mels, p->n_fft / 2 + 1, B, T, false, 1.0f,
                         (p->opt_flags & GL_OPT_CLAMP_MAG) != 0, p->n_cu, (hipStream_t)stream);
}

int gl_logmel_to_mag(gl_plan* p, const float* filterbank, const float* logmel, float* mag, int n_mels,
                     float compression_factor, int B, int T, void* stream) {
    int rc = check_call(p, B, T, "gl_logmel_to_mag");
    if (rc) return rc;
    if (!filterbank || !logmel || !mag) return gl_set_error(GL_ERR_INVALID, "gl_logmel_to_mag: NULL buffer");
    if (n_mels < 1 || n_mels > 4096) return gl_set_error(GL_ERR_INVALID, "gl_logmel_to_mag: n_mels=%d", n_mels);
    if (!(compression_factor > 0.0f))
        return gl_set_error(GL_ERR_INVALID, "gl_logmel_to_mag: compression_factor must be > 0");
    DeviceGuard guard(p->device);
    return gl_mel_launch(filterbank, logmel, mag, n_mels, p->n_fft / 2 + 1, B, T, true, compression_factor,
                         (p->opt_flags & GL_OPT_CLAMP_MAG) != 0, p->n_cu, (hipStream_t)stream);
}

int gl_analysis_num_frames(int hop, long long len) {
    if (hop < 1 || len < 1) return 0;
    return (int)(1 + len / hop);      // librosa.stft(center=True): 1 + len // hop_length
}

int gl_wav_to_logmel(gl_plan* p, const float* y, long long y_stride, long long len, const float* filterbank,
                     float* mag, float* logmel, int n_mels, int pad_mode, float clip_min, float compression_factor,
                     int B, void* stream) {
    if (!p || !y || !filterbank || !mag || !logmel) return gl_set_error(GL_ERR_INVALID, "gl_wav_to_logmel: NULL argument");
    const int T = gl_analysis_num_frames(p->hop, len);
    int rc = check_call(p, B, T > 0 ? T : 1, "gl_wav_to_logmel");
    if (rc) return rc;
    if (len < 1 || y_stride < len) return gl_set_error(GL_ERR_INVALID, "gl_wav_to_logmel: bad len/stride");
    if (pad_mode != 0 && pad_mode != 1) return gl_set_error(GL_ERR_INVALID, "gl_wav_to_logmel: pad_mode must be 0 or 1");
    if (pad_mode == 1 && len <= p->n_fft / 2)
        return gl_set_error(GL_ERR_INVALID, "gl_wav_to_logmel: reflect padding needs len > n_fft/2 (got %lld)", len);
    if (n_mels < 1 || n_mels > 128) return gl_set_error(GL_ERR_INVALID, "gl_wav_to_logmel: n_mels=%d outside [1, 128]", n_mels);
    if (!(compression_factor > 0.0f)) return gl_set_error(GL_ERR_INVALID, "gl_wav_to_logmel: compression_factor must be > 0");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    rc = gl_analysis_mag(p, y, y_stride, len, mag, pad_mode, B, T, st);
    if (rc) return rc;
    return gl_mag_to_logmel_launch(filterbank, mag, logmel, n_mels, p->n_fft / 2 + 1, B, T, clip_min,
                                   compression_factor, st);
}

long long gl_resample_length(long long n_in, int sr_in, int sr_out) {
    if (n_in < 0 || sr_in < 1 || sr_out < 1) return 0;
    return (long long)ceil((double)n_in * ((double)sr_out / (double)sr_in));      // librosa.resample: ceil(n * ratio)
}

int gl_resample(gl_plan* p, const float* x, long long x_stride, long long n_in, float* y, long long y_stride,
                int sr_in, int sr_out, int B, void* stream) {
    if (!p || !x || !y) return gl_set_error(GL_ERR_INVALID, "gl_resample: NULL argument");
    if (sr_in < 1 || sr_out < 1) return gl_set_error(GL_ERR_INVALID, "gl_resample: sample rates must be positive");
    if (B < 1 || B > 65535) return gl_set_error(GL_ERR_INVALID, "gl_resample: B=%d outside [1, 65535]", B);
    const long long n_out = gl_resample_length(n_in, sr_in, sr_out);
    if (n_in < 1 || x_stride < n_in || y_stride < n_out || n_out < 1)
        return gl_set_error(GL_ERR_INVALID, "gl_resample: bad lengths / strides (n_in %lld, n_out %lld)", n_in, n_out);
    const double ratio = (double)sr_out / (double)sr_in;
    if ((int)((ratio < 1.0 ? ratio : 1.0) * GL_RS_TABLE) < 1)
        return gl_set_error(GL_ERR_INVALID, "gl_resample: ratio %g too small for the filter table", ratio);
    DeviceGuard guard(p->device);
    const long long n_calc = (long long)((double)n_in * ratio);                   // resampy: int(n * ratio) samples
    return gl_resample_launch(p, x, x_stride, n_in, y, y_stride, n_out, n_calc < n_out ? n_calc : n_out, ratio, B,
                              (hipStream_t)stream);
}

int gl_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T, void* stream) {
    int rc = check_call(p, B, T, "gl_run");
    if (rc) return rc;
    if (!mag || !x0 || !x_out) return gl_set_error(GL_ERR_INVALID, "gl_run: NULL buffer");
    if (iters < 0) return gl_set_error(GL_ERR_INVALID, "gl_run: iters=%d", iters);
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    if (iters == 0) return gl_copy_rows(x_out, x0, (long long)B * gl_len(p->n_fft, p->hop, T), st);
    if (p->opt_flags & GL_OPT_WSUM_NORMALISE) {
        rc = gl_wsum_enqueue(p, T, st);
        if (rc) return rc;
    }
    return use_fast(p, GL_FAST_OP_RUN) ? gl_fast_run(p, mag, x0, x_out, iters, B, T, st)
                       : gl_generic_run(p, mag, x0, x_out, iters, B, T, st);
}

int gl_mel_run_supported(const gl_plan* p) {
    // aligned dummies: only the plan's geometry and path are asked about here
    return p && use_fast(p, GL_FAST_OP_RUN) && gl_fast_takes_packed(p, nullptr, nullptr) ? 1 : 0;
}

int gl_mel_run(gl_plan* p, const float* filterbank, const float* mel, int n_mels, int mel_layout,
               float compression_factor, const int* frames, const float* x0, float* x_out, int iters, int B, int T,
               void* stream) {
    int rc = check_call(p, B, T, "gl_mel_run");
    if (rc) return rc;
    if (!filterbank || !mel || !x0 || !x_out) return gl_set_error(GL_ERR_INVALID, "gl_mel_run: NULL buffer");
    if (n_mels < 1 || n_mels > 4096) return gl_set_error(GL_ERR_INVALID, "gl_mel_run: n_mels=%d", n_mels);
    if (mel_layout != 0 && mel_layout != 1) return gl_set_error(GL_ERR_INVALID, "gl_mel_run: mel_layout must be 0 or 1");
    if (mel_layout == 1 && !(compression_factor > 0.0f))
        return gl_set_error(GL_ERR_INVALID, "gl_mel_run: compression_factor must be > 0");
    if (iters < 0) return gl_set_error(GL_ERR_INVALID, "gl_mel_run: iters=%d", iters);
    if (!use_fast(p, GL_FAST_OP_RUN) || !gl_fast_takes_packed(p, x0, x_out))
        return gl_set_error(GL_ERR_UNSUPPORTED,
                            "gl_mel_run: this plan's loop reads plain magnitudes (needs n_fft 2048, hop 128, 256, 512 or "
                            "1024, the default path and 8-byte aligned signals): call gl_mel_to_mag + gl_run");
    DeviceGuard guard(p->device);
    hipStream_t st = (hipStream_t)stream;
    if (iters == 0) return gl_copy_rows(x_out, x0, (long long)B * gl_len(p->n_fft, p->hop, T), st);
    rc = gl_mel_launch(filterbank, mel, p->d_magp, n_mels, p->n_fft / 2 + 1, B, T, mel_layout == 1,
                       mel_layout == 1 ? compression_factor : 1.0f, (p->opt_flags & GL_OPT_CLAMP_MAG) != 0, p->n_cu, st,
                       true, frames);
    if (rc) return rc;
    if (p->opt_flags & GL_OPT_WSUM_NORMALISE) {
        rc = gl_wsum_enqueue(p, T, st);
        if (rc) return rc;
    }
    return gl_fast_run_packed(p, x0, x_out, iters, B, T, st);
}

struct gl_graph {
    hipGraphExec_t exec;
    int device;
};

int gl_graph_begin(void* stream) {
    if (!stream) return gl_set_error(GL_ERR_INVALID, "gl_graph_begin: capture needs a non-default stream");
    GL_HIP_CHECK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
    return 0;
}

int gl_graph_end(void* stream, gl_graph** out) {
    if (!stream || !out) return gl_set_error(GL_ERR_INVALID, "gl_graph_end: NULL argument");
    *out = nullptr;
    hipGraph_t graph = nullptr;
    GL_HIP_CHECK(hipStreamEndCapture((hipStream_t)stream, &graph));
    if (!graph) return gl_set_error(GL_ERR_HIP, "gl_graph_end: the capture was invalidated");
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return gl_set_error(GL_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    gl_graph* g = new (std::nothrow) gl_graph();
    if (!g) {
        (void)hipGraphExecDestroy(exec);
        return gl_set_error(GL_ERR_NOMEM, "host allocation failed");
    }
    g->exec = exec;
    g->device = -1;
    (void)hipGetDevice(&g->device);
    *out = g;
    return 0;
}

int gl_graph_launch(gl_graph* g, void* stream) {
    if (!g) return gl_set_error(GL_ERR_INVALID, "gl_graph_launch: graph is NULL");
    GL_HIP_CHECK(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return 0;
}

int gl_graph_destroy(gl_graph* g) {
    if (!g) return 0;
    (void)hipGraphExecDestroy(g->exec);
    delete g;
    return 0;
}

int gl_mask_frames(gl_plan* p, float* mag, const int* frames, int B, int T, void* stream) {
    int rc = check_call(p, B, T, "gl_mask_frames");
    if (rc) return rc;
    if (!mag || !frames) return gl_set_error(GL_ERR_INVALID, "gl_mask_frames: NULL buffer");
    DeviceGuard guard(p->device);
    return gl_mask_launch(mag, frames, B, T, p->n_fft / 2 + 1, (hipStream_t)stream);
}

int gl_randn(float* out, long long n, uint64_t seed, void* stream) {
    if (!out || n < 0) return gl_set_error(GL_ERR_INVALID, "gl_randn: bad argument");
    if (n == 0) return 0;
    return gl_randn_launch(out, n, seed, 0, (hipStream_t)stream);
}

int gl_randn_at(float* out, long long n, uint64_t seed, long long offset, void* stream) {
    if (!out || n < 0 || offset < 0) return gl_set_error(GL_ERR_INVALID, "gl_randn_at: bad argument");
    if (n == 0) return 0;
    return gl_randn_launch(out, n, seed, offset, (hipStream_t)stream);
}

int gl_stitch_add(float* out, long long out_len, const float* chunk, long long chunk_len, long long out_offset,
                  long long fade_in, long long fade_out, void* stream) {
    if (!out || !chunk || out_len < 0 || chunk_len < 0 || fade_in < 0 || fade_out < 0)
        return gl_set_error(GL_ERR_INVALID, "gl_stitch_add: bad argument");
    if (fade_in + fade_out > chunk_len)
        return gl_set_error(GL_ERR_INVALID, "gl_stitch_add: fades (%lld + %lld) longer than the chunk (%lld)", fade_in,
                            fade_out, chunk_len);
    if (chunk_len == 0) return 0;
    return gl_stitch_launch(out, out_len, chunk, chunk_len, out_offset, fade_in, fade_out, (hipStream_t)stream);
}

int gl_peak_normalise(gl_plan* p, float* x, long long row_stride, long long len, float* peaks, int mode, int B,
                      void* stream) {
    if (!p || !x) return gl_set_error(GL_ERR_INVALID, "gl_peak_normalise: NULL argument");
    if (B < 1 || B > p->max_B) return gl_set_error(GL_ERR_INVALID, "gl_peak_normalise: B=%d outside [1, %d]", B, p->max_B);
    if (len < 1 || row_stride < len || (mode != 0 && mode != 1))
        return gl_set_error(GL_ERR_INVALID, "gl_peak_normalise: bad len/stride/mode");
    DeviceGuard guard(p->device);
    return gl_peak_launch(p, x, row_stride, len, peaks, mode, B, (hipStream_t)stream);
}

int gl_pcm16_encode(gl_plan* p, const float* x, long long row_stride, long long len, int16_t* pcm,
                    long long pcm_stride, float* peaks, int B, void* stream) {
    if (!p || !x || !pcm) return gl_set_error(GL_ERR_INVALID, "gl_pcm16_encode: NULL argument");
    if (B < 1 || B > p->max_B) return gl_set_error(GL_ERR_INVALID, "gl_pcm16_encode: B=%d outside [1, %d]", B, p->max_B);
    if (len < 1 || row_stride < len || pcm_stride < len)
        return gl_set_error(GL_ERR_INVALID, "gl_pcm16_encode: bad len/stride");
    DeviceGuard guard(p->device);
    return gl_pcm16_launch(p, x, row_stride, len, pcm, pcm_stride, peaks, B, (hipStream_t)stream);
}

}  // extern "C"
