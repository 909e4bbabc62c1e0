// Internal declarations shared by the translation units of libgl_hip.so.
// gfx950 (MI355X) only; see include/gl_hip.h for the public C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/gl_hip.h"

struct gl_plan {
    int n_fft;        // N
    int hop;          // H
    int max_T;
    int max_B;
    long long cap_frames;   // calls need B * T <= cap_frames: every frame-proportional buffer is sized from it
    int device;
    int path;         // 0 auto, 1 generic
    int run_frames;   // frames per wave-run of the fused kernels; 0 = choose from (B, T)
    int n_cu;         // compute units of the device
    float* d_window;  // [N]   symmetric Hann
    float2* d_tw;     // [N]   exp(-2*pi*i*k/N)
    // Everything below is allocated by gl_plan_create (what the plan's default path can touch), by
    // gl_plan_set_path(1) (generic scratch of a plan whose geometry has fused kernels) or by gl_plan_set_options
    // (the variants' buffers) -- never by gl_run / gl_stft / gl_istft, which only enqueue work.
    float* d_frames;  // generic path scratch [cap_frames][N]
    float* d_xa;      // ping-pong signal buffers, gl_signal_floats() each
    float* d_xb;
    float* d_xc;      // third signal buffer of the n_fft-1024 run kernel (read / write / clear rotation)
    float* d_tail_a;  // n_fft 2048 fused kernels: per-run partial-sum tails
    float* d_tail_b;
    float* d_fast_tab;  // constant table of k_fast2048
    float* d_fast_tab2; // constant table of the register-FFT kernels
    float* d_magp;      // register-ring kernel: magnitudes repacked per lane [cap_frames][1088]
    unsigned* d_peaks;  // [max_B] bit patterns of max|x|
    int opt_flags;      // GL_OPT_* (gl_plan_set_options)
    float momentum;     // fast Griffin-Lim extrapolation factor, 0 = off
    double* d_w2;       // [N] squared window in double (window-sum option)
    float* d_inv_wsum;  // [L(max_T)] 1 / sum_t w^2(s - tH), rebuilt on the call's stream when T changes
    int wsum_T;         // frame count the table holds (-1: unknown)
    int wsum_captured;  // a stream capture has recorded a table build: replays rewrite the table behind the host's back
    int w2_uploaded;    // d_w2 holds the squared window
    float* d_prev;      // gl_signal_floats(): previous iterate for the momentum option
    float* d_window_p;  // [N] periodic Hann of the analysis path
    double* d_rs_win;   // resampler: [GL_RS_ZEROS * GL_RS_TABLE + 1] x (filter value, difference to the next one)
    float* d_yframes;   // small problems: two sets of windowed inverse frames [2][frames][N]
    size_t yframes_bytes;
    long long bytes;
};

// error plumbing (gl_api.hip)
int gl_set_error(int code, const char* fmt, ...);
#define GL_HIP_CHECK(expr)                                                            \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            (void)hipGetLastError(); /* reported here: do not leave it for the next caller of the runtime */ \
            return gl_set_error(GL_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                                hipGetErrorString(e_), __FILE__, __LINE__);           \
        }                                                                             \
    } while (0)

static inline long long gl_len(int N, int H, int T) { return (long long)T * H + N; }
// floats of one plan-owned signal buffer [B][L]: B * L = (B * T) * H + B * N <= cap_frames * H + min(max_B, cap_frames) * N
static inline size_t gl_signal_floats(const gl_plan* p) {
    const long long b = p->max_B < p->cap_frames ? p->max_B : p->cap_frames;
    return (size_t)(p->cap_frames * p->hop + b * p->n_fft);
}
int gl_plan_alloc(gl_plan* p, void** ptr, size_t bytes);  // tracked hipMalloc

// ---- generic (any power-of-two N) kernels: gl_generic.hip ---------------------
int gl_generic_stft(gl_plan* p, const float* x, long long x_stride, float* S, int B, int T, hipStream_t st);
int gl_generic_istft(gl_plan* p, const float* S, float* y, int B, int T, hipStream_t st);
int gl_generic_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                   hipStream_t st);

int gl_generic_alloc(gl_plan* p);     // d_frames + ping-pong signals
int gl_analysis_mag(gl_plan* p, const float* y, long long y_stride, long long len, float* mag, int reflect, int B,
                    int T, hipStream_t st);

// ---- fused wave-per-frame kernels: gl_fast.hip -----------------------------------
enum { GL_FAST_OP_STFT = 0, GL_FAST_OP_ISTFT = 1, GL_FAST_OP_RUN = 2 };
bool gl_fast_supported(const gl_plan* p, int op);
int gl_fast_alloc(gl_plan* p);        // tables and scratch of the fused entry points (gl_plan_create)
// per-iteration post pass of the off-by-default options (window-sum normalisation, momentum):
// out[s] = f(x_main[s] + tail) ; tail == nullptr when x_main is already complete
static inline bool gl_post_needed(const gl_plan* p) {
    return (p->opt_flags & GL_OPT_WSUM_NORMALISE) != 0 || p->momentum != 0.0f;
}
int gl_post_iter(gl_plan* p, const float* x_main, const float* tail, float* out, int B, int T, int G, int nruns,
                 bool first, bool last, hipStream_t st);
int gl_wsum_enqueue(gl_plan* p, int T, hipStream_t st);   // window-sum table for T frames, built on the stream
int gl_fast_stft(gl_plan* p, const float* x, long long x_stride, float* S, int B, int T, hipStream_t st);
int gl_fast_istft(gl_plan* p, const float* S, float* y, int B, int T, hipStream_t st);
int gl_fast_analysis_mag(gl_plan* p, const float* y, long long y_stride, long long len, float* mag, int reflect,
                         int B, int T, const float* win_p, hipStream_t st);
int gl_fast_run(gl_plan* p, const float* mag, const float* x0, float* x_out, int iters, int B, int T,
                hipStream_t st);
// the loop on magnitudes ALREADY in p->d_magp (gl_mel_launch(packed) wrote them): gl_mel_run
bool gl_fast_takes_packed(const gl_plan* p, const void* x0, const void* x_out);
int gl_fast_run_packed(gl_plan* p, const float* x0, float* x_out, int iters, int B, int T, hipStream_t st);

// ---- mel inversion (MFMA): gl_mel.hip ---------------------------------------------
// packed: `mag` is [B][T][1088] in the register-ring kernels' own layout (n_fft 2048), rows t >= frames[b] zeroed
int gl_mel_launch(const float* fb, const float* mel, float* mag, int n_mels, int F, int B, int T, bool logmel,
                  float compression_factor, bool clamp, int n_cu, hipStream_t st, bool packed = false,
                  const int* frames = nullptr);

int gl_mag_to_logmel_launch(const float* fb, const float* mag, float* logmel, int n_mels, int F, int B, int T,
                            float clip_min, float compression_factor, hipStream_t st);

// ---- misc: gl_misc.hip -------------------------------------------------------------
// resampy 'kaiser_best': 64 zero crossings, 512 table points per crossing, Kaiser window and roll-off as below
#define GL_RS_ZEROS 64
#define GL_RS_TABLE 512
#define GL_RS_BETA 14.769656459379492
#define GL_RS_ROLLOFF 0.9475937167399596
int gl_resample_launch(const gl_plan* p, const float* x, long long x_stride, long long n_in, float* y,
                       long long y_stride, long long n_out, long long n_calc, double ratio, int B, hipStream_t st);
int gl_randn_launch(float* out, long long n, uint64_t seed, long long offset, hipStream_t st);
int gl_stitch_launch(float* out, long long out_len, const float* chunk, long long chunk_len, long long out_offset,
                     long long fade_in, long long fade_out, hipStream_t st);
int gl_peak_launch(gl_plan* p, float* x, long long row_stride, long long len, float* peaks, int mode, int B,
                   hipStream_t st);
int gl_static_scratch_setup(int device, unsigned** ptr, int* n_cu);   // gl_peak_normalise_static
int gl_peak_launch_on(unsigned* d_peaks, int n_cu, float* x, long long row_stride, long long len, float* peaks,
                      int mode, int B, hipStream_t st);
int gl_mask_launch(float* mag, const int* frames, int B, int T, int F, hipStream_t st);
int gl_copy_rows(float* dst, const float* src, long long n, hipStream_t st);
int gl_pcm16_launch(gl_plan* p, const float* x, long long row_stride, long long len, int16_t* pcm,
                    long long pcm_stride, float* peaks, int B, hipStream_t st);

// ---- small complex helpers ------------------------------------------------------------
__device__ __forceinline__ float2 c_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 c_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 c_mul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 c_mulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 c_conj(float2 a) { return make_float2(a.x, -a.y); }

// Magnitude projection, audio_utilities.py:181-183: mag * exp(1j*angle(X)), computed as
// X * (mag * rsq(|X|^2)) with one v_rsq_f32.  np.angle(0) == 0, so a zero bin becomes
// (mag, 0); bins whose |X|^2 is below 1e-36 (|X| < 1e-18, at the edge of fp32's normal
// range, 11 orders of magnitude under the fp32 rounding noise of a unit-scale frame)
// are treated as zero.  Branch-free.
#define GL_TINY_R2 1e-36f
__device__ __forceinline__ float2 gl_project(float2 X, float m) {
    const float r2 = fmaf(X.x, X.x, X.y * X.y);
    const float s = m * __builtin_amdgcn_rsqf(fmaxf(r2, GL_TINY_R2));
    const bool ok = r2 >= GL_TINY_R2;
    return make_float2(ok ? X.x * s : m, ok ? X.y * s : 0.0f);
}
